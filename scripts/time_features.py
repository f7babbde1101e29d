"""Diagnostic (GPU box): the dual trunk-features launch of a recurrent actor + critic at rollout size for several input widths;
run under rocprofv3 (scripts/prof_any.sh) for kernel times.  usage: python scripts/time_features.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mappo_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 640
torch.manual_seed(0)
for Da, Dc in ((176, 322), (176, 320), (128, 128), (512, 512), (30, 48)):
    da, dc = ops.net_desc(Da, 18, recurrent=True), ops.net_desc(Dc, 1, recurrent=True)
    pa = torch.randn(ops.net_param_count(da), device="cuda") * 0.1
    pc = torch.randn(ops.net_param_count(dc), device="cuda") * 0.1
    xa, xc = torch.randn(B, Da, device="cuda"), torch.randn(B, Dc, device="cuda")
    fa, fc = torch.empty(64, B, device="cuda"), torch.empty(64, B, device="cuda")
    for _ in range(5):
        ops.mlp_features_dual(pa, da, xa, fa, pc, dc, xc, fc, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            ops.mlp_features_dual(pa, da, xa, fa, pc, dc, xc, fc, B)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"B={B} D=({Da},{Dc}): {e0.elapsed_time(e1) * 20:.1f} us per launch (back to back in a graph)")
