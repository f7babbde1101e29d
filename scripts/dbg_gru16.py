"""Diagnostic: per-parameter gradient error of the recurrent golden ppo_update case (scripts only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import golden, sub
import test_gpu_e2e as E

class NS: pass
def mk():
    import mappo_amd
    from mappo_amd.config import get_config
    from mappo_amd.utils.shared_buffer import SharedReplayBuffer
    from mappo_amd.utils.util import Discrete
    from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO
    from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    ns = NS(); ns.get_config, ns.SharedReplayBuffer, ns.Discrete, ns.R_MAPPO, ns.R_MAPPOPolicy = get_config, SharedReplayBuffer, Discrete, R_MAPPO, R_MAPPOPolicy
    return ns
M = mk()
g = golden("ppo_update"); c = 9; d = sub(g, f"c{c}")
T, N, Ma, D, S, A, H = [int(x) for x in d["dims"]]
hy = d["hyper"]
a = E.make_args(M, episode_length=T, n_rollout_threads=N, lr=float(hy[5]), critic_lr=float(hy[6]), use_recurrent_policy=True, data_chunk_length=int(hy[9]))
pol = E.load_policy(M, a, g, f"c{c}", D, S, A)
tr = M.R_MAPPO(a, pol)
E.set_vn(tr, d["vn0"])
sample = tuple(d[f"sample/{nm}"] for nm in E.TUPLE)
out = tr.ppo_update(sample)
print("stats", np.array(out), d["r0/stats"])
for tag, net, seg in (("actor", pol.actor, 0), ("critic", pol.critic, 1)):
    lo = pol.seg_bounds[seg]
    for key, off, shape in net.layout:
        n = int(np.prod(shape))
        got = pol.flat_grad[lo + off: lo + off + n].view(shape).cpu().numpy().astype(np.float64)
        ref = np.asarray(d[f"r0/{tag}_grad/{key}"], dtype=np.float64)
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)
        print(f"{tag:6s} {key:34s} max|ref| {np.abs(ref).max():.3e} relmax err {err:.2e}")
        if err > 1e-3 and got.ndim == 2 and os.environ.get("DBG_ROWS"):
            e = np.abs(got - ref).max(axis=1) / max(np.abs(ref).max(), 1e-12)
            print("   rows with err:", np.nonzero(e > 1e-4)[0][:40])
