"""Diagnostic: experiment builds of ONE translation unit.  For every `name=flags` argument compile csrc/<tu>.hip with the extra
flags and link it with the product objects of the other translation units into build_diag/lib_<name>.so (travels to the GPU
box; load with MAPPO_HIP_LIB).  usage: python scripts/exp_build.py [--tu mlp_upd16_r1_l1] name="-DEXP_A -DEXP_B=2" ..."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mappo_amd import build as B
args = sys.argv[1:]
tu = "mlp_upd16_r1_l1"
if args and args[0] == "--tu":
    tu = args[1]; args = args[2:]
B.build(verbose=False)
out = os.path.join(ROOT, "build_diag"); os.makedirs(out, exist_ok=True)
others = [s[:-4] + ".o" for s in B.sources() if os.path.basename(s) != tu + ".hip"]
def one(spec):
    name, flags = spec.split("=", 1)
    o = os.path.join(out, f"{tu}_{name}.o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + ["-w"] + flags.split() + ["-c", os.path.join(B.CSRC, tu + ".hip"), "-o", o])
    lib = os.path.join(out, f"lib_{name}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o] + others)
    return lib
with ThreadPoolExecutor(4) as ex:
    for lib in ex.map(one, args):
        print(lib)
