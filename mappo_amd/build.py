"""Build libmappo_hip.so (gfx950) in-tree with hipcc.  `python -m mappo_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmappo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(not os.path.exists(d) or os.path.getmtime(d) > t for d in deps)


def _depfile_deps(depfile):
    """Prerequisites recorded by `hipcc -MD -MF` for one object (None if there is no usable depfile yet)."""
    try:
        with open(depfile) as f:
            text = f.read().replace("\\\n", " ")
    except OSError:
        return None
    if ":" not in text:
        return None
    deps = [d for d in text.split(":", 1)[1].split() if not d.startswith("/opt/rocm")]
    return deps or None


def build(force=False, verbose=True, extra_flags=(), lib=None, objdir=None):
    """Compile every csrc/*.hip (in parallel) and link `lib` (default: the in-tree product library).  `extra_flags`,
    `lib`, `objdir` serve diagnostic builds (scripts/stamps.py) that must not touch the product library."""
    lib = lib or LIB
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mappo_hip.h"))
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o" if objdir is None else os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        # the object's own header list (from the last compile) decides staleness: editing one kernel header does not
        # rebuild the translation units that never include it
        deps = _depfile_deps(o + ".d")
        if force or _stale(o, deps if deps is not None else [s] + headers):
            cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-MD", "-MF", o + ".d", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"FAILED: {s}\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
