"""CPU test of bench.py's N-rank launcher: `python bench.py --gpus N` with no launcher around it must start
N ranks itself (child process of torch.distributed.run, 127.0.0.1 rendezvous), and under the driver's own spelling
(`python -m torch.distributed.run ... bench.py --gpus N`) it must be one of the N ranks — never a 1-rank run that prints
n_gpus: 1.  `--dry_launch` makes every rank print its RANK / WORLD_SIZE and exit before any GPU call."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _ranks(out):
    recs = [json.loads(ln) for ln in out.splitlines() if ln.strip().startswith("{")]
    return sorted((r["rank"], r["world"], r["local_rank"], r["gpus"]) for r in recs)


def test_bare_gpus_flag_spawns_n_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dry_launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _ranks(p.stdout) == [(0, 3, 0, 3), (1, 3, 1, 3), (2, 3, 2, 3)]


def test_under_torchrun_is_one_of_the_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--dry_launch"]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _ranks(p.stdout) == [(0, 2, 0, 2), (1, 2, 1, 2)]


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry_launch"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_single_gpu_default_does_not_spawn():
    p = subprocess.run([sys.executable, BENCH, "--dry_launch"], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _ranks(p.stdout) == [(0, 1, 0, 1)]
