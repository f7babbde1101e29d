"""Two-process test of the REAL data-parallel trainer path (R_MAPPO._update(part=...), the moments / gradient all-reduces,
`shard_threads`, the rank-keyed sampling seed): 2 ranks on a split buffer must reproduce 1 rank on the whole buffer —
parameters, Adam moments, ValueNorm state and the logged statistics — with unequal shards and a second train() call.

backend "gloo": both ranks share GPU 0 (gloo moves the CUDA tensors through the host), so it runs on a one-GPU box and
covers the eager trainer wiring.  backend "nccl" (RCCL over xGMI, one rank per GPU, per-epoch hipGraph segments): skipped
when fewer than 2 GPUs are visible, so the driver's multi-GPU node exercises it."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, MA, D, A = 12, 3, 18, 5
N_GLOBAL = 7                       # 2 ranks: 4 + 3 threads (unequal shards)
ITERS = 2


def _args(n_threads):
    from mappo_amd.config import get_config
    a = get_config().parse_known_args([])[0]
    a.use_recurrent_policy = False
    a.use_naive_recurrent_policy = False
    a.episode_length, a.n_rollout_threads, a.ppo_epoch, a.lr, a.critic_lr, a.seed = T, n_threads, 3, 7e-4, 7e-4, 1
    return a


def _global_data(it):
    rng = np.random.default_rng(100 + it)
    f = np.float32
    N = N_GLOBAL
    return dict(
        share_obs=rng.standard_normal((T + 1, N, MA, D * MA)).astype(f), obs=rng.standard_normal((T + 1, N, MA, D)).astype(f),
        rewards=rng.standard_normal((T, N, MA, 1)).astype(f), value_preds=(rng.standard_normal((T + 1, N, MA, 1)) * 0.3).astype(f),
        returns=(rng.standard_normal((T + 1, N, MA, 1)) * 2).astype(f), actions=rng.integers(0, A, (T, N, MA, 1)).astype(f),
        action_log_probs=(-np.abs(rng.standard_normal((T, N, MA, 1))) - 1).astype(f),
        active_masks=(rng.random((T + 1, N, MA, 1)) > 0.2).astype(f))


def _run(dist_group, lo, hi):
    """train() ITERS times on threads [lo, hi) of the global data; returns everything the ranks must agree on."""
    from mappo_amd.utils.util import Discrete
    from mappo_amd.utils.shared_buffer import SharedReplayBuffer
    from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO
    from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    a = _args(hi - lo)
    torch.manual_seed(1)                                         # identical replicas
    pol = R_MAPPOPolicy(a, [D], [D * MA], Discrete(A))
    tr = R_MAPPO(a, pol, dist_group=dist_group)
    buf = SharedReplayBuffer(a, MA, [D], [D * MA], Discrete(A))
    infos = []
    for it in range(ITERS):
        g = _global_data(it)
        for k, v in g.items():
            getattr(buf, k).copy_(torch.from_numpy(np.ascontiguousarray(v[:, lo:hi])))
        infos.append(tr.train(buf))
    torch.cuda.synchronize()
    out = dict(params=pol.flat_params.cpu().numpy(), exp_avg=pol.exp_avg.cpu().numpy(), exp_avg_sq=pol.exp_avg_sq.cpu().numpy(),
               vn=tr.value_normalizer.state.cpu().numpy(), seed=np.array([pol.actor._seed % (2 ** 63)], np.int64))
    for i, info in enumerate(infos):
        for k, v in info.items():
            out[f"info{i}/{k}"] = np.float64(v)
    return out


def _worker(rank, world, port, backend, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from mappo_amd.distributed import DataParallel, shard_threads
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_threads(N_GLOBAL, rank, world)
        out = _run(DataParallel(), lo, hi)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_match_one_rank(gpu_device, tmp_path, backend):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: fewer than 2 GPUs visible")
    import torch.multiprocessing as mp
    ref = _run(None, 0, N_GLOBAL)                                # single process, whole buffer
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("data-parallel worker timed out")
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    # replicas stay identical (same reduced gradient, same Adam): bit for bit
    for k in ("params", "exp_avg", "exp_avg_sq", "vn"):
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=f"rank 0 vs rank 1: {k}")
    assert int(r0["seed"][0]) != int(r1["seed"][0]), "sampling streams must differ across ranks (parameters must not)"
    # 2 ranks == 1 rank up to the fp32 summation order of the gradient
    np.testing.assert_allclose(r0["params"], ref["params"], rtol=0, atol=3e-6, err_msg="parameters")
    np.testing.assert_allclose(r0["exp_avg"], ref["exp_avg"], rtol=2e-4, atol=1e-8, err_msg="exp_avg")
    np.testing.assert_allclose(r0["exp_avg_sq"], ref["exp_avg_sq"], rtol=4e-4, atol=1e-12, err_msg="exp_avg_sq")
    np.testing.assert_allclose(r0["vn"], ref["vn"], rtol=2e-6, atol=1e-9, err_msg="ValueNorm state")
    for k in ref:
        if k.startswith("info"):
            np.testing.assert_allclose(r0[k], ref[k], rtol=1e-4, atol=1e-6, err_msg=k)
            np.testing.assert_allclose(r1[k], ref[k], rtol=1e-4, atol=1e-6, err_msg=k)
