"""Two-process test of the REAL data-parallel trainer path (R_MAPPO._update(part=...), the moments / gradient all-reduces,
`shard_threads`, the rank-keyed sampling seed): 2 ranks on a split buffer must reproduce 1 rank on the whole buffer —
parameters, Adam moments, ValueNorm state and the logged statistics — with unequal shards and a second train() call.
Both the MLP trainer and the RECURRENT trainer (mappo_amd/recurrent.py: GRU policy at a small BASELINE configs[3] shape —
wide inputs 176 / 322, 18 actions, chunks of data_chunk_length, num_mini_batch = 2) are covered; the recurrent case feeds
shard-respecting chunk permutations: the reference permutes the GLOBAL chunk list (shared_buffer.py:397), a rank
can only permute its own chunks, so the 1-rank run is given the permutation whose minibatch k is the union of the ranks'
minibatches k.

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


# recurrent case: a small BASELINE configs[3] shape (wide inputs: the wide trunk kernels + GRU training kernels + wide trunk backward)
R_T, R_MA, R_D, R_S, R_A, R_L, R_NMB, R_EPOCHS = 12, 2, 176, 322, 18, 6, 2, 2


def _args(n_threads, recurrent=False):
    from mappo_amd.config import get_config
    a = get_config().parse_known_args([])[0]
    a.use_recurrent_policy = recurrent
    a.use_naive_recurrent_policy = False
    a.episode_length, a.n_rollout_threads, a.ppo_epoch, a.lr, a.critic_lr, a.seed = T, n_threads, 3, 7e-4, 7e-4, 1
    if recurrent:
        a.algorithm_name = "rmappo"
        a.episode_length, a.ppo_epoch, a.num_mini_batch, a.data_chunk_length, a.perm_device = R_T, R_EPOCHS, R_NMB, R_L, "cpu"
    return a


def _global_data_rec(it):
    rng = np.random.default_rng(300 + it)
    f = np.float32
    N, Tn, Mn = N_GLOBAL, R_T, R_MA
    act = rng.integers(0, R_A, (Tn, N, Mn, 1))
    avail = (rng.random((Tn + 1, N, Mn, R_A)) > 0.3).astype(f)
    np.put_along_axis(avail[:Tn], act, 1.0, axis=-1)
    return dict(
        share_obs=rng.standard_normal((Tn + 1, N, Mn, R_S)).astype(f), obs=rng.standard_normal((Tn + 1, N, Mn, R_D)).astype(f),
        rnn_states=rng.standard_normal((Tn + 1, N, Mn, 1, 64)).astype(f), rnn_states_critic=rng.standard_normal((Tn + 1, N, Mn, 1, 64)).astype(f),
        rewards=rng.standard_normal((Tn, N, Mn, 1)).astype(f), value_preds=(rng.standard_normal((Tn + 1, N, Mn, 1)) * 0.3).astype(f),
        returns=(rng.standard_normal((Tn + 1, N, Mn, 1)) * 2).astype(f), actions=act.astype(f),
        action_log_probs=(-np.abs(rng.standard_normal((Tn, N, Mn, 1))) - 1).astype(f),
        masks=(rng.random((Tn + 1, N, Mn, 1)) > 0.15).astype(f), active_masks=(rng.random((Tn + 1, N, Mn, 1)) > 0.2).astype(f),
        available_actions=avail)


def _rec_perms(lo, hi, world_shards):
    """Chunk permutations for train() call `it`, epoch e of the threads [lo, hi): every shard (lo_s, hi_s) of `world_shards`
    draws its own local permutation from a generator keyed by (it, e, lo_s); a buffer that spans several shards gets, for
    minibatch k, the concatenation of the shards' minibatches k mapped to its own chunk numbering (chunk = series * (T/L) +
    block, series = thread * M + agent)."""
    per = R_T // R_L
    out = []
    for it in range(ITERS):
        for e in range(R_EPOCHS):
            mbs = [[] for _ in range(R_NMB)]
            for (ls, hs) in world_shards:
                if ls < lo or hs > hi:
                    continue
                n_loc = (hs - ls) * R_MA * per
                pl = np.random.default_rng(1000 * it + 10 * e + ls).permutation(n_loc)
                h = n_loc // R_NMB
                for k in range(R_NMB):
                    mbs[k].append(pl[k * h:(k + 1) * h] + (ls - lo) * R_MA * per)
            out.append(np.concatenate([np.concatenate(m) for m in mbs]))
    return out


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


def _run(dist_group, lo, hi, recurrent=False, world_shards=None):
    """train() ITERS times on threads [lo, hi) of the global data; returns everything the ranks must agree on."""
    from mappo_amd.utils.util import Discrete
    from mappo_amd.utils.shared_buffer import SharedReplayBuffer
    from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO
    from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    a = _args(hi - lo, recurrent)
    torch.manual_seed(1)                                         # identical replicas
    dims = (R_MA, R_D, R_S, R_A) if recurrent else (MA, D, D * MA, A)
    pol = R_MAPPOPolicy(a, [dims[1]], [dims[2]], Discrete(dims[3]))
    tr = R_MAPPO(a, pol, dist_group=dist_group)
    buf = SharedReplayBuffer(a, dims[0], [dims[1]], [dims[2]], Discrete(dims[3]))
    if recurrent:
        perms = iter(_rec_perms(lo, hi, world_shards))
        buf._randperm = lambda n: torch.from_numpy(next(perms)).to(buf.device)     # the permutation stream of recurrent_rows
    infos = []
    for it in range(ITERS):
        g = _global_data_rec(it) if recurrent else _global_data(it)
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


def _worker(rank, world, port, backend, out_dir, recurrent=False):
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
        shards = [shard_threads(N_GLOBAL, r, world) for r in range(world)]
        out = _run(DataParallel(), lo, hi, recurrent, shards)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("recurrent", [False, True])
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_match_one_rank(gpu_device, tmp_path, backend, recurrent):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: fewer than 2 GPUs visible")
    import torch.multiprocessing as mp
    from mappo_amd.distributed import shard_threads
    shards = [shard_threads(N_GLOBAL, r, 2) for r in range(2)]
    ref = _run(None, 0, N_GLOBAL, recurrent, shards)             # single process, whole buffer
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, str(tmp_path), recurrent)) for r in range(2)]
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
