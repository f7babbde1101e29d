"""CPU, world_size 2, gloo: the data-parallel protocol of mappo_amd/distributed.py.

Two ranks each own half of the rollout threads.  With the oracle as the (CPU) compute, each rank evaluates the
loss on its shard against GLOBAL denominators obtained through DataParallel.all_reduce_sum_, all-reduces the flat
gradient, and must end with the single-process gradient / statistics / ValueNorm state; the advantage moments
follow the same route.  This is the host logic R_MAPPO runs around the HIP kernels when bench.py is launched with
--gpus N (there the tensors live in HBM and the backend is RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_batch(seed, T, N, M, D, S, A):
    rng = np.random.default_rng(seed)
    f = np.float32
    R = N * M
    d = dict(obs=rng.standard_normal((T, N, M, D)).astype(f), sobs=rng.standard_normal((T, N, M, S)).astype(f),
             actions=rng.integers(0, A, (T, N, M, 1)).astype(f), old=(-np.abs(rng.standard_normal((T, N, M, 1))) - 1).astype(f),
             active=(rng.random((T, N, M, 1)) > 0.3).astype(f), ret=(rng.standard_normal((T, N, M, 1)) * 3).astype(f),
             vold=rng.standard_normal((T, N, M, 1)).astype(f) * 0.3, vp=rng.standard_normal((T, N, M, 1)).astype(f) * 0.3)
    return d


def _loss_and_grads(O, args, actor, critic, d, vn_mean, vn_var, sum_active):
    """Gradients of the reference's objectives on one shard, scaled by GLOBAL denominators."""
    t = lambda x: torch.from_numpy(x.reshape(-1, x.shape[-1]))
    for p in list(actor.parameters()) + list(critic.parameters()):
        p.grad = None
    lp, ent_rows, z = None, None, None
    feats, _ = actor.features(t(d["obs"]), None, None)
    zl = actor.act.action_out.linear(feats)
    logp, ent, _ = actor.act.logp_entropy(zl, t(d["actions"]))
    act, adv, old = t(d["active"]), t(d["adv"]), t(d["old"])
    imp = torch.exp(logp - old)
    surr = torch.min(imp * adv, torch.clamp(imp, 1 - args.clip_param, 1 + args.clip_param) * adv)
    pl = (-surr * act).sum() / sum_active
    e = (ent * act.squeeze(-1)).sum() / sum_active
    (pl - args.entropy_coef * e).backward()
    v = critic(t(d["sobs"]), None, None)[0]
    tgt = (t(d["ret"]) - vn_mean) / np.sqrt(vn_var)
    vo = t(d["vold"])
    vclip = vo + (v - vo).clamp(-args.clip_param, args.clip_param)
    l = torch.max(O.huber_ref(tgt - v, args.huber_delta), O.huber_ref(tgt - vclip, args.huber_delta))
    vl = (l * act).sum() / sum_active
    (vl * args.value_loss_coef).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in list(actor.parameters()) + list(critic.parameters()) if p.grad is not None])
    stats = torch.tensor([vl.item(), pl.item(), e.item()], dtype=torch.float64)
    return flat.double(), stats


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from oracle import mappo_oracle as O
    from mappo_amd.distributed import DataParallel, init_from_env, shard_threads
    torch.set_num_threads(1)
    dp = init_from_env("gloo")
    assert isinstance(dp, DataParallel) and dp.world == world and dp.rank == rank
    T, N, M, D, S, A = 5, 6, 3, 10, 30, 5
    args = O.default_args()
    torch.manual_seed(0)
    actor, critic = O.ActorRef(args, D, A), O.CriticRef(args, S)
    full = _make_batch(0, T, N, M, D, S, A)
    lo, hi = shard_threads(N, rank, world)
    shard = {k: v[:, lo:hi] for k, v in full.items()}

    # --- advantage moments: local sums -> all-reduce -> identical normalisation everywhere (r_mappo.py:174-182)
    def moments(d):
        a = (d["ret"] - d["vp"]).astype(np.float64)
        m = d["active"] != 0
        return torch.tensor([a[m].sum(), (a[m] ** 2).sum(), float(m.sum())], dtype=torch.float64)
    mom = dp.all_reduce_sum_(moments(shard))
    mean = mom[0] / mom[2]; std = torch.sqrt(mom[1] / mom[2] - mean ** 2)
    ref_adv, ref_mean, ref_std = O.normalized_advantages_ref(np.concatenate([full["ret"], full["ret"][:1]]),
                                                             np.concatenate([full["vp"], full["vp"][:1]]),
                                                             np.concatenate([full["active"], full["active"][:1]]))
    np.testing.assert_allclose([mean.item(), std.item()], [ref_mean, ref_std], rtol=1e-5)
    for d in (shard, full):
        d["adv"] = ((d["ret"] - d["vp"] - np.float32(ref_mean)) / (np.float32(ref_std) + np.float32(1e-5))).astype(np.float32)

    # --- minibatch moments (global ValueNorm update + loss denominators), then flat gradient all-reduce
    def mb(d):
        r = d["ret"].astype(np.float64)
        return torch.tensor([r.sum(), (r ** 2).sum(), float(d["active"].sum()), float(r.size)], dtype=torch.float64)
    g = dp.all_reduce_sum_(mb(shard))
    np.testing.assert_allclose(g.numpy(), mb(full).numpy(), rtol=1e-12)
    vn = O.ValueNormRef(); vn.update(full["ret"].reshape(-1, 1))
    vn_d = O.ValueNormRef()
    bm, bsq, w = np.float32(g[0] / g[3]), np.float32(g[1] / g[3]), 0.99999
    vn_d.running_mean.mul_(w).add_(torch.tensor([bm]) * (1.0 - w)); vn_d.running_mean_sq.mul_(w).add_(torch.tensor([bsq]) * (1.0 - w))
    vn_d.debiasing_term.mul_(w).add_(1.0 * (1.0 - w))
    np.testing.assert_allclose(vn_d.state(), vn.state(), rtol=2e-6)
    m_, v_ = vn.mean_var()
    grad, stats = _loss_and_grads(O, args, actor, critic, shard, float(m_), float(v_), float(g[2]))
    dp.all_reduce_sum_(grad); dp.all_reduce_sum_(stats)
    ref_grad, ref_stats = _loss_and_grads(O, args, actor, critic, full, float(m_), float(v_), float(full["active"].sum()))
    np.testing.assert_allclose(grad.numpy(), ref_grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(stats.numpy(), ref_stats.numpy(), rtol=1e-5)
    t = torch.tensor([float(rank)]); dp.all_reduce_max_(t); assert t.item() == world - 1
    out[rank] = True
    dist.destroy_process_group()


def test_shard_threads_partition():
    from mappo_amd.distributed import shard_threads
    for n, w in ((1024, 8), (10, 4), (7, 2), (3, 4)):
        spans = [shard_threads(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_data_parallel_protocol_world2_gloo():
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(0) and out.get(1)
