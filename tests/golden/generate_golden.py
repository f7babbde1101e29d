#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's hot-path modules (build container only).

Run:  python tests/golden/generate_golden.py            (needs /root/reference; writes *.npz here)

The reference ships no tests or golden vectors, so these outputs of the reference
itself are what pins the oracle (oracle/mappo_oracle.py) and, through it, the HIP kernels.  Only the
resulting .npz files (data: inputs + expected outputs) are committed; no reference source travels.
Import recipe: SURVEY.md §8(c) — register an empty `onpolicy` namespace package so that only the
torch/numpy-only modules get imported.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MAPPO_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
pkg = types.ModuleType("onpolicy")
pkg.__path__ = [os.path.join(REF, "onpolicy")]
sys.modules["onpolicy"] = pkg

from onpolicy.config import get_config                                             # noqa: E402
from onpolicy.utils.shared_buffer import SharedReplayBuffer                       # noqa: E402
from onpolicy.utils.valuenorm import ValueNorm                                     # noqa: E402
from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO                            # noqa: E402
from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy       # noqa: E402


class Discrete:  # matched by class name in the reference (utils/util.py:40-42)
    def __init__(self, n):
        self.n = n


def make_args(**kw):
    a = get_config().parse_known_args([])[0]
    a.use_recurrent_policy = False          # "mappo" wiring of train_mpe.py:68-75
    a.use_naive_recurrent_policy = False
    for k, v in kw.items():
        assert hasattr(a, k), k
        setattr(a, k, v)
    return a


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


def sd_arrays(prefix, module):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def fill_buffer(buf, rng, zero_mask_prob=0.15, dead_prob=0.2, avail_zero_prob=0.3):
    """Random contents for every buffer array, with zeros in masks / active / bad / avail."""
    f = np.float32
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "rewards",
                 "action_log_probs"):
        arr = getattr(buf, name)
        arr[...] = rng.standard_normal(arr.shape).astype(f)
    buf.action_log_probs[...] = -np.abs(buf.action_log_probs) - 0.5
    A = buf.available_actions.shape[-1]
    buf.actions[...] = rng.integers(0, A, buf.actions.shape).astype(f)
    buf.masks[...] = (rng.random(buf.masks.shape) > zero_mask_prob).astype(f)
    buf.bad_masks[...] = (rng.random(buf.masks.shape) > zero_mask_prob).astype(f)
    buf.active_masks[...] = (rng.random(buf.masks.shape) > dead_prob).astype(f)
    av = (rng.random(buf.available_actions.shape) > avail_zero_prob).astype(f)
    # the taken action must be available
    T = buf.actions.shape[0]
    idx = buf.actions.astype(np.int64)
    np.put_along_axis(av[:T], idx, 1.0, axis=-1)
    av[T, ..., 0] = 1.0
    buf.available_actions[...] = av


def buffer_arrays(prefix, buf):
    names = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns",
             "available_actions", "actions", "action_log_probs", "rewards", "masks", "bad_masks",
             "active_masks")
    return {f"{prefix}/{n}": getattr(buf, n).copy() for n in names}


# ----------------------------------------------------------------------------------------------
def gen_valuenorm():
    rng = np.random.default_rng(7)
    vn = ValueNorm(1)
    out = {}
    for i in range(4):
        x = (rng.standard_normal((37, 1)) * (1 + i) + 0.5 * i).astype(np.float32)
        vn.update(torch.from_numpy(x))
        out[f"x{i}"] = x
        out[f"state{i}"] = np.array([vn.running_mean.item(), vn.running_mean_sq.item(),
                                     vn.debiasing_term.item()], dtype=np.float32)
        out[f"norm{i}"] = vn.normalize(torch.from_numpy(x)).numpy()
        out[f"denorm{i}"] = vn.denormalize(x)
    save("valuenorm", **out)


def vn_with_state(k_updates, rng):
    vn = ValueNorm(1)
    for _ in range(k_updates):
        vn.update(torch.from_numpy((rng.standard_normal((50, 1)) * 3 + 1).astype(np.float32)))
    return vn


def vn_state(vn):
    return np.array([vn.running_mean.item(), vn.running_mean_sq.item(), vn.debiasing_term.item()],
                    dtype=np.float32)


def gen_gae():
    out = {}
    case = 0
    for (T, N) in ((5, 2), (25, 4), (8, 3)):
        for use_gae in (True, False):
            for ptl in (False, True):
                for use_vn in (True, False):
                    for k_upd in ((0, 3) if use_vn else (0,)):
                        rng = np.random.default_rng(100 + case)
                        a = make_args(episode_length=T, n_rollout_threads=N, use_gae=use_gae,
                                      use_proper_time_limits=ptl, use_valuenorm=use_vn)
                        buf = SharedReplayBuffer(a, 3, [18], [54], Discrete(5))
                        fill_buffer(buf, rng)
                        buf.returns[...] = 0
                        nv = rng.standard_normal((N, 3, 1)).astype(np.float32)
                        vn = vn_with_state(k_upd, rng) if use_vn else None
                        p = f"c{case}"
                        out.update({f"{p}/rewards": buf.rewards.copy(), f"{p}/value_preds": buf.value_preds.copy(),
                                    f"{p}/masks": buf.masks.copy(), f"{p}/bad_masks": buf.bad_masks.copy(),
                                    f"{p}/next_value": nv,
                                    f"{p}/flags": np.array([use_gae, ptl, use_vn], dtype=np.int64),
                                    f"{p}/hyper": np.array([a.gamma, a.gae_lambda], dtype=np.float64),
                                    f"{p}/vn_state": vn_state(vn) if use_vn else np.zeros(3, np.float32)})
                        buf.compute_returns(nv, vn)
                        out[f"{p}/returns"] = buf.returns.copy()
                        out[f"{p}/value_preds_after"] = buf.value_preds.copy()
                        case += 1
    out["n_cases"] = np.array(case)
    save("gae", **out)


def gen_advnorm():
    out = {}
    for case, use_vn in enumerate((True, False)):
        rng = np.random.default_rng(200 + case)
        a = make_args(episode_length=8, n_rollout_threads=4, use_valuenorm=use_vn)
        buf = SharedReplayBuffer(a, 3, [18], [54], Discrete(5))
        fill_buffer(buf, rng)
        buf.returns[...] = rng.standard_normal(buf.returns.shape).astype(np.float32)
        vn = vn_with_state(2, rng) if use_vn else None
        # r_mappo.py:174-182 executed by the reference itself: run R_MAPPO.train for one epoch with the
        # generator replaced by a recorder, so the advantages it was handed are the reference's own.
        a.ppo_epoch = 1
        pol = R_MAPPOPolicy(a, [18], [54], Discrete(5))
        tr = R_MAPPO(a, pol)
        if use_vn:
            tr.value_normalizer = vn
        captured = {}

        def recorder(advantages, num_mini_batch=None, mini_batch_size=None):
            captured["adv"] = advantages.copy()
            return iter(())

        buf.feed_forward_generator = recorder
        tr.train(buf)
        adv_n = captured["adv"]
        # auxiliary (not produced by the reference as outputs): the statistics it used
        raw = buf.returns[:-1] - (vn.denormalize(buf.value_preds[:-1]) if use_vn else buf.value_preds[:-1])
        raw[buf.active_masks[:-1] == 0.0] = np.nan
        mean, std = np.nanmean(raw), np.nanstd(raw)
        p = f"c{case}"
        out.update({f"{p}/returns": buf.returns.copy(), f"{p}/value_preds": buf.value_preds.copy(),
                    f"{p}/active_masks": buf.active_masks.copy(),
                    f"{p}/vn_state": vn_state(vn) if use_vn else np.zeros(3, np.float32),
                    f"{p}/use_vn": np.array(use_vn), f"{p}/adv": adv_n, f"{p}/mean": np.array(mean),
                    f"{p}/std": np.array(std)})
    out["n_cases"] = np.array(2)
    save("advnorm", **out)


TUPLE = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
         "masks", "active_masks", "old_action_log_probs", "adv_targ", "available_actions")


def gen_generators():
    out = {}
    case = 0
    specs = [("ff", 5, 4, 1, None), ("ff", 5, 4, 2, None), ("ff", 7, 3, 4, None),
             ("rec", 20, 2, 1, 10), ("rec", 25, 2, 2, 10), ("rec", 25, 3, 1, 10), ("rec", 8, 4, 2, 4),
             ("naive", 6, 4, 2, None), ("naive", 5, 3, 3, None)]
    for kind, T, N, nmb, L in specs:
        rng = np.random.default_rng(300 + case)
        a = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8)
        buf = SharedReplayBuffer(a, 3, [6], [18], Discrete(5))
        fill_buffer(buf, rng)
        buf.returns[...] = rng.standard_normal(buf.returns.shape).astype(np.float32)
        adv = rng.standard_normal(buf.rewards.shape).astype(np.float32)
        p = f"c{case}"
        out.update(buffer_arrays(p + "/buf", buf))
        out[p + "/adv"] = adv
        out[p + "/spec"] = np.array([{"ff": 0, "rec": 1, "naive": 2}[kind], T, N, 3, nmb, L or 0, 1000 + case])
        torch.manual_seed(1000 + case)
        if kind == "ff":
            gen = buf.feed_forward_generator(adv, nmb)
            S = T * N * 3
        elif kind == "rec":
            gen = buf.recurrent_generator(adv, nmb, L)
            S = (T * N * 3) // L
        else:
            gen = buf.naive_recurrent_generator(adv, nmb)
            S = N * 3
        batches = list(gen)
        torch.manual_seed(1000 + case)
        out[p + "/rand"] = torch.randperm(S).numpy()
        out[p + "/n_batches"] = np.array(len(batches))
        for bi, sample in enumerate(batches):
            for nm, arr in zip(TUPLE, sample):
                out[f"{p}/b{bi}/{nm}"] = arr
        case += 1
    out["n_cases"] = np.array(case)
    save("generators", **out)


def gen_insert():
    rng = np.random.default_rng(400)
    T, N, M = 4, 2, 3
    a = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8)
    buf = SharedReplayBuffer(a, M, [6], [18], Discrete(5))
    out = {}
    f = np.float32
    for s in range(T + 2):       # wraps around step
        d = dict(share_obs=rng.standard_normal((N, M, 18)).astype(f), obs=rng.standard_normal((N, M, 6)).astype(f),
                 rnn_a=rng.standard_normal((N, M, 1, 8)).astype(f), rnn_c=rng.standard_normal((N, M, 1, 8)).astype(f),
                 actions=rng.integers(0, 5, (N, M, 1)).astype(f), logp=rng.standard_normal((N, M, 1)).astype(f),
                 values=rng.standard_normal((N, M, 1)).astype(f), rewards=rng.standard_normal((N, M, 1)).astype(f),
                 masks=(rng.random((N, M, 1)) > 0.3).astype(f), bad=(rng.random((N, M, 1)) > 0.3).astype(f),
                 active=(rng.random((N, M, 1)) > 0.3).astype(f), avail=(rng.random((N, M, 5)) > 0.3).astype(f))
        for k, v in d.items():
            out[f"in{s}/{k}"] = v
        buf.insert(d["share_obs"], d["obs"], d["rnn_a"], d["rnn_c"], d["actions"], d["logp"], d["values"],
                   d["rewards"], d["masks"], d["bad"], d["active"], d["avail"])
        out[f"step_after{s}"] = np.array(buf.step)
        if s == T - 1:
            out.update(buffer_arrays("full", buf))
            buf.after_update()
            out.update(buffer_arrays("after_update", buf))
    out.update(buffer_arrays("final", buf))
    out["n_inserts"] = np.array(T + 2)
    save("insert", **out)


def gen_forward():
    out = {}
    case = 0
    specs = [dict(use_ReLU=True, rec=False, D=18, S=54, A=5, B=24),
             dict(use_ReLU=False, rec=False, D=18, S=54, A=5, B=24),
             dict(use_ReLU=True, rec=True, D=30, S=48, A=9, B=12),
             dict(use_ReLU=False, rec=True, D=18, S=54, A=5, B=12),
             dict(use_ReLU=True, rec=False, D=176, S=322, A=18, B=10)]
    for sp in specs:
        torch.manual_seed(500 + case)
        rng = np.random.default_rng(500 + case)
        a = make_args(use_ReLU=sp["use_ReLU"], use_recurrent_policy=sp["rec"])
        pol = R_MAPPOPolicy(a, [sp["D"]], [sp["S"]], Discrete(sp["A"]))
        # perturb LayerNorm affine / biases / head so that they are not the trivial (1, 0) init
        with torch.no_grad():
            for net in (pol.actor, pol.critic):
                for n_, p_ in net.named_parameters():
                    if "norm" in n_ or "bias" in n_ or ".2." in n_:
                        p_.add_(0.1 * torch.randn_like(p_))
                    if "action_out" in n_ and "weight" in n_:
                        p_.mul_(50.0)
        B, D, S, A, H = sp["B"], sp["D"], sp["S"], sp["A"], a.hidden_size
        f = np.float32
        obs = rng.standard_normal((B, D)).astype(f)
        sobs = rng.standard_normal((B, S)).astype(f)
        ha = rng.standard_normal((B, 1, H)).astype(f)
        hc = rng.standard_normal((B, 1, H)).astype(f)
        masks = (rng.random((B, 1)) > 0.3).astype(f)
        avail = (rng.random((B, A)) > 0.4).astype(f)
        avail[:, 1] = 1.0
        p = f"c{case}"
        out[p + "/spec"] = np.array([int(sp["use_ReLU"]), int(sp["rec"]), D, S, A, B, H])
        out.update(sd_arrays(p + "/actor", pol.actor)); out.update(sd_arrays(p + "/critic", pol.critic))
        out.update({p + "/obs": obs, p + "/share_obs": sobs, p + "/rnn_a": ha, p + "/rnn_c": hc,
                    p + "/masks": masks, p + "/avail": avail})
        with torch.no_grad():
            for tag, av in (("avail", avail), ("noavail", None)):
                v, act, lp, ra, rc = pol.get_actions(sobs, obs, ha, hc, masks, av, deterministic=True)
                out.update({f"{p}/{tag}/values": v.numpy(), f"{p}/{tag}/actions": act.numpy(),
                            f"{p}/{tag}/logp": lp.numpy(), f"{p}/{tag}/rnn_a": ra.numpy(), f"{p}/{tag}/rnn_c": rc.numpy()})
                vv = pol.get_values(sobs, hc, masks)
                out[f"{p}/{tag}/get_values"] = vv.numpy()
                actions_in = act.numpy().astype(f)
                active = (rng.random((B, 1)) > 0.3).astype(f)
                ev, elp, eent = pol.evaluate_actions(sobs, obs, ha, hc, actions_in, masks, av, active)
                out.update({f"{p}/{tag}/active": active, f"{p}/{tag}/eval_values": ev.numpy(),
                            f"{p}/{tag}/eval_logp": elp.numpy(), f"{p}/{tag}/eval_entropy": np.array(eent.item(), dtype=f)})
            if sp["rec"]:
                # L-step chunked sequence (rnn.py:30-77) with zeros inside the chunk
                L, Nc = 10, 6
                obs_s = rng.standard_normal((L * Nc, D)).astype(f)
                sobs_s = rng.standard_normal((L * Nc, S)).astype(f)
                h0a = rng.standard_normal((Nc, 1, H)).astype(f)
                h0c = rng.standard_normal((Nc, 1, H)).astype(f)
                m_s = (rng.random((L * Nc, 1)) > 0.25).astype(f)
                av_s = (rng.random((L * Nc, A)) > 0.4).astype(f); av_s[:, 0] = 1.0
                act_s = np.zeros((L * Nc, 1), dtype=f)
                active_s = (rng.random((L * Nc, 1)) > 0.3).astype(f)
                ev, elp, eent = pol.evaluate_actions(sobs_s, obs_s, h0a, h0c, act_s, m_s, av_s, active_s)
                out.update({p + "/seq/obs": obs_s, p + "/seq/share_obs": sobs_s, p + "/seq/h0a": h0a, p + "/seq/h0c": h0c,
                            p + "/seq/masks": m_s, p + "/seq/avail": av_s, p + "/seq/actions": act_s,
                            p + "/seq/active": active_s, p + "/seq/values": ev.numpy(), p + "/seq/logp": elp.numpy(),
                            p + "/seq/entropy": np.array(eent.item(), dtype=f), p + "/seq/LN": np.array([L, Nc])})
        case += 1
    out["n_cases"] = np.array(case)
    save("forward", **out)


def adam_arrays(prefix, opt, module):
    out = {}
    names = {id(p): n for n, p in module.named_parameters()}
    for p_, st in opt.state.items():
        n = names[id(p_)]
        out[f"{prefix}/{n}/exp_avg"] = st["exp_avg"].numpy().copy()
        out[f"{prefix}/{n}/exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
        out[f"{prefix}/{n}/step"] = np.array(float(st["step"]))
    return out


def gen_ppo_update():
    out = {}
    case = 0
    # hidden_size 64 (the kernels' native width) for the main cases, 16 for the flag variants (fixture size)
    variants = [dict(), dict(use_huber_loss=False, hidden_size=16), dict(use_clipped_value_loss=False, hidden_size=16),
                dict(use_value_active_masks=False, use_policy_active_masks=False, hidden_size=16),
                dict(max_grad_norm=0.05, hidden_size=16), dict(use_max_grad_norm=False, hidden_size=16),
                dict(_update_actor=False, hidden_size=16),
                dict(use_ReLU=False), dict(use_valuenorm=False, hidden_size=16),
                dict(use_recurrent_policy=True, _rec=True),
                dict(use_recurrent_policy=True, _rec=True, use_ReLU=False, hidden_size=16),
                dict(_two_steps=True, hidden_size=16)]
    for var in variants:
        var = dict(var)
        update_actor = var.pop("_update_actor", True)
        rec = var.pop("_rec", False)
        two = var.pop("_two_steps", False)
        torch.manual_seed(600 + case)
        rng = np.random.default_rng(600 + case)
        T, N, M, D, S, A = (20, 2, 3, 10, 30, 5) if rec else (8, 4, 3, 18, 54, 5)
        a = make_args(episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, **var)
        pol = R_MAPPOPolicy(a, [D], [S], Discrete(A))
        with torch.no_grad():
            for net in (pol.actor, pol.critic):
                for n_, p_ in net.named_parameters():
                    if "norm" in n_ or "bias" in n_ or ".2." in n_:
                        p_.add_(0.1 * torch.randn_like(p_))
                    if "action_out" in n_ and "weight" in n_:
                        p_.mul_(30.0)
        tr = R_MAPPO(a, pol)
        if a.use_valuenorm:
            tr.value_normalizer = vn_with_state(2, rng)
        buf = SharedReplayBuffer(a, M, [D], [S], Discrete(A))
        fill_buffer(buf, rng)
        buf.value_preds[...] *= 0.3
        buf.returns[...] = (rng.standard_normal(buf.returns.shape) * 2).astype(np.float32)
        adv = rng.standard_normal(buf.rewards.shape).astype(np.float32)
        torch.manual_seed(2000 + case)
        if rec:
            sample = next(buf.recurrent_generator(adv, 1, a.data_chunk_length))
        else:
            sample = next(buf.feed_forward_generator(adv, 1))
        p = f"c{case}"
        flags = dict(use_huber_loss=a.use_huber_loss, use_clipped_value_loss=a.use_clipped_value_loss,
                     use_value_active_masks=a.use_value_active_masks, use_policy_active_masks=a.use_policy_active_masks,
                     use_max_grad_norm=a.use_max_grad_norm, use_ReLU=a.use_ReLU, use_valuenorm=a.use_valuenorm,
                     use_recurrent_policy=a.use_recurrent_policy, update_actor=update_actor, two_steps=two)
        out[p + "/flags"] = np.array([int(v) for v in flags.values()])
        out[p + "/flag_names"] = np.array(list(flags.keys()))
        out[p + "/hyper"] = np.array([a.clip_param, a.entropy_coef, a.value_loss_coef, a.huber_delta, a.max_grad_norm,
                                      a.lr, a.critic_lr, a.opti_eps, a.weight_decay, a.data_chunk_length], dtype=np.float64)
        out[p + "/dims"] = np.array([T, N, M, D, S, A, a.hidden_size])
        out.update(sd_arrays(p + "/actor0", pol.actor)); out.update(sd_arrays(p + "/critic0", pol.critic))
        out[p + "/vn0"] = vn_state(tr.value_normalizer) if a.use_valuenorm else np.zeros(3, np.float32)
        for nm, arr in zip(TUPLE, sample):
            out[f"{p}/sample/{nm}"] = arr
        for rep in range(2 if two else 1):
            vl, cgn, pl, ent, agn, imp = tr.ppo_update(sample, update_actor)
            q = f"{p}/r{rep}"
            out[q + "/stats"] = np.array([vl.item(), float(cgn), pl.item(), ent.item(), float(agn), imp.mean().item()],
                                         dtype=np.float64)
            out[q + "/imp"] = imp.detach().numpy()
            for tag, net in (("actor", pol.actor), ("critic", pol.critic)):
                for n_, p_ in net.named_parameters():
                    if p_.grad is not None:
                        out[f"{q}/{tag}_grad/{n_}"] = p_.grad.numpy().copy()      # post-clip (what Adam consumed)
            out.update(sd_arrays(q + "/actor", pol.actor)); out.update(sd_arrays(q + "/critic", pol.critic))
            out.update(adam_arrays(q + "/actor_adam", pol.actor_optimizer, pol.actor))
            out.update(adam_arrays(q + "/critic_adam", pol.critic_optimizer, pol.critic))
            out[q + "/vn"] = vn_state(tr.value_normalizer) if a.use_valuenorm else np.zeros(3, np.float32)
        case += 1
    out["n_cases"] = np.array(case)
    save("ppo_update", **out)


def gen_train():
    out = {}
    case = 0
    for rec, nmb in ((False, 2), (True, 1), (False, 1)):
        torch.manual_seed(700 + case)
        rng = np.random.default_rng(700 + case)
        T, N, M, D, S, A = (20, 2, 3, 10, 30, 5) if rec else (8, 4, 3, 18, 54, 5)
        a = make_args(episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2,
                      num_mini_batch=nmb, use_recurrent_policy=rec)
        pol = R_MAPPOPolicy(a, [D], [S], Discrete(A))
        tr = R_MAPPO(a, pol)
        buf = SharedReplayBuffer(a, M, [D], [S], Discrete(A))
        fill_buffer(buf, rng)
        buf.value_preds[...] *= 0.3
        nv = rng.standard_normal((N, M, 1)).astype(np.float32)
        buf.compute_returns(nv, tr.value_normalizer)
        p = f"c{case}"
        out[p + "/dims"] = np.array([T, N, M, D, S, A, a.hidden_size, nmb, int(rec), a.ppo_epoch, a.data_chunk_length])
        out.update(buffer_arrays(p + "/buf", buf))
        out.update(sd_arrays(p + "/actor0", pol.actor)); out.update(sd_arrays(p + "/critic0", pol.critic))
        seed = 3000 + case
        torch.manual_seed(seed)
        n_perm = (T * N * M) // a.data_chunk_length if rec else T * N * M
        perms = [torch.randperm(n_perm).numpy() for _ in range(a.ppo_epoch)]
        torch.manual_seed(seed)
        tr.prep_training()
        info = tr.train(buf)
        out[p + "/perms"] = np.stack(perms)
        out[p + "/info_keys"] = np.array(list(info.keys()))
        out[p + "/info"] = np.array([float(v) for v in info.values()], dtype=np.float64)
        out.update(sd_arrays(p + "/actor1", pol.actor)); out.update(sd_arrays(p + "/critic1", pol.critic))
        out[p + "/vn1"] = vn_state(tr.value_normalizer)
        case += 1
    out["n_cases"] = np.array(case)
    save("train", **out)


def gen_separated():
    """share_policy=False: the reference's SeparatedReplayBuffer (onpolicy/utils/separated_buffer.py:13-393) — one buffer,
    policy and trainer per agent (runner/separated/base_runner.py:70-132).  Slot semantics incl. the step wrap, after_update,
    compute_returns, the feed-forward generator, and R_MAPPO.train on such a buffer."""
    from onpolicy.utils.separated_buffer import SeparatedReplayBuffer
    rng = np.random.default_rng(900)
    f = np.float32
    out = {}
    # (1) insert / after_update
    T, N = 4, 3
    a = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8)
    buf = SeparatedReplayBuffer(a, [6], [18], Discrete(5))
    names = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns", "available_actions", "actions",
             "action_log_probs", "rewards", "masks", "bad_masks", "active_masks")
    arrs = lambda pre, b: {f"{pre}/{n}": getattr(b, n).copy() for n in names}
    for s_ in range(T + 2):
        d = dict(share_obs=rng.standard_normal((N, 18)).astype(f), obs=rng.standard_normal((N, 6)).astype(f),
                 rnn_a=rng.standard_normal((N, 1, 8)).astype(f), rnn_c=rng.standard_normal((N, 1, 8)).astype(f),
                 actions=rng.integers(0, 5, (N, 1)).astype(f), logp=rng.standard_normal((N, 1)).astype(f),
                 values=rng.standard_normal((N, 1)).astype(f), rewards=rng.standard_normal((N, 1)).astype(f),
                 masks=(rng.random((N, 1)) > 0.3).astype(f), bad=(rng.random((N, 1)) > 0.3).astype(f),
                 active=(rng.random((N, 1)) > 0.3).astype(f), avail=(rng.random((N, 5)) > 0.3).astype(f))
        for k, v in d.items():
            out[f"ins/in{s_}/{k}"] = v
        buf.insert(d["share_obs"], d["obs"], d["rnn_a"], d["rnn_c"], d["actions"], d["logp"], d["values"], d["rewards"], d["masks"],
                   d["bad"], d["active"], d["avail"])
        out[f"ins/step_after{s_}"] = np.array(buf.step)
        if s_ == T - 1:
            out.update(arrs("ins/full", buf))
            buf.after_update()
            out.update(arrs("ins/after_update", buf))
    out.update(arrs("ins/final", buf))
    out["ins/n_inserts"] = np.array(T + 2)
    # (2) compute_returns (GAE + ValueNorm, the default flags) and the feed-forward generator, (3) train()
    T, N, D, S, A = 8, 6, 18, 54, 5
    torch.manual_seed(901)
    a = make_args(episode_length=T, n_rollout_threads=N, lr=7e-4, critic_lr=7e-4, ppo_epoch=2, num_mini_batch=2)
    pol = R_MAPPOPolicy(a, [D], [S], Discrete(A))
    tr = R_MAPPO(a, pol)
    buf = SeparatedReplayBuffer(a, [D], [S], Discrete(A))
    for n in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "rewards", "action_log_probs"):
        arr = getattr(buf, n); arr[...] = rng.standard_normal(arr.shape).astype(f)
    buf.action_log_probs[...] = -np.abs(buf.action_log_probs) - 0.5
    buf.value_preds[...] *= 0.3
    buf.actions[...] = rng.integers(0, A, buf.actions.shape).astype(f)
    buf.masks[...] = (rng.random(buf.masks.shape) > 0.15).astype(f)
    buf.active_masks[...] = (rng.random(buf.masks.shape) > 0.2).astype(f)
    nv = rng.standard_normal((N, 1)).astype(f)
    buf.compute_returns(nv, tr.value_normalizer)
    out["tr/dims"] = np.array([T, N, D, S, A, a.ppo_epoch, a.num_mini_batch])
    out["tr/next_value"] = nv
    out.update(arrs("tr/buf", buf))
    adv = (buf.returns[:-1] - tr.value_normalizer.denormalize(buf.value_preds[:-1])).astype(f)
    torch.manual_seed(55)
    rand = torch.randperm(T * N).numpy()
    torch.manual_seed(55)
    batches = list(buf.feed_forward_generator(adv, 2))
    out["gen/rand"] = rand
    out["gen/adv"] = adv
    tuple_names = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns", "masks", "active_masks",
                   "old_action_log_probs", "adv_targ", "available_actions")
    for bi, sample in enumerate(batches):
        for nm, arr in zip(tuple_names, sample):
            out[f"gen/b{bi}/{nm}"] = np.asarray(arr)
    out["gen/n_batches"] = np.array(len(batches))
    out.update(sd_arrays("tr/actor0", pol.actor)); out.update(sd_arrays("tr/critic0", pol.critic))
    torch.manual_seed(3900)
    perms = [torch.randperm(T * N).numpy() for _ in range(a.ppo_epoch)]
    torch.manual_seed(3900)
    tr.prep_training()
    info = tr.train(buf)
    out["tr/perms"] = np.stack(perms)
    out["tr/info_keys"] = np.array(list(info.keys()))
    out["tr/info"] = np.array([float(v) for v in info.values()], dtype=np.float64)
    out.update(sd_arrays("tr/actor1", pol.actor)); out.update(sd_arrays("tr/critic1", pol.critic))
    out["tr/vn1"] = vn_state(tr.value_normalizer)
    save("separated", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ns = ap.parse_args()
    torch.set_num_threads(1)
    gens = dict(valuenorm=gen_valuenorm, gae=gen_gae, advnorm=gen_advnorm, generators=gen_generators,
                insert=gen_insert, forward=gen_forward, ppo_update=gen_ppo_update, train=gen_train, separated=gen_separated)
    for k, fn in gens.items():
        if ns.only in (None, k):
            fn()
