"""Tensor-level wrappers over the C ABI (include/mappo_hip.h).  torch is plumbing here: device memory,
streams, shapes.  Every function enqueues on torch's current stream of the tensors' device and returns
without synchronising."""
import ctypes as C

import torch

from . import _lib
from ._lib import NetDesc, PpoCfg


def _ptr(t, dtype=torch.float32, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("tensor required")
    if not t.is_cuda:
        raise _lib.MappoHipError("mappo_amd ops need tensors in HBM (cuda/hip device); got a CPU tensor — "
                                 "there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def net_desc(in_dim, out_dim, layer_N=1, use_relu=True, use_feature_norm=True, recurrent=False, hidden=64):
    return NetDesc(int(in_dim), int(hidden), int(out_dim), int(layer_N), int(bool(use_relu)),
                   int(bool(use_feature_norm)), int(bool(recurrent)))


def net_param_count(desc):
    return int(_lib.load().mappo_net_param_count(C.byref(desc)))


# ---- K1 -------------------------------------------------------------------------------------------
def insert_mpe(obs, rewards, dones, obs_dst, share_dst, rew_dst, mask_dst, centralized):
    """obs [N, M, D] (innermost stride 1), rewards [N, M(, 1)], dones [N, M] bool — device tensors, arbitrary outer strides."""
    N, M, D = obs.shape
    if rewards.dim() == 3:
        rewards = rewards[..., 0]
    assert obs.is_cuda and obs.dtype == torch.float32 and obs.stride(2) == 1 and rewards.dtype == torch.float32 and dones.dtype == torch.bool
    rc = _lib.load().mappo_insert_mpe(C.c_void_p(obs.data_ptr()), obs.stride(0), obs.stride(1), C.c_void_p(rewards.data_ptr()),
                                      rewards.stride(0), rewards.stride(1), C.c_void_p(dones.data_ptr()), dones.stride(0),
                                      dones.stride(1), _ptr(obs_dst), _ptr(share_dst), _ptr(rew_dst), _ptr(mask_dst), int(N), int(M),
                                      int(D), int(bool(centralized)), _stream())
    _lib.check(rc, "mappo_insert_mpe")


def insert_mpe_rnn(obs, rewards, dones, obs_dst, share_dst, rew_dst, mask_dst, centralized, rnn_states, rnn_states_critic, rnn_dst,
                   rnn_critic_dst):
    """insert_mpe + rnn_dst / rnn_critic_dst = states * (1 - done) in the same launch (contiguous fp32 [N*M, H] state arrays)."""
    N, M, D = obs.shape
    if rewards.dim() == 3:
        rewards = rewards[..., 0]
    assert obs.is_cuda and obs.dtype == torch.float32 and obs.stride(2) == 1 and rewards.dtype == torch.float32 and dones.dtype == torch.bool
    H = rnn_states.numel() // (N * M)
    assert rnn_states.numel() == rnn_states_critic.numel() == rnn_dst.numel() == rnn_critic_dst.numel() == N * M * H
    rc = _lib.load().mappo_insert_mpe_rnn(C.c_void_p(obs.data_ptr()), obs.stride(0), obs.stride(1), C.c_void_p(rewards.data_ptr()),
                                          rewards.stride(0), rewards.stride(1), C.c_void_p(dones.data_ptr()), dones.stride(0),
                                          dones.stride(1), _ptr(obs_dst), _ptr(share_dst), _ptr(rew_dst), _ptr(mask_dst), int(N), int(M),
                                          int(D), int(bool(centralized)), _ptr(rnn_states), _ptr(rnn_states_critic), _ptr(rnn_dst),
                                          _ptr(rnn_critic_dst), int(H), _stream())
    _lib.check(rc, "mappo_insert_mpe_rnn")


def insert_smac(obs, share_obs, avail, rewards, dones, bad, rnn_states, rnn_states_critic, obs_dst, share_dst, avail_dst, rew_dst, mask_dst,
                bad_dst, active_dst, rnn_dst, rnn_critic_dst):
    """SMAC rollout insert in one launch (mappo_insert_smac): contiguous fp32 obs [N, M, D] / share_obs [N, M, S] / avail [N, M, A] (or
    None), rewards [N, M(, 1)] (any strides), dones [N, M] bool, bad [N, M] bool contiguous or None, states [N*M, ., H] or None."""
    N, M, D = obs.shape
    S = share_obs.shape[-1]
    if rewards.dim() == 3:
        rewards = rewards[..., 0]
    A = avail.shape[-1] if avail is not None else 0
    H = rnn_states.numel() // (N * M) if rnn_states is not None else 0
    n = lambda t: _ptr(t, allow_none=True)
    rc = _lib.load().mappo_insert_smac(_ptr(obs), _ptr(share_obs), n(avail), C.c_void_p(rewards.data_ptr()), rewards.stride(0),
                                       rewards.stride(1), C.c_void_p(dones.data_ptr()), dones.stride(0), dones.stride(1),
                                       C.c_void_p(bad.data_ptr()) if bad is not None else None, n(rnn_states), n(rnn_states_critic),
                                       _ptr(obs_dst), _ptr(share_dst), n(avail_dst), _ptr(rew_dst), _ptr(mask_dst), _ptr(bad_dst),
                                       _ptr(active_dst), n(rnn_dst), n(rnn_critic_dst), int(N), int(M), int(D), int(S), int(A), int(H),
                                       _stream())
    _lib.check(rc, "mappo_insert_smac")


def recurrent_rollout_step(actor_params, actor_desc, critic_params, critic_desc, obs, share_obs, avail, rewards, dones, bad, actor_h, critic_h,
                           actor_h_next, critic_h_next, deterministic, seed, counter, counter_dev, actions, logp, values, slot):
    """mappo_recurrent_rollout_step: SMAC insert of the pending env output (obs [N, M, D], share_obs [N, M, S], avail [N, M, A] or None,
    rewards [N, M(, 1)], dones [N, M] bool, bad [N, M] bool or None; states [N*M, ., 64]) into the arrays of `slot` (dict: obs, share_obs,
    available_actions, rewards, masks, bad_masks, active_masks, rnn_states, rnn_states_critic) + get_actions / get_values on it."""
    N, M, _ = obs.shape
    if rewards.dim() == 3:
        rewards = rewards[..., 0]
    sl = _lib.SmacSlot(*[(slot[k].data_ptr() if slot.get(k) is not None else None) for k, _ in _lib.SmacSlot._fields_])
    for k, _ in _lib.SmacSlot._fields_:
        t = slot.get(k)
        if t is not None and not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise ValueError(f"slot[{k}] must be a contiguous fp32 device tensor")
    n = lambda t: _ptr(t, allow_none=True)
    rc = _lib.load().mappo_recurrent_rollout_step(
        _ptr(actor_params), C.byref(actor_desc), _ptr(critic_params), C.byref(critic_desc), _ptr(obs), _ptr(share_obs), n(avail),
        C.c_void_p(rewards.data_ptr()), rewards.stride(0), rewards.stride(1), C.c_void_p(dones.data_ptr()), dones.stride(0), dones.stride(1),
        C.c_void_p(bad.data_ptr()) if bad is not None else None, _ptr(actor_h), _ptr(critic_h), _ptr(actor_h_next), _ptr(critic_h_next),
        int(N), int(M), int(bool(deterministic)), int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1),
        _ptr(counter_dev, torch.int64, allow_none=True), _ptr(actions), _ptr(logp), _ptr(values), C.byref(sl), _stream())
    _lib.check(rc, "mappo_recurrent_rollout_step")


def recurrent_rows(perm, L, T, R, num_mini_batch):
    """perm [E, data_chunks] int64 (device) -> rows [E, nmb, L*mbs], h0_rows [E, nmb, mbs] int32 (mappo_recurrent_rows)."""
    E, chunks = perm.shape
    mbs = chunks // num_mini_batch
    rows = torch.empty(E, num_mini_batch, L * mbs, dtype=torch.int32, device=perm.device)
    h0 = torch.empty(E, num_mini_batch, mbs, dtype=torch.int32, device=perm.device)
    rc = _lib.load().mappo_recurrent_rows(_ptr(perm, torch.int64), int(E), int(chunks), int(L), int(T), int(R), int(num_mini_batch),
                                          _ptr(rows, torch.int32), _ptr(h0, torch.int32), _stream())
    _lib.check(rc, "mappo_recurrent_rows")
    return rows, h0


def copy_batch(pairs):
    """[(dst, src), ...] contiguous fp32 device tensors of equal numel per pair, copied in ONE launch (<= 16 pairs)."""
    n = len(pairs)
    dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    src = (C.c_void_p * n)(*[s.data_ptr() for _, s in pairs])
    cnt = (C.c_int64 * n)(*[d.numel() for d, _ in pairs])
    for d, s_ in pairs:
        assert d.is_cuda and s_.is_cuda and d.dtype == torch.float32 and s_.dtype == torch.float32 and d.is_contiguous() \
            and s_.is_contiguous() and d.numel() == s_.numel()
    rc = _lib.load().mappo_copy_batch(n, dst, src, cnt, _stream())
    _lib.check(rc, "mappo_copy_batch")


# ---- K2 -------------------------------------------------------------------------------------------
def gae_scan(rewards, value_preds, next_value, masks, bad_masks, returns, vn_state, gamma, gae_lambda,
             use_gae=True, use_proper_time_limits=False):
    """compute_returns (shared_buffer.py:168-224) in place on [T(+1), R] views of the buffer arrays."""
    T = rewards.shape[0]
    R = rewards.numel() // T
    lib = _lib.load()
    rc = lib.mappo_gae_scan(_ptr(rewards), _ptr(value_preds), _ptr(next_value), _ptr(masks),
                            _ptr(bad_masks, allow_none=True), _ptr(returns), _ptr(vn_state, allow_none=True),
                            T, R, float(gamma), float(gae_lambda), int(bool(use_gae)),
                            int(bool(use_proper_time_limits)), _stream())
    _lib.check(rc, "mappo_gae_scan")


# ---- K3 -------------------------------------------------------------------------------------------
def adv_moments(returns, value_preds, active_masks, vn_state, adv, moments, workspace=None):
    lib = _lib.load()
    n = adv.numel()
    if workspace is None:
        workspace = _ws(lib.mappo_adv_workspace_bytes(n), adv.device)
    rc = lib.mappo_adv_moments(_ptr(returns), _ptr(value_preds), _ptr(active_masks), _ptr(vn_state, allow_none=True),
                               _ptr(adv), _ptr(moments, torch.float64), _ptr(workspace, torch.uint8), n, _stream())
    _lib.check(rc, "mappo_adv_moments")


def adv_normalize(adv, moments):
    rc = _lib.load().mappo_adv_normalize(_ptr(adv), _ptr(moments, torch.float64), adv.numel(), _stream())
    _lib.check(rc, "mappo_adv_normalize")


# ---- K12 ------------------------------------------------------------------------------------------
def minibatch_moments(returns, active_masks, rows, B, mb_moments, workspace=None):
    lib = _lib.load()
    if workspace is None:
        workspace = _ws(lib.mappo_moments_workspace_bytes(B), returns.device)
    rc = lib.mappo_minibatch_moments(_ptr(returns), _ptr(active_masks), _ptr(rows, torch.int32, allow_none=True), int(B),
                                     _ptr(mb_moments, torch.float64), _ptr(workspace, torch.uint8), _stream())
    _lib.check(rc, "mappo_minibatch_moments")


def valuenorm_update(vn_state, mb_moments, beta=0.99999):
    rc = _lib.load().mappo_valuenorm_update(_ptr(vn_state), _ptr(mb_moments, torch.float64), float(beta), _stream())
    _lib.check(rc, "mappo_valuenorm_update")


def valuenorm_update_n(vn_state, mb_moments, beta, n, states_out):
    """n ValueNorm.update calls with the same batch moments in one launch; states_out [n, 3]."""
    rc = _lib.load().mappo_valuenorm_update_n(_ptr(vn_state), _ptr(mb_moments, torch.float64), float(beta), int(n), _ptr(states_out),
                                              _stream())
    _lib.check(rc, "mappo_valuenorm_update_n")


# ---- K5 -------------------------------------------------------------------------------------------
def ppo_cfg(args, accumulate_partials=False):
    return PpoCfg(float(args.clip_param), float(args.entropy_coef), float(args.value_loss_coef), float(args.huber_delta),
                  int(bool(args.use_huber_loss)), int(bool(args.use_clipped_value_loss)),
                  int(bool(args.use_policy_active_masks)), int(bool(args.use_value_active_masks)),
                  int(bool(args.use_valuenorm)), int(bool(accumulate_partials)))


def ppo_loss_fwd_bwd(logits, values, rows, avail, actions, old_logp, adv, active, v_old, returns, vn_state, mb_moments,
                     dlogits, dvalues, stats, cfg, workspace=None):
    lib = _lib.load()
    B, A = logits.shape
    if workspace is None:
        workspace = _ws(lib.mappo_ppo_loss_workspace_bytes(B), logits.device)
    rc = lib.mappo_ppo_loss_fwd_bwd(_ptr(logits), _ptr(values), _ptr(rows, torch.int32, allow_none=True),
                                    _ptr(avail, allow_none=True), _ptr(actions), _ptr(old_logp), _ptr(adv), _ptr(active),
                                    _ptr(v_old), _ptr(returns), _ptr(vn_state, allow_none=True),
                                    _ptr(mb_moments, torch.float64), _ptr(dlogits), _ptr(dvalues),
                                    _ptr(stats, torch.float64), _ptr(workspace, torch.uint8), C.byref(cfg), int(B), int(A),
                                    _stream())
    _lib.check(rc, "mappo_ppo_loss_fwd_bwd")


# ---- K7 / K8 --------------------------------------------------------------------------------------
def mlp_forward(params, desc, x, rows, B, out):
    rc = _lib.load().mappo_mlp_forward(_ptr(params), C.byref(desc), _ptr(x), _ptr(rows, torch.int32, allow_none=True),
                                       int(B), _ptr(out), _stream())
    _lib.check(rc, "mappo_mlp_forward")


def actor_act(params, desc, obs, avail, B, deterministic, seed, counter, actions, logp, counter_dev=None):
    rc = _lib.load().mappo_actor_act(_ptr(params), C.byref(desc), _ptr(obs), _ptr(avail, allow_none=True), int(B),
                                     int(bool(deterministic)), int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1),
                                     _ptr(counter_dev, torch.int64, allow_none=True), _ptr(actions), _ptr(logp), _stream())
    _lib.check(rc, "mappo_actor_act")


def rollout_step(actor_params, actor_desc, critic_params, critic_desc, obs, share_obs, M, B, avail, deterministic, seed, counter,
                 counter_dev, actions, logp, values, insert=None):
    """Fused rollout step.  obs / share_obs: (tensor, stride_n, stride_m) row sources (strides in elements; M == 0 means
    contiguous [B][D] and the strides are ignored).  insert: None or dict(obs_dst, share_dst, rewards=(tensor, sn, sm),
    dones=(tensor, sn, sm), rew_dst, mask_dst, centralized) — the env output `obs` comes from is copied into those slots."""
    ot, osn, osm = obs
    st, ssn, ssm = share_obs
    if insert is None:
        ins = (None, None, None, 0, 0, None, 0, 0, None, None, 0)
    else:
        rt, rsn, rsm = insert["rewards"]
        dt, dsn, dsm = insert["dones"]
        ins = (C.c_void_p(insert["obs_dst"].data_ptr()), C.c_void_p(insert["share_dst"].data_ptr()), C.c_void_p(rt.data_ptr()), int(rsn), int(rsm),
               C.c_void_p(dt.data_ptr()), int(dsn), int(dsm), C.c_void_p(insert["rew_dst"].data_ptr()), C.c_void_p(insert["mask_dst"].data_ptr()),
               int(bool(insert["centralized"])))
    rc = _lib.load().mappo_rollout_step(_ptr(actor_params), C.byref(actor_desc), _ptr(critic_params), C.byref(critic_desc),
                                        C.c_void_p(ot.data_ptr()), int(osn), int(osm), C.c_void_p(st.data_ptr()), int(ssn), int(ssm), int(M), int(B),
                                        _ptr(avail, allow_none=True), int(bool(deterministic)), int(seed) & (2 ** 64 - 1),
                                        int(counter) & (2 ** 64 - 1), _ptr(counter_dev, torch.int64, allow_none=True),
                                        _ptr(actions, allow_none=True), _ptr(logp, allow_none=True), _ptr(values), *ins, _stream())
    _lib.check(rc, "mappo_rollout_step")


def mlp_backward_slabs(B):
    return int(_lib.load().mappo_mlp_backward_slabs(int(B)))


PRODUCER_MLP_BACKWARD, PRODUCER_ACTOR_UPDATE, PRODUCER_CRITIC_UPDATE, PRODUCER_TRUNK_BACKWARD = 0, 1, 2, 3


def wide_layout(desc, producer):
    """Layout id the `producer` entry point leaves in a wide workspace for `desc` (a pure function: mappo_wide_layout)."""
    v = int(_lib.load().mappo_wide_layout(C.byref(desc), int(producer)))
    if v < 0:
        _lib.check(v, "mappo_wide_layout")
    return v


def _wide(desc, x, rows, B, slabs, slab_stride, slab_col0, params, wide_ws, producer):
    """Second launch of a wide-input (in_dim > 64) backward: W1 + feature-norm gradient columns."""
    if wide_ws is not None:
        rc = _lib.load().mappo_wide_l1_backward(_ptr(params), C.byref(desc), _ptr(x), _ptr(rows, torch.int32, allow_none=True), int(B),
                                                _ptr(wide_ws), _ptr(slabs), int(slab_stride), int(slab_col0),
                                                wide_layout(desc, producer), _stream())
        _lib.check(rc, "mappo_wide_l1_backward")


_wide_ws_cache = {}


def wide_workspace(desc, B, device):
    """Scratch of the wide-input path (None for in_dim <= 64), cached per (B, device, stream): two networks' backward
    passes may run on two streams at the same time (recurrent training) and must not share it."""
    if desc.in_dim <= 64:
        return None
    key = (int(B), str(device), int(torch.cuda.current_stream().cuda_stream))
    ws = _wide_ws_cache.get(key)
    if ws is None:
        ws = torch.empty(int(_lib.load().mappo_wide_workspace_floats(int(B))), dtype=torch.float32, device=device)
        _wide_ws_cache[key] = ws
    return ws


def wide_l1_slabs(B):
    return int(_lib.load().mappo_wide_l1_slabs(int(B)))


def mlp_backward(params, desc, x, rows, B, dout, slabs, slab_stride, slab_col0):
    ws = wide_workspace(desc, B, x.device)
    rc = _lib.load().mappo_mlp_backward(_ptr(params), C.byref(desc), _ptr(x), _ptr(rows, torch.int32, allow_none=True),
                                        int(B), _ptr(dout), _ptr(slabs), int(slab_stride), int(slab_col0),
                                        _ptr(ws, allow_none=True), _stream())
    _lib.check(rc, "mappo_mlp_backward")
    _wide(desc, x, rows, B, slabs, slab_stride, slab_col0, params, ws, PRODUCER_MLP_BACKWARD)


# ---- fused update kernels ------------------------------------------------------------------------
def update_partials(device):
    return torch.zeros(int(_lib.load().mappo_update_partials_bytes()) // 8, dtype=torch.float64, device=device)


def actor_update(params, desc, obs, rows, B, avail, actions, old_logp, adv, active, mb_moments, cfg, slabs, slab_stride,
                 slab_col0, partials, n_blocks=0):
    ws = wide_workspace(desc, B, obs.device)
    rc = _lib.load().mappo_actor_update(_ptr(params), C.byref(desc), _ptr(obs), _ptr(rows, torch.int32, allow_none=True), int(B),
                                        _ptr(avail, allow_none=True), _ptr(actions), _ptr(old_logp), _ptr(adv), _ptr(active),
                                        _ptr(mb_moments, torch.float64), C.byref(cfg), _ptr(slabs), int(slab_stride),
                                        int(slab_col0), _ptr(partials, torch.float64), _ptr(ws, allow_none=True), int(n_blocks),
                                        _stream())
    _lib.check(rc, "mappo_actor_update")
    _wide(desc, obs, rows, B, slabs, slab_stride, slab_col0, params, ws, PRODUCER_ACTOR_UPDATE)


def critic_update(params, desc, share_obs, rows, B, v_old, returns, active, vn_state, mb_moments, cfg, slabs, slab_stride,
                  slab_col0, partials, n_blocks=0):
    ws = wide_workspace(desc, B, share_obs.device)
    rc = _lib.load().mappo_critic_update(_ptr(params), C.byref(desc), _ptr(share_obs), _ptr(rows, torch.int32, allow_none=True),
                                         int(B), _ptr(v_old), _ptr(returns), _ptr(active), _ptr(vn_state, allow_none=True),
                                         _ptr(mb_moments, torch.float64), C.byref(cfg), _ptr(slabs), int(slab_stride),
                                         int(slab_col0), _ptr(partials, torch.float64), _ptr(ws, allow_none=True), int(n_blocks),
                                         _stream())
    _lib.check(rc, "mappo_critic_update")
    _wide(desc, share_obs, rows, B, slabs, slab_stride, slab_col0, params, ws, PRODUCER_CRITIC_UPDATE)


def dual_update_slabs(actor_desc, critic_desc, B):
    return int(_lib.load().mappo_dual_update_slabs(C.byref(actor_desc), C.byref(critic_desc), int(B)))


def actor_critic_update(actor_params, actor_desc, obs, critic_params, critic_desc, share_obs, rows, B, avail, actions, old_logp, adv,
                        active, v_old, returns, vn_state, mb_moments, cfg, slabs, slab_stride, actor_col0, critic_col0,
                        actor_partials, critic_partials):
    """mappo_actor_update + mappo_critic_update in one launch (both in_dim <= 64): each network on half the CUs, writing
    dual_update_slabs(actor_desc, critic_desc, B) slab rows / partial rows."""
    rc = _lib.load().mappo_actor_critic_update(_ptr(actor_params), C.byref(actor_desc), _ptr(obs), _ptr(critic_params), C.byref(critic_desc),
                                               _ptr(share_obs), _ptr(rows, torch.int32, allow_none=True), int(B),
                                               _ptr(avail, allow_none=True), _ptr(actions), _ptr(old_logp), _ptr(adv), _ptr(active),
                                               _ptr(v_old), _ptr(returns), _ptr(vn_state, allow_none=True),
                                               _ptr(mb_moments, torch.float64), C.byref(cfg), _ptr(slabs), int(slab_stride),
                                               int(actor_col0), int(critic_col0), _ptr(actor_partials, torch.float64),
                                               _ptr(critic_partials, torch.float64), _stream())
    _lib.check(rc, "mappo_actor_critic_update")


def update_stats(actor_partials, n_actor, critic_partials, n_critic, mb_moments, cfg, stats, acc=None):
    rc = _lib.load().mappo_update_stats(_ptr(actor_partials, torch.float64, allow_none=True), int(n_actor),
                                        _ptr(critic_partials, torch.float64), int(n_critic), _ptr(mb_moments, torch.float64),
                                        C.byref(cfg), _ptr(stats, torch.float64), _ptr(acc, torch.float64, allow_none=True), _stream())
    _lib.check(rc, "mappo_update_stats")


# ---- K9: recurrent layer -------------------------------------------------------------------------
def mlp_features(params, desc, x, rows, B, featT):
    rc = _lib.load().mappo_mlp_features(_ptr(params), C.byref(desc), _ptr(x), _ptr(rows, torch.int32, allow_none=True), int(B),
                                        _ptr(featT), _stream())
    _lib.check(rc, "mappo_mlp_features")


def gru_forward(params, desc, featT, h0, h0_rows, masks, rows, L, Nc, h_last=None, head_mode=0, out=None,
                avail=None, deterministic=False, seed=0, counter=0, counter_dev=None, actions=None, logp=None):
    rc = _lib.load().mappo_gru_forward(_ptr(params), C.byref(desc), _ptr(featT), _ptr(h0),
                                       _ptr(h0_rows, torch.int32, allow_none=True),
                                       _ptr(masks), _ptr(rows, torch.int32, allow_none=True), int(L), int(Nc),
                                       _ptr(h_last, allow_none=True), int(head_mode),
                                       _ptr(out, allow_none=True), _ptr(avail, allow_none=True), int(bool(deterministic)),
                                       int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1),
                                       _ptr(counter_dev, torch.int64, allow_none=True), _ptr(actions, allow_none=True),
                                       _ptr(logp, allow_none=True), _stream())
    _lib.check(rc, "mappo_gru_forward")


def gru_step_dual(actor_params, actor_desc, actor_featT, actor_h0, actor_h_last, critic_params, critic_desc, critic_featT, critic_h0,
                  critic_h_last, masks, Nc, avail, deterministic, seed, counter, counter_dev, actions, logp, values):
    """One rollout step of a recurrent actor and critic in ONE launch (mappo_gru_step_dual)."""
    rc = _lib.load().mappo_gru_step_dual(_ptr(actor_params), C.byref(actor_desc), _ptr(actor_featT), _ptr(actor_h0), _ptr(actor_h_last),
                                         _ptr(critic_params), C.byref(critic_desc), _ptr(critic_featT), _ptr(critic_h0),
                                         _ptr(critic_h_last), _ptr(masks), int(Nc), _ptr(avail, allow_none=True), int(bool(deterministic)),
                                         int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1),
                                         _ptr(counter_dev, torch.int64, allow_none=True), _ptr(actions), _ptr(logp), _ptr(values), _stream())
    _lib.check(rc, "mappo_gru_step_dual")


def recurrent_step_dual(actor_params, actor_desc, actor_obs, actor_h0, actor_h_last, critic_params, critic_desc, critic_obs, critic_h0,
                  critic_h_last, masks, Nc, avail, deterministic, seed, counter, counter_dev, actions, logp, values):
    """One rollout step of a recurrent actor and critic in ONE launch (mappo_recurrent_step_dual)."""
    rc = _lib.load().mappo_recurrent_step_dual(_ptr(actor_params), C.byref(actor_desc), _ptr(actor_obs), _ptr(actor_h0), _ptr(actor_h_last),
                                         _ptr(critic_params), C.byref(critic_desc), _ptr(critic_obs), _ptr(critic_h0),
                                         _ptr(critic_h_last), _ptr(masks), int(Nc), _ptr(avail, allow_none=True), int(bool(deterministic)),
                                         int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1),
                                         _ptr(counter_dev, torch.int64, allow_none=True), _ptr(actions), _ptr(logp), _ptr(values), _stream())
    _lib.check(rc, "mappo_recurrent_step_dual")


def mlp_features_dual(params_a, desc_a, x_a, featT_a, params_c, desc_c, x_c, featT_c, B):
    """Trunk features of two networks on the same B rows in one launch (in_dim <= 64, same layer_N / activation)."""
    rc = _lib.load().mappo_mlp_features_dual(_ptr(params_a), C.byref(desc_a), _ptr(x_a), _ptr(featT_a), _ptr(params_c), C.byref(desc_c),
                                             _ptr(x_c), _ptr(featT_c), int(B), _stream())
    _lib.check(rc, "mappo_mlp_features_dual")


def trunk_backward(params, desc, x, rows, B, dxT, slabs, slab_stride, slab_col0):
    ws = wide_workspace(desc, B, x.device)
    rc = _lib.load().mappo_trunk_backward(_ptr(params), C.byref(desc), _ptr(x), _ptr(rows, torch.int32, allow_none=True), int(B),
                                          _ptr(dxT), _ptr(slabs), int(slab_stride), int(slab_col0), _ptr(ws, allow_none=True), _stream())
    _lib.check(rc, "mappo_trunk_backward")
    _wide(desc, x, rows, B, slabs, slab_stride, slab_col0, params, ws, PRODUCER_TRUNK_BACKWARD)


# ---- K9 training pass on 16-sequence tiles (gru_train16.hip) ---------------------------------------
def gru16_scratch_floats(L, Nc):
    return int(_lib.load().mappo_gru16_scratch_floats(int(L), int(Nc)))


def gru16_blocked_floats(L, Nc):
    return int(_lib.load().mappo_gru16_blocked_floats(int(L), int(Nc)))


def gru16_slabs(L, Nc):
    return int(_lib.load().mappo_gru16_slabs(int(L), int(Nc)))


def mlp_features_seq(params, desc, x, rows, L, Nc, out_blocked):
    rc = _lib.load().mappo_mlp_features_seq(_ptr(params), C.byref(desc), _ptr(x), _ptr(rows, torch.int32, allow_none=True), int(L), int(Nc),
                                            _ptr(out_blocked), _stream())
    _lib.check(rc, "mappo_mlp_features_seq")


def gru16_forward_loss(params, desc, x, x_blocked, h0, h0_rows, masks, rows, L, Nc, head, avail, actions, old_logp, adv, active, v_old,
                       returns, vn_state, mb_moments, cfg, scratch, slabs, slab_stride, slab_col0, partials):
    o = lambda t: _ptr(t, allow_none=True)
    rc = _lib.load().mappo_gru16_forward_loss(_ptr(params), C.byref(desc), _ptr(x), int(bool(x_blocked)), _ptr(h0),
                                              _ptr(h0_rows, torch.int32, allow_none=True), _ptr(masks),
                                              _ptr(rows, torch.int32, allow_none=True), int(L), int(Nc), int(head), o(avail), o(actions),
                                              o(old_logp), o(adv), _ptr(active), o(v_old), o(returns), o(vn_state),
                                              _ptr(mb_moments, torch.float64), C.byref(cfg), _ptr(scratch), _ptr(slabs), int(slab_stride),
                                              int(slab_col0), _ptr(partials, torch.float64), _stream())
    _lib.check(rc, "mappo_gru16_forward_loss")


def gru16_backward(params, desc, masks, rows, L, Nc, scratch, dxT=None):
    rc = _lib.load().mappo_gru16_backward(_ptr(params), C.byref(desc), _ptr(masks), _ptr(rows, torch.int32, allow_none=True), int(L), int(Nc),
                                          _ptr(scratch), _ptr(dxT, allow_none=True), _stream())
    _lib.check(rc, "mappo_gru16_backward")


def gru16_wgrad(desc, x, x_blocked, scratch, L, Nc, slabs, slab_stride, slab_col0):
    rc = _lib.load().mappo_gru16_wgrad(C.byref(desc), _ptr(x), int(bool(x_blocked)), _ptr(scratch), int(L), int(Nc), _ptr(slabs),
                                       int(slab_stride), int(slab_col0), _stream())
    _lib.check(rc, "mappo_gru16_wgrad")


def trunk_backward_seq(params, desc, x, rows, L, Nc, dx_blocked, slabs, slab_stride, slab_col0):
    ws = wide_workspace(desc, L * Nc, x.device)
    rc = _lib.load().mappo_trunk_backward_seq(_ptr(params), C.byref(desc), _ptr(x), _ptr(rows, torch.int32, allow_none=True), int(L), int(Nc),
                                              _ptr(dx_blocked), _ptr(slabs), int(slab_stride), int(slab_col0), _ptr(ws, allow_none=True),
                                              _stream())
    _lib.check(rc, "mappo_trunk_backward_seq")
    _wide(desc, x, rows, L * Nc, slabs, slab_stride, slab_col0, params, ws, PRODUCER_TRUNK_BACKWARD)


# ---- K10 / K11 ------------------------------------------------------------------------------------
def slab_reduce(slabs, n_slabs, slab_stride, P, grad):
    rc = _lib.load().mappo_slab_reduce(_ptr(slabs), int(n_slabs), int(slab_stride), int(P), _ptr(grad), _stream())
    _lib.check(rc, "mappo_slab_reduce")


def optim_workspace(P, device):
    return _ws(_lib.load().mappo_optim_workspace_bytes(int(P)), device)


def clip_adam(params, grad, exp_avg, exp_avg_sq, seg_bounds, opt_hyper, opt_step, grad_norms, workspace, norm_acc=None):
    n_seg = len(seg_bounds) - 1
    arr = (C.c_int64 * (n_seg + 1))(*[int(b) for b in seg_bounds])
    rc = _lib.load().mappo_clip_adam(_ptr(params), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), arr, n_seg,
                                     _ptr(opt_hyper), _ptr(opt_step, torch.int32), _ptr(grad_norms),
                                     _ptr(norm_acc, torch.float64, allow_none=True), _ptr(workspace, torch.uint8), _stream())
    _lib.check(rc, "mappo_clip_adam")


def reduce_clip_adam(slabs, n_slabs, slab_stride, params, grad, exp_avg, exp_avg_sq, seg_bounds, opt_hyper, opt_step, grad_norms,
                     workspace, norm_acc=None):
    """slab_reduce + clip_adam in two launches (single-process training)."""
    n_seg = len(seg_bounds) - 1
    arr = (C.c_int64 * (n_seg + 1))(*[int(b) for b in seg_bounds])
    rc = _lib.load().mappo_reduce_clip_adam(_ptr(slabs), int(n_slabs), int(slab_stride), _ptr(params), _ptr(grad), _ptr(exp_avg),
                                            _ptr(exp_avg_sq), arr, n_seg, _ptr(opt_hyper), _ptr(opt_step, torch.int32), _ptr(grad_norms),
                                            _ptr(norm_acc, torch.float64, allow_none=True), _ptr(workspace, torch.uint8), _stream())
    _lib.check(rc, "mappo_reduce_clip_adam")


def selftest_mfma(A, Bm, D):
    rc = _lib.load().mappo_selftest_mfma(_ptr(A), _ptr(Bm), _ptr(D), _stream())
    _lib.check(rc, "mappo_selftest_mfma")


# ---- measurement hook -----------------------------------------------------------------------------
PROF_IDS = dict(gae=0, ppo_loss=1, mlp_fwd=2, mlp_bwd=3, slab_reduce=4, adam=5, act=6)


def profile_arm(kernel, start_event, stop_event):
    """Bracket the next launch of `kernel`'s dominant HIP kernel with two torch.cuda.Event(enable_timing=True)
    (recorded by the C side on the stream that launch uses).  One-shot."""
    for ev in (start_event, stop_event):
        if ev.cuda_event == 0:                 # torch creates the hipEvent lazily on first record
            ev.record()
    rc = _lib.load().mappo_profile_arm(PROF_IDS[kernel], C.c_void_p(start_event.cuda_event), C.c_void_p(stop_event.cuda_event))
    _lib.check(rc, "mappo_profile_arm")


# ---- GPU-vectorised MPE simple_spread (csrc/mpe_env.hip) -------------------------------------------------------------
def mpe_spread_reset(agent_pos, agent_vel, landmark_pos, tstep, episode, obs, N, M, L, seed):
    f64 = torch.float64
    rc = _lib.load().mappo_mpe_spread_reset(_ptr(agent_pos, f64), _ptr(agent_vel, f64), _ptr(landmark_pos, f64), _ptr(tstep, torch.int32),
                                            _ptr(episode, torch.int64), _ptr(obs),
                                            int(N), int(M), int(L), int(seed) & (2 ** 64 - 1), _stream())
    _lib.check(rc, "mappo_mpe_spread_reset")


def mpe_spread_step(agent_pos, agent_vel, landmark_pos, tstep, episode, actions, action_mode, obs, rewards, dones, N, M, L,
                    episode_length, seed):
    f64 = torch.float64
    rc = _lib.load().mappo_mpe_spread_step(_ptr(agent_pos, f64), _ptr(agent_vel, f64), _ptr(landmark_pos, f64), _ptr(tstep, torch.int32),
                                           _ptr(episode, torch.int64), _ptr(actions), int(action_mode), _ptr(obs), _ptr(rewards),
                                           _ptr(dones, torch.uint8), int(N), int(M), int(L),
                                           int(episode_length), int(seed) & (2 ** 64 - 1), _stream())
    _lib.check(rc, "mappo_mpe_spread_step")


def synth_smac_pool(obs, share_obs, avail, rewards, dead, dones, p_death, p_term, seed, counter):
    """P steps of the synthetic SMAC-shaped env in one launch: pools obs [P, N, M, D], share_obs, avail, rewards [P, N], dones [P, N, M]."""
    P, N, M, D = obs.shape
    assert counter.numel() == 34
    rc = _lib.load().mappo_synth_smac_pool(_ptr(obs), _ptr(share_obs), _ptr(avail), _ptr(rewards), _ptr(dead, torch.bool), _ptr(dones, torch.bool),
                                           int(P), int(N), int(M), int(D), int(share_obs.shape[3]), int(avail.shape[3]), float(p_death),
                                           float(p_term), int(seed), _ptr(counter, torch.int64), _stream())
    _lib.check(rc, "mappo_synth_smac_pool")


def synth_smac_step(obs, share_obs, avail, rewards, dead, dones, p_death, p_term, seed, counter):
    """One step of the synthetic SMAC-shaped env (bench utility, csrc/synth_env.hip); `counter`: int64 [34] device tensor
    {Philox counter, 33 tickets (0)}: the launch advances the counter itself."""
    assert counter.numel() == 34
    N, M, D = obs.shape
    rc = _lib.load().mappo_synth_smac_step(_ptr(obs), _ptr(share_obs), _ptr(avail), _ptr(rewards), _ptr(dead, torch.bool), _ptr(dones, torch.bool),
                                           int(N), int(M), int(D), int(share_obs.shape[2]), int(avail.shape[2]), float(p_death), float(p_term),
                                           int(seed), _ptr(counter, torch.int64), _stream())
    _lib.check(rc, "mappo_synth_smac_step")
