"""GRU (recurrent policy) path — `onpolicy/algorithms/utils/rnn.py:7-80`, the chunked / whole-episode generators
(`shared_buffer.py:288-494`) and the recurrent branch of `R_MAPPO.train` (`r_mappo.py:194-200`).

A recurrent network is trunk -> GRU -> LayerNorm -> head.  Forward outside the training pass: mappo_mlp_features (trunk,
feature-major [64][B]) + mappo_gru_forward, or the one-launch rollout step of both networks (mappo_recurrent_step_dual).
Training, per network and minibatch of `mbs` chunks x L steps (time-major, the order of the reference's stacked chunks), on
16-sequence tiles (csrc/gru_train16.hip): trunk features (blocked) -> gru16_forward_loss (input + hidden products, gates, rnn.norm,
head, loss and head backward in one kernel; stores the gate values and d h) -> gru16_backward (reverse time, d gates in place, d x) ->
gru16_wgrad -> trunk backward, all writing gradient slabs over the joint flat layout.
Row indices come from SharedReplayBuffer.recurrent_rows / naive_recurrent_rows (the reference's index arithmetic,
including chunks that straddle two series when T % L != 0)."""
import numpy as np
import os

import torch

from mappo_amd import ops
from mappo_amd.utils.util import to_device_f32

H = 64


class _Scratch:
    """Feature-major work arrays of one network pass, cached per (L, Nc)."""

    def __init__(self):
        self._c = {}

    def get(self, device, L, Nc, training, tag=0):
        """`tag` separates the networks: the actor's and the critic's passes run on two streams at the same time."""
        key = (L, Nc, training, tag)
        s = self._c.get(key)
        if s is None:
            s = dict(featT=torch.empty(H, L * Nc, dtype=torch.float32, device=device))
            self._c[key] = s
        return s

    def get16(self, device, L, Nc, tag, narrow):
        """Work arrays of the 16-sequence-tile training kernels (gru_train16.hip): the blocked scratch [6][L][n_ct][4][256], the
        trunk features (blocked for in_dim <= 64, feature-major otherwise) and, for wide inputs, the feature-major d x."""
        key = ("g16", L, Nc, tag, narrow)
        s = self._c.get(key)
        if s is None:
            f = lambda n: torch.empty(int(n), dtype=torch.float32, device=device)
            comp = ops.gru16_blocked_floats(L, Nc)
            scratch = f(ops.gru16_scratch_floats(L, Nc))
            s = dict(scratch=scratch, dx=scratch[5 * comp:6 * comp])
            if narrow:
                s["feat"] = f(comp)
            else:
                s["feat"] = torch.empty(H, L * Nc, dtype=torch.float32, device=device)
                s["dxT"] = torch.empty(H, L * Nc, dtype=torch.float32, device=device)
            self._c[key] = s
        return s


_scratch = _Scratch()
_WIDE_BLOCKED = os.environ.get("MAPPO_WIDE_BLOCKED", "1") != "0"             # A/B switch
_TWO_STREAMS = os.environ.get("MAPPO_REC_TWO_STREAMS", "1") != "0"   # the two networks' chains on two streams (A/B: one after the other)


def _seq_shape(n_rows, rnn_states):
    Nc = rnn_states.shape[0]
    assert n_rows % Nc == 0, "rows must be L * (number of rnn states)"
    return n_rows // Nc, Nc


# ---- forward entry points used by R_Actor / R_Critic --------------------------------------------------------------
def actor_step(actor, obs, rnn_states, masks, avail, deterministic, actions_f, logp, counter=None):
    """R_Actor.forward for a recurrent actor (r_actor_critic.py:43-70): one GRU step per row (rollout / act)."""
    B = obs.shape[0]
    L, Nc = _seq_shape(B, rnn_states)
    if L != 1:
        raise NotImplementedError("R_Actor.forward samples one step per row (the reference does the same: rnn.py:25-29)")
    s = _scratch.get(actor.device_, 1, Nc, False, "actor")
    ops.mlp_features(actor.flat, actor.desc, obs, None, B, s["featT"])
    h_next = torch.empty(Nc, actor._recurrent_N, H, dtype=torch.float32, device=actor.device_)
    if counter is None:
        counter = actor._sample_counter
        actor._sample_counter += 1
    ops.gru_forward(actor.flat, actor.desc, s["featT"], rnn_states.reshape(Nc, H), None, masks.reshape(B), None, 1, Nc,
                    h_last=h_next.view(Nc, H), head_mode=2, avail=avail, deterministic=deterministic, seed=actor._seed,
                    counter=counter, counter_dev=actor._counter_dev, actions=actions_f, logp=logp)
    return h_next


def can_step_dual(actor, critic):
    """Both networks recurrent with inputs of the same class (<= 64 wide, or 65..512) and the same trunk shape: their rollout
    step runs as two dual launches."""
    da, dc = actor.desc, critic.desc
    same_class = (da.in_dim <= 64) == (dc.in_dim <= 64) and max(da.in_dim, dc.in_dim) <= 512      # both narrow or both wide
    return (da.recurrent and dc.recurrent and same_class and da.layer_N == dc.layer_N
            and da.use_relu == dc.use_relu and actor._recurrent_N == 1 and critic._recurrent_N == 1)


def step_dual(actor, critic, obs, cent_obs, rnn_a, rnn_c, masks, avail, deterministic, actions_f, logp, values, counter):
    """R_Actor.forward + R_Critic.forward of one rollout step on the same rows, as TWO launches on one stream (features of
    both networks; GRU step + head of both networks) instead of two 2-kernel chains on two queues — the fork / join latency
    between the queues was as long as the kernels.  Returns the next (actor, critic) states [Nc, 1, H]."""
    Nc = obs.shape[0]
    dev = actor.device_
    ha = torch.empty(Nc, 1, H, dtype=torch.float32, device=dev)
    hc = torch.empty(Nc, 1, H, dtype=torch.float32, device=dev)
    wide = min(actor.desc.in_dim, critic.desc.in_dim) > 64
    if (((max(actor.desc.in_dim, critic.desc.in_dim) <= 64 and actor.desc.layer_N <= 1 and Nc <= 1024) or (wide and Nc <= 4096))
            and os.environ.get("MAPPO_FUSED_STEP", "1") != "0"):
        # wide inputs: split-K trunks + GRU step + heads of both networks in one launch, one 16-row tile per 4-wave workgroup
        # narrow inputs, at most 64 tiles per network: trunks, GRU steps and heads of both networks in ONE launch (every wave
        # of a tile's workgroup holds the trunk's weights in registers — with more tiles the separate launches, which stage the
        # weights once per workgroup of four tiles, are faster: 3 072 rows measured 2x slower fused)
        ops.recurrent_step_dual(actor.flat, actor.desc, obs, rnn_a.reshape(Nc, H), ha.view(Nc, H), critic.flat, critic.desc, cent_obs,
                                rnn_c.reshape(Nc, H), hc.view(Nc, H), masks.reshape(Nc), Nc, avail, deterministic, actor._seed, counter,
                                actor._counter_dev, actions_f, logp, values.view(Nc))
        return ha, hc
    sa, sc = _scratch.get(dev, 1, Nc, False, "actor"), _scratch.get(dev, 1, Nc, False, "critic")
    ops.mlp_features_dual(actor.flat, actor.desc, obs, sa["featT"], critic.flat, critic.desc, cent_obs, sc["featT"], Nc)
    ops.gru_step_dual(actor.flat, actor.desc, sa["featT"], rnn_a.reshape(Nc, H), ha.view(Nc, H), critic.flat, critic.desc, sc["featT"],
                      rnn_c.reshape(Nc, H), hc.view(Nc, H), masks.reshape(Nc), Nc, avail, deterministic, actor._seed, counter,
                      actor._counter_dev, actions_f, logp, values.view(Nc))
    return ha, hc


def actor_sequence_logits(actor, obs, rnn_states, masks):
    """Pre-mask logits for every row of a (L*Nc)-row time-major batch (evaluate_actions, r_actor_critic.py:72-107)."""
    B = obs.shape[0]
    L, Nc = _seq_shape(B, rnn_states)
    s = _scratch.get(actor.device_, L, Nc, False, "actor")
    ops.mlp_features(actor.flat, actor.desc, obs, None, B, s["featT"])
    logits = torch.empty(B, actor.n_actions, dtype=torch.float32, device=actor.device_)
    ops.gru_forward(actor.flat, actor.desc, s["featT"], rnn_states.reshape(Nc, H), None, masks.reshape(B), None, L, Nc,
                    head_mode=1, out=logits)
    return logits


def critic_forward(critic, cent_obs, rnn_states, masks, values):
    """R_Critic.forward (r_actor_critic.py:146-165): single step (rows == states) or L-step sequences."""
    B = cent_obs.shape[0]
    L, Nc = _seq_shape(B, rnn_states)
    s = _scratch.get(critic.device_, L, Nc, False, "critic")
    ops.mlp_features(critic.flat, critic.desc, cent_obs, None, B, s["featT"])
    h_next = torch.empty(Nc, critic._recurrent_N, H, dtype=torch.float32, device=critic.device_)
    ops.gru_forward(critic.flat, critic.desc, s["featT"], rnn_states.reshape(Nc, H), None, masks.reshape(B), None, L, Nc,
                    h_last=h_next.view(Nc, H), head_mode=1, out=values.view(B, 1))
    return h_next


# ---- training -------------------------------------------------------------------------------------------------------
def _update_recurrent(tr, src, rows, h0_rows, L, Nc, update_actor, epochs=None):
    """One PPO update on `Nc` sequences of `L` steps (r_mappo.py:91-164 with the recurrent evaluate_actions).
    `epochs` = (e, n, states[n, 3]): the minibatch is the same set of rows in every one of the n ppo epochs (num_mini_batch
    == 1: every permutation of the chunks covers the same rows), so epoch 0 takes the batch moments and performs all n
    ValueNorm updates in one launch (mappo_valuenorm_update_n) and update e normalises with states[e]."""
    pol = tr.policy
    B = L * Nc
    lib = ops._lib.load()
    dev = tr.device
    vn_state = tr.value_normalizer.state if tr._use_valuenorm else None
    if epochs is None or epochs[0] == 0:
        ops.minibatch_moments(src["returns"], src["active"], rows, B, tr._mb_moments,
                              tr._bytes("mom_ws", lib.mappo_moments_workspace_bytes(B)))
        if tr._dist is not None:
            tr._dist.all_reduce_sum_(tr._mb_moments)
        if tr._use_valuenorm:
            if epochs is None:
                ops.valuenorm_update(vn_state, tr._mb_moments, tr.value_normalizer.beta)
            else:
                ops.valuenorm_update_n(vn_state, tr._mb_moments, tr.value_normalizer.beta, epochs[1], epochs[2])
    if epochs is not None and tr._use_valuenorm:
        vn_state = epochs[2][epochs[0]]
    n_trunk = ops.mlp_backward_slabs(B)
    n_bwd = ops.gru16_slabs(L, Nc)                             # rows of loss partials / slab rows the 16-sequence-tile kernels may write
    n_slabs = max(n_trunk, n_bwd)
    P = pol.n_flat
    slabs = tr._buf("slabs_rec", (n_slabs, P), zero=True)      # rows a kernel never writes stay zero
    if not update_actor and not tr._actor_slabs_clean:
        slabs[:, :pol.seg_bounds[1]].zero_()
    tr._actor_slabs_clean = not update_actor
    pa = tr._buf("partials_a", (1024,), torch.float64, zero=True)
    pc = tr._buf("partials_c", (1024,), torch.float64, zero=True)
    nets = []
    if update_actor:
        nets.append((pol.actor, src["obs"], src["h0_a"], 1, pa, 0, "actor"))
    nets.append((pol.critic, src["share_obs"], src["h0_c"], 2, pc, pol.seg_bounds[1], "critic"))

    def one_net(net, x, h0, head, part, col0, tag):          # 16-sequence-tile kernels (gru_train16.hip)
        # blocked trunk features / d x (one contiguous KiB per wave access): narrow inputs always, wide ones when the sequence count
        # is a multiple of 16 (the flat 16-row tiles of the wide kernels then are the (t, 16 sequences) tiles)
        narrow = net.desc.layer_N <= 1 and (4 <= net.desc.in_dim <= 64 or (64 < net.desc.in_dim <= 512 and Nc % 16 == 0 and _WIDE_BLOCKED))
        s = _scratch.get16(dev, L, Nc, tag, narrow)
        if narrow:
            ops.mlp_features_seq(net.flat, net.desc, x, rows, L, Nc, s["feat"])
        else:
            ops.mlp_features(net.flat, net.desc, x, rows, B, s["feat"])
        ops.gru16_forward_loss(net.flat, net.desc, s["feat"], narrow, h0, h0_rows, src["masks"], rows, L, Nc, head,
                               src["avail"] if head == 1 else None, src["actions"] if head == 1 else None,
                               src["old_logp"] if head == 1 else None, src["adv"] if head == 1 else None, src["active"],
                               src["v_old"] if head == 2 else None, src["returns"] if head == 2 else None,
                               vn_state if head == 2 else None, tr._mb_moments, tr._cfg, s["scratch"], slabs, P, col0, part)
        ops.gru16_backward(net.flat, net.desc, src["masks"], rows, L, Nc, s["scratch"], None if narrow else s["dxT"])
        ops.gru16_wgrad(net.desc, s["feat"], narrow, s["scratch"], L, Nc, slabs, P, col0)
        if narrow:
            ops.trunk_backward_seq(net.flat, net.desc, x, rows, L, Nc, s["dx"], slabs, P, col0)
        else:
            ops.trunk_backward(net.flat, net.desc, x, rows, B, s["dxT"], slabs, P, col0)

    # The two networks' chains run side by side on two streams (they write disjoint slab columns and disjoint partials): at small
    # sizes (config-2 rmappo: 120 workgroups per sequence kernel) the second chain fills the idle CUs (train 6.96 -> 4.86 ms); at
    # config-3 size every kernel fills the chip by itself and the streams neither help nor hurt (23.0 ms either way).
    if len(nets) == 2 and _TWO_STREAMS:
        cur = torch.cuda.current_stream()
        if tr._side_stream is None:
            tr._side_stream = torch.cuda.Stream(device=dev)
        side = tr._side_stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            one_net(*nets[0])
        one_net(*nets[1])
        cur.wait_stream(side)
    else:
        for nt in nets:
            one_net(*nt)
    ops.update_stats(pa if update_actor else None, n_bwd, pc, n_bwd, tr._mb_moments, tr._cfg, tr._stats, tr._acc)
    if update_actor != tr._actor_enabled:
        pol.opt_hyper[0, 7] = 1.0 if update_actor else 0.0
        tr._actor_enabled = update_actor
    if tr._dist is None:                   # single process: reduction + clip + Adam in two launches
        ops.reduce_clip_adam(slabs, n_slabs, P, pol.flat_params, pol.flat_grad, pol.exp_avg, pol.exp_avg_sq, pol.seg_bounds,
                             pol.opt_hyper, pol.opt_step, pol.grad_norms, pol.opt_workspace, norm_acc=tr._acc[4:])
        return
    ops.slab_reduce(slabs, n_slabs, P, P, pol.flat_grad)
    tr._dist.all_reduce_sum_(pol.flat_grad)
    ops.clip_adam(pol.flat_params, pol.flat_grad, pol.exp_avg, pol.exp_avg_sq, pol.seg_bounds, pol.opt_hyper, pol.opt_step,
                  pol.grad_norms, pol.opt_workspace, norm_acc=tr._acc[4:])


def _buffer_sources(tr, buffer, adv):
    T = buffer.episode_length
    S = T * buffer.n_rollout_threads * buffer.num_agents
    flat = lambda a: a[:T].view(S, -1)
    return dict(obs=flat(buffer.obs), share_obs=flat(buffer.share_obs),
                avail=flat(buffer.available_actions) if buffer.available_actions is not None else None,
                actions=buffer.actions.view(S), old_logp=buffer.action_log_probs.view(S), adv=adv,
                active=buffer.active_masks[:T].view(S), v_old=buffer.value_preds[:T].view(S), returns=buffer.returns[:T].view(S),
                masks=buffer.masks[:T].view(S), h0_a=buffer.rnn_states[:T].view(S, H), h0_c=buffer.rnn_states_critic[:T].view(S, H))


def train_recurrent(tr, buffer, update_actor=True):
    """R_MAPPO.train for use_recurrent_policy (chunks of data_chunk_length) / use_naive_recurrent_policy (episodes): the launch
    sequence only, no host synchronisation (R_MAPPO.train captures it into a hipGraph and reads the statistics afterwards)."""
    if buffer.recurrent_N != 1:
        raise NotImplementedError("recurrent_N != 1")
    adv = tr.compute_advantages(buffer)
    src = _buffer_sources(tr, buffer, adv)
    T = buffer.episode_length
    tr._acc.zero_()
    all_batches = None
    if tr._use_recurrent_policy and buffer.perm_device != "cpu":
        all_batches = buffer.recurrent_rows_epochs(tr.ppo_epoch, tr.num_mini_batch, tr.data_chunk_length)   # off the epochs' critical path
    for epoch in range(tr.ppo_epoch):
        if all_batches is not None:
            L = tr.data_chunk_length
            batches = all_batches[epoch]
        elif tr._use_recurrent_policy:
            L = tr.data_chunk_length
            batches = buffer.recurrent_rows(tr.num_mini_batch, L)
        else:
            L = T
            batches = buffer.naive_recurrent_rows(tr.num_mini_batch)
        ep = (epoch, tr.ppo_epoch, tr._buf("vn_states", (tr.ppo_epoch, 3))) if tr.num_mini_batch == 1 else None
        for rows, h0_rows in batches:
            _update_recurrent(tr, src, rows, h0_rows, L, h0_rows.numel(), update_actor, ep)


def ppo_update_recurrent(tr, sample, update_actor=True):
    """R_MAPPO.ppo_update with an explicit (already gathered, time-major) recurrent sample tuple."""
    (share_obs, obs, rnn_a, rnn_c, actions, v_old, ret, masks, active, old_logp, adv, avail) = sample
    d = lambda x: to_device_f32(x, tr.device)
    rnn_a, rnn_c = d(rnn_a), d(rnn_c)
    Nc = rnn_a.shape[0]
    B = (obs.shape[0] if torch.is_tensor(obs) else np.shape(obs)[0])
    L = B // Nc
    src = dict(obs=d(obs), share_obs=d(share_obs), avail=d(avail) if avail is not None else None, actions=d(actions).view(B),
               old_logp=d(old_logp).view(B), adv=d(adv).view(B), active=d(active).view(B), v_old=d(v_old).view(B),
               returns=d(ret).view(B), masks=d(masks).view(B), h0_a=rnn_a.reshape(Nc, H), h0_c=rnn_c.reshape(Nc, H))
    tr._acc.zero_()
    _update_recurrent(tr, src, None, None, L, Nc, update_actor)
    a = tr._acc.cpu().numpy()
    return a[0], a[5], a[1], a[2], a[4], a[3]
