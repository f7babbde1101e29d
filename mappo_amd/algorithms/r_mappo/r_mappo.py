"""R_MAPPO trainer — API of `onpolicy/algorithms/r_mappo/r_mappo.py:8-227`.

`train(buffer)` keeps the reference's structure (advantage normalisation, ppo_epoch x num_mini_batch updates,
six averaged statistics) but every update is a fixed sequence of HIP launches on the HBM-resident buffer:

    minibatch moments -> ValueNorm update -> actor update kernel -> critic update kernel ->
    slab reduce -> [RCCL all-reduce] -> clip + Adam

where an "update kernel" (mappo_actor_update / mappo_critic_update) runs forward, PPO loss gradient and backward
of one network per 32-sample tile inside one wavefront, so logits / values / their gradients never reach HBM.
`--unfused_update` switches to the equivalent five-launch sequence (forward x2, mappo_ppo_loss_fwd_bwd,
backward x2) built from the standalone ops.

The minibatch is never materialised: kernels take int32 row indices into the buffer (or stream it in place
when num_mini_batch == 1, where the permutation only reorders the terms of sums).  No `.item()` inside the
loop (the reference syncs 3x per minibatch, r_mappo.py:207-209): statistics accumulate on the device and
are read once at the end of train()."""
import numpy as np
import os
import torch

from mappo_amd import ops
from mappo_amd.utils.util import to_device_f32
from mappo_amd.utils.valuenorm import ValueNorm


class R_MAPPO():
    def __init__(self, args, policy, device=torch.device("cuda"), dist_group=None):
        self.device = torch.device(device)
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.policy = policy
        self.args = args

        self.clip_param = args.clip_param
        self.ppo_epoch = args.ppo_epoch
        self.num_mini_batch = args.num_mini_batch
        self.data_chunk_length = args.data_chunk_length
        self.value_loss_coef = args.value_loss_coef
        self.entropy_coef = args.entropy_coef
        self.max_grad_norm = args.max_grad_norm
        self.huber_delta = args.huber_delta

        self._use_recurrent_policy = args.use_recurrent_policy
        self._use_naive_recurrent = args.use_naive_recurrent_policy
        self._use_max_grad_norm = args.use_max_grad_norm
        self._use_clipped_value_loss = args.use_clipped_value_loss
        self._use_huber_loss = args.use_huber_loss
        self._use_popart = args.use_popart
        self._use_valuenorm = args.use_valuenorm
        self._use_value_active_masks = args.use_value_active_masks
        self._use_policy_active_masks = args.use_policy_active_masks
        self._exact_order = bool(getattr(args, "exact_minibatch_order", False))
        self._fused = not bool(getattr(args, "unfused_update", False))
        self._use_graph = bool(getattr(args, "use_hip_graph", True))
        self._concurrent_update = bool(getattr(args, "concurrent_update", False))
        self._graphs = {}
        self._dp_graphs = {}                       # data-parallel runs: hipGraphs of the collective-free segments
        self._graph_buffers = {}                   # id(buffer) -> buffer: referenced for as long as its graphs are cached

        assert (self._use_popart and self._use_valuenorm) == False, \
            "self._use_popart and self._use_valuenorm can not be set True simultaneously"
        if self._use_popart:
            raise NotImplementedError("use_popart: PopArt.update raises in the reference itself (SURVEY.md §8c)")
        self.value_normalizer = ValueNorm(1, device=self.device) if self._use_valuenorm else None

        self._cfg = ops.ppo_cfg(args)
        self._cfg_acc = ops.ppo_cfg(args, accumulate_partials=True)
        self._dual_update = bool(getattr(args, "dual_update", True)) and os.environ.get("MAPPO_DUAL_UPDATE", "1") != "0"
        self._epochs = None                        # whole-buffer fused train(): per-epoch ValueNorm states + deferred statistics
        self._dist = dist_group                      # mappo_amd.distributed.DataParallel or None
        if dist_group is not None:
            # the actor keyed its sampling stream by the rank it saw at construction — rank 0 if the process group was created
            # AFTER the policy; re-key from the group this trainer reduces over, so shards never share exploration noise
            from mappo_amd.distributed import sampling_seed
            policy.actor._seed = sampling_seed(int(getattr(args, "seed", 1)), dist_group.rank)
        f64 = dict(dtype=torch.float64, device=self.device)
        self._mb_moments = torch.zeros(4, **f64)
        self._adv_moments = torch.zeros(3, **f64)
        self._stats = torch.zeros(6, **f64)
        # one allocation, one fill per train(): [statistics accumulators (6, padded to 8) | actor loss partials | critic loss partials]
        self._zbuf = torch.zeros(8 + 2 * 1024, **f64)
        self._acc = self._zbuf[:6]                   # value_loss, policy_loss, dist_entropy, ratio, actor_gn, critic_gn
        self._pa, self._pc = self._zbuf[8:8 + 1024], self._zbuf[8 + 1024:]
        self._ws = {}
        self._training = False

    # ---- workspaces (allocated once per shape; nothing is allocated inside the update loop) -----------------
    def _buf(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    def _bytes(self, name, nbytes):
        return self._buf(name, (int(nbytes),), torch.uint8)

    # ---- one PPO update on rows of flat source arrays (r_mappo.py:91-164) -------------------------------------
    def _update(self, src, rows, B, update_actor=True, moments_ready=False, part="all"):
        """part (data-parallel runs only): "kernels" = everything up to the local gradient (no collective inside: this is
        the segment the data-parallel trainer captures into a hipGraph), "optim" = gradient all-reduce + clip + Adam."""
        pol = self.policy
        A = pol.actor.n_actions
        lib = ops._lib.load()
        vn_state = self.value_normalizer.state if self._use_valuenorm else None
        n_slabs = ops.mlp_backward_slabs(B)
        P = pol.n_flat
        if part != "optim":
            self._slab_rows = n_slabs
            slabs = self._update_kernels(src, rows, B, update_actor, moments_ready, vn_state, n_slabs, P)
            n_slabs = self._slab_rows                      # the dual launch writes fewer slab rows per network
        else:
            slabs = self._buf("slabs", (n_slabs, P))
        if update_actor != self._actor_enabled and part != "kernels":      # torch >= 2: grad None => Adam skips the actor
            pol.opt_hyper[0, 7] = 1.0 if update_actor else 0.0
            self._actor_enabled = update_actor
        if self._dist is None:
            # single process: reduction + clip + Adam in two launches
            ops.reduce_clip_adam(slabs, n_slabs, P, pol.flat_params, pol.flat_grad, pol.exp_avg, pol.exp_avg_sq, pol.seg_bounds,
                                 pol.opt_hyper, pol.opt_step, pol.grad_norms, pol.opt_workspace, norm_acc=self._acc[4:])
            return
        if part != "optim":
            ops.slab_reduce(slabs, n_slabs, P, P, pol.flat_grad)
        if part != "kernels":
            self._dist.all_reduce_sum_(pol.flat_grad)          # C1: one flat fp32 all-reduce per minibatch
            ops.clip_adam(pol.flat_params, pol.flat_grad, pol.exp_avg, pol.exp_avg_sq, pol.seg_bounds, pol.opt_hyper,
                          pol.opt_step, pol.grad_norms, pol.opt_workspace, norm_acc=self._acc[4:])

    def _moments(self, src, rows, B, vn_state):
        """{sum ret, sum ret^2, sum active, B} of the minibatch (+ the data-parallel all-reduce) and, in the epoch-batched
        mode, all ppo_epoch ValueNorm updates in one launch (every epoch sees the same batch moments)."""
        lib = ops._lib.load()
        ops.minibatch_moments(src["returns"], src["active"], rows, B, self._mb_moments,
                              self._bytes("mom_ws", lib.mappo_moments_workspace_bytes(B)))
        if self._dist is not None:
            self._dist.all_reduce_sum_(self._mb_moments)
        ep = self._epochs
        if ep is not None and self._use_valuenorm:
            ops.valuenorm_update_n(vn_state, self._mb_moments, self.value_normalizer.beta, ep["n"], ep["states"])

    def _update_kernels(self, src, rows, B, update_actor, moments_ready, vn_state, n_slabs, P):
        pol = self.policy
        A = pol.actor.n_actions
        lib = ops._lib.load()
        # denominators of the masked means + the moments ValueNorm.update needs (cal_value_loss, r_mappo.py:65);
        # `moments_ready`: the minibatch is the whole buffer again, its sums were taken by the first epoch
        ep = self._epochs
        if not moments_ready:
            self._moments(src, rows, B, vn_state)
        # (ValueNorm.update and the loss statistics are tiny kernels that nothing waits for immediately; forking them to a
        # side stream next to the update kernels measured SLOWER inside the captured hipGraph — train 1.96 ms vs 1.69 ms
        # at config 2 — so the chain stays on one stream.)
        cfg = self._cfg
        if ep is not None:
            cfg = self._cfg_acc                               # loss sums accumulate over the epochs, one statistics launch at the end
            if self._use_valuenorm:
                vn_state = ep["states"][ep["e"]]              # state after this epoch's ValueNorm.update
        elif self._use_valuenorm:
            ops.valuenorm_update(vn_state, self._mb_moments, self.value_normalizer.beta)
        slabs = self._buf("slabs", (n_slabs, P), zero=True)
        if not update_actor and not self._actor_slabs_clean:
            slabs[:, :pol.seg_bounds[1]].zero_()
        self._actor_slabs_clean = not update_actor
        if self._fused:
            pa, pc = self._pa, self._pc
            # Each update kernel wants one workgroup per CU (its LDS footprint), so two full-size launches run one after
            # the other.  --concurrent_update instead splits the 256 CUs between the two networks and launches them on two
            # streams; on MI355X this measured SLOWER (train 3.08 ms vs 2.52 ms at config 2), so it is off by default.
            na, nc = (0, 0)
            if update_actor and self._concurrent_update:      # measured slower on MI355X (the kernels interfere): off by default
                na, nc = self._split_grid(n_slabs, pol.actor.desc.in_dim, pol.critic.desc.in_dim, B)
            if na:
                cur = torch.cuda.current_stream()
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=self.device)
                side = self._side_stream
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.actor_update(pol.actor.flat, pol.actor.desc, src["obs"], rows, B, src["avail"], src["actions"],
                                     src["old_logp"], src["adv"], src["active"], self._mb_moments, cfg, slabs, P, 0, pa, na)
                ops.critic_update(pol.critic.flat, pol.critic.desc, src["share_obs"], rows, B, src["v_old"], src["returns"],
                                  src["active"], vn_state, self._mb_moments, cfg, slabs, P, pol.seg_bounds[1], pc, nc)
                cur.wait_stream(side)
                n_pa, n_pc = na, nc
            elif update_actor and self._dual_update and pol.can_dual_update():
                # both networks in ONE launch, half the CUs each (mappo_actor_critic_update): one ragged tail instead of two
                ops.actor_critic_update(pol.actor.flat, pol.actor.desc, src["obs"], pol.critic.flat, pol.critic.desc, src["share_obs"],
                                        rows, B, src["avail"], src["actions"], src["old_logp"], src["adv"], src["active"], src["v_old"],
                                        src["returns"], vn_state, self._mb_moments, cfg, slabs, P, 0, pol.seg_bounds[1], pa, pc)
                n_pa = n_pc = ops.dual_update_slabs(pol.actor.desc, pol.critic.desc, B)
                self._slab_rows = n_pa
            else:
                if update_actor:
                    ops.actor_update(pol.actor.flat, pol.actor.desc, src["obs"], rows, B, src["avail"], src["actions"],
                                     src["old_logp"], src["adv"], src["active"], self._mb_moments, cfg, slabs, P, 0, pa)
                ops.critic_update(pol.critic.flat, pol.critic.desc, src["share_obs"], rows, B, src["v_old"], src["returns"],
                                  src["active"], vn_state, self._mb_moments, cfg, slabs, P, pol.seg_bounds[1], pc)
                n_pa = n_pc = n_slabs
            if ep is None:
                ops.update_stats(pa if update_actor else None, n_pa, pc, n_pc, self._mb_moments, self._cfg, self._stats, self._acc)

        else:
            # evaluate_actions: logits and values (rMAPPOPolicy.py:88-114)
            logits = self._buf("logits", (B, A))
            values = self._buf("values", (B,))
            ops.mlp_forward(pol.actor.flat, pol.actor.desc, src["obs"], rows, B, logits)
            ops.mlp_forward(pol.critic.flat, pol.critic.desc, src["share_obs"], rows, B, values)
            dlogits = self._buf("dlogits", (B, A))
            dvalues = self._buf("dvalues", (B,))
            ops.ppo_loss_fwd_bwd(logits, values, rows, src["avail"], src["actions"], src["old_logp"], src["adv"],
                                 src["active"], src["v_old"], src["returns"], vn_state, self._mb_moments, dlogits, dvalues,
                                 self._stats, self._cfg, self._bytes("loss_ws", lib.mappo_ppo_loss_workspace_bytes(B)))
            self._acc[:4].add_(self._stats[:4])
            if update_actor:
                ops.mlp_backward(pol.actor.flat, pol.actor.desc, src["obs"], rows, B, dlogits, slabs, P, 0)
            ops.mlp_backward(pol.critic.flat, pol.critic.desc, src["share_obs"], rows, B, dvalues, slabs, P, pol.seg_bounds[1])
        return slabs

    _actor_slabs_clean = True
    _actor_enabled = True
    _side_stream = None

    @staticmethod
    def _split_grid(n_slabs, d_actor, d_critic, B):
        """(actor blocks, critic blocks) for the concurrent launch, or (0, 0) to launch one after the other.  Only worth
        it when every CU is busy anyway (n_slabs == 256) and both networks take the narrow-input kernels."""
        if n_slabs < 256 or d_actor > 64 or d_critic > 64:
            return 0, 0
        tiles = (B + 31) // 32
        cost = lambda d: 1.0 + 0.25 * (d > 32)            # measured per-tile cost ratio (critic D=54 vs actor D=18)
        best = None
        for na in range(96, 161, 4):
            nc = 256 - na
            t = max(-(-tiles // (4 * na)) * cost(d_actor), -(-tiles // (4 * nc)) * cost(d_critic))
            if best is None or t < best[0]:
                best = (t, na, nc)
        return best[1], best[2]

    def _buffer_sources(self, buffer, adv):
        T = buffer.episode_length
        R = buffer.n_rollout_threads * buffer.num_agents
        S = T * R
        flat = lambda a: a[:T].view(S, -1)
        return dict(obs=flat(buffer.obs), share_obs=flat(buffer.share_obs),
                    avail=flat(buffer.available_actions) if buffer.available_actions is not None else None,
                    actions=buffer.actions.view(S), old_logp=buffer.action_log_probs.view(S), adv=adv,
                    active=buffer.active_masks[:T].view(S), v_old=buffer.value_preds[:T].view(S),
                    returns=buffer.returns[:T].view(S)), S

    # ---- r_mappo.py:166-219 ---------------------------------------------------------------------------------
    @torch.no_grad()
    def train(self, buffer, update_actor=True, after_update=False):
        """`after_update=True` also performs buffer.after_update() (base_runner.py:124) inside the same launch
        sequence.  With num_mini_batch == 1 the whole sequence (advantages, ppo_epoch updates, after_update) is
        captured into ONE hipGraph after a first eager run and replayed afterwards: the inner loop is a fixed
        chain of ~15 short kernels per update, launch-bound when issued from Python."""
        self._sync_actor_mode(update_actor)
        if self._use_recurrent_policy or self._use_naive_recurrent:
            from mappo_amd.recurrent import train_recurrent
            # the chunk permutation is drawn on the device (graph-safe philox stream), so the ppo_epoch x ~25 launches of
            # the two networks' chains replay as one hipGraph; the reference's CPU permutation stream stays eager
            static = buffer.perm_device != "cpu"
            body = lambda: (train_recurrent(self, buffer, update_actor), buffer.after_update() if after_update else None)
        else:
            static = self.num_mini_batch == 1 and not self._exact_order   # no randperm inside => capturable
            body = lambda: self._train_body(buffer, update_actor, after_update)
        if not (self._use_graph and static and self._dist is None):
            body()
            return self._finish_train_info()
        key = (self._buffer_key(buffer), bool(update_actor), bool(after_update))
        state = self._graphs.get(key)
        if state is None:
            body()                                                         # eager: allocates workspaces, sets attributes
            self._graphs[key] = "warm"
        else:
            if state == "warm":
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
                self._graphs[key] = state = g
            state.replay()
        return self._finish_train_info()

    def _buffer_key(self, buffer):
        """Cache key of the captured graphs of a buffer.  The trainer keeps the buffer alive while it holds graphs recorded
        against its addresses, so the key (its id) cannot be reused by another object."""
        k = id(buffer)
        held = self._graph_buffers.setdefault(k, buffer)
        assert held is buffer
        return k

    def _sync_actor_mode(self, update_actor):
        """Host-side switches of update_actor (torch >= 2: grad None => Adam skips the actor), applied eagerly before the
        launch sequence so that a replayed hipGraph never depends on what the previous train() call was."""
        if update_actor != self._actor_enabled:
            self.policy.opt_hyper[0, 7] = 1.0 if update_actor else 0.0
            self._actor_enabled = update_actor
        if not update_actor and not self._actor_slabs_clean:
            for (name, _, _), t in self._ws.items():
                if name in ("slabs", "slabs_rec"):
                    t[:, :self.policy.seg_bounds[1]].zero_()
        self._actor_slabs_clean = not update_actor

    def _train_body(self, buffer, update_actor, after_update):
        T = buffer.episode_length
        S = T * buffer.n_rollout_threads * buffer.num_agents
        adv = self.compute_advantages(buffer)
        src, _ = self._buffer_sources(buffer, adv)
        self._zbuf.zero_()
        whole = self.num_mini_batch == 1 and not self._exact_order
        dp_graph = self._dist is not None and self._use_graph and whole and self._fused and self._dist.world_is_gpu
        key = (self._buffer_key(buffer), bool(update_actor))
        self._epochs = None
        if whole and self._fused and (self._dist is None or dp_graph) and not self._concurrent_update \
                and os.environ.get("MAPPO_EPOCH_BATCH", "1") != "0":
            # every update sees the same minibatch (the whole buffer): ValueNorm's ppo_epoch updates are one launch, the
            # loss sums accumulate in the kernels' partials and the statistics kernel runs once after the last epoch
            self._epochs = dict(n=self.ppo_epoch, e=0, states=self._buf("vn_states", (self.ppo_epoch, 3)))
        if dp_graph:
            # data parallel: the moments (one 4-double all-reduce) come first, so that EVERY epoch is the same collective-free
            # kernel segment (captured per epoch: the ValueNorm state an epoch reads sits at its own address)
            vn_state = self.value_normalizer.state if self._use_valuenorm else None
            self._moments(src, None, S, vn_state)
        for epoch in range(self.ppo_epoch):
            if self._epochs is not None:
                self._epochs["e"] = epoch
            if whole:
                batches = [(None, S)]                      # whole buffer in place (see module docstring)
            else:
                batches = [(rows, rows.numel()) for rows in buffer.feed_forward_rows(self.num_mini_batch)]
            for rows, B in batches:
                if dp_graph:
                    # data parallel: the collective stays outside; the kernels between two collectives are one hipGraph
                    gkey = key + (epoch,)
                    st = self._dp_graphs.get(gkey)
                    if st is None and self._dp_graphs.get(key) == "warm":
                        st = "warm"
                    if st == "warm":
                        torch.cuda.synchronize()
                        try:
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, capture_error_mode="thread_local"):     # RCCL's watchdog thread stays legal
                                self._update(src, rows, B, update_actor, moments_ready=True, part="kernels")
                            self._dp_graphs[gkey] = st = g
                        except Exception as e:                      # never let a failed capture take a multi-GPU run down
                            import warnings
                            warnings.warn(f"hipGraph capture of the data-parallel update segment failed ({e}); launching eagerly")
                            torch.cuda.synchronize()
                            self._dp_graphs[gkey] = st = "off"
                    if st == "off":
                        st = None
                    if st is None:
                        self._update(src, rows, B, update_actor, moments_ready=True, part="kernels")
                    else:
                        st.replay()
                    self._update(src, rows, B, update_actor, moments_ready=True, part="optim")
                else:
                    self._update(src, rows, B, update_actor, moments_ready=whole and epoch > 0)
        if self._epochs is not None:
            # (rows of loss partials the update kernels wrote; not taken from a side effect of _update_kernels: under data
            # parallelism the epochs may have been graph replays, during which no Python runs)
            n_rows = ops.dual_update_slabs(self.policy.actor.desc, self.policy.critic.desc, S) if (update_actor and self._dual_update and self.policy.can_dual_update()) \
                else ops.mlp_backward_slabs(S)
            ops.update_stats(self._pa if update_actor else None, n_rows, self._pc, n_rows, self._mb_moments, self._cfg, self._stats,
                             self._acc)
            self._epochs = None
        if dp_graph and self._dp_graphs.get(key) is None:
            self._dp_graphs[key] = "warm"                      # first train() ran eagerly: workspaces exist now
        if after_update:
            buffer.after_update()

    def compute_advantages(self, buffer):
        """r_mappo.py:174-182 as two kernels around an (optional) 3-double all-reduce."""
        T = buffer.episode_length
        S = T * buffer.n_rollout_threads * buffer.num_agents
        adv = self._buf("adv", (S,))
        lib = ops._lib.load()
        vn_state = self.value_normalizer.state if self._use_valuenorm else None
        ops.adv_moments(buffer.returns[:T].view(S), buffer.value_preds[:T].view(S), buffer.active_masks[:T].view(S), vn_state,
                        adv, self._adv_moments, self._bytes("adv_ws", lib.mappo_adv_workspace_bytes(S)))
        if self._dist is not None:
            self._dist.all_reduce_sum_(self._adv_moments)
        ops.adv_normalize(adv, self._adv_moments)
        return adv

    def _finish_train_info(self):
        num_updates = self.ppo_epoch * self.num_mini_batch
        if self._dist is not None:
            self._dist.all_reduce_sum_(self._acc[:4])          # local numerators / global denominators -> global stats
        acc = (self._acc / max(num_updates, 1)).cpu().numpy()    # the only host sync of train()
        return dict(value_loss=float(acc[0]), policy_loss=float(acc[1]), dist_entropy=float(acc[2]),
                    actor_grad_norm=float(acc[4]), critic_grad_norm=float(acc[5]), ratio=float(acc[3]))

    # ---- r_mappo.py:91-164 with an explicit (already gathered) sample tuple ---------------------------------
    @torch.no_grad()
    def ppo_update(self, sample, update_actor=True):
        if self._use_recurrent_policy or self._use_naive_recurrent:
            from mappo_amd.recurrent import ppo_update_recurrent
            return ppo_update_recurrent(self, sample, update_actor)
        (share_obs, obs, rnn_a, rnn_c, actions, v_old, ret, masks, active, old_logp, adv, avail) = sample
        d = lambda x: to_device_f32(x, self.device)
        B = np.shape(obs)[0] if not torch.is_tensor(obs) else obs.shape[0]
        src = dict(obs=d(obs), share_obs=d(share_obs), avail=d(avail) if avail is not None else None,
                   actions=d(actions).view(B), old_logp=d(old_logp).view(B), adv=d(adv).view(B), active=d(active).view(B),
                   v_old=d(v_old).view(B), returns=d(ret).view(B))
        self._acc.zero_()
        self._update(src, None, B, update_actor)
        a = self._acc.cpu().numpy()
        # value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights (mean here)
        return a[0], a[5], a[1], a[2], a[4], a[3]

    # ---- r_mappo.py:52-89 (API completeness; train() uses the fused kernel instead) --------------------------
    @torch.no_grad()
    def cal_value_loss(self, values, value_preds_batch, return_batch, active_masks_batch):
        d = lambda x: to_device_f32(x, self.device)
        values, value_preds_batch, return_batch, active_masks_batch = d(values), d(value_preds_batch), d(return_batch), d(active_masks_batch)
        clipped = value_preds_batch + (values - value_preds_batch).clamp(-self.clip_param, self.clip_param)
        if self._use_valuenorm:
            self.value_normalizer.update(return_batch)
            tgt = self.value_normalizer.normalize(return_batch)
        else:
            tgt = return_batch
        e_c, e_o = tgt - clipped, tgt - values

        def loss(e):
            if self._use_huber_loss:
                dl = self.huber_delta
                return torch.where(e.abs() <= dl, e * e / 2, dl * (e.abs() - dl / 2))
            return e * e / 2
        l = torch.max(loss(e_o), loss(e_c)) if self._use_clipped_value_loss else loss(e_o)
        if self._use_value_active_masks:
            return (l * active_masks_batch).sum() / active_masks_batch.sum()
        return l.mean()

    def prep_training(self):
        self._training = True

    def prep_rollout(self):
        self._training = False
