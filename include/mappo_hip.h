/*
 * mappo_hip.h — C ABI of libmappo_hip.so: the MI355X (gfx950) hot path of MAPPO training
 * (rollout forward -> GAE -> PPO minibatch update) as hand-written HIP kernels.
 *
 * The reference (Chen001117/mappo) has no FFI on this path: everything is Python/NumPy/torch
 * (SURVEY.md §8b).  Each entry point therefore cites the reference *Python* site it replaces; the
 * binding a maintainer would add on the reference side is the ctypes stub in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MAPPO_E* code; mappo_last_error() gives text.
 *   - all pointers are DEVICE pointers owned by the caller unless marked "host"; no hidden allocation,
 *     no host synchronisation; work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - all arrays are float32, row-major.  Buffer arrays use the reference's memory order
 *     [T(+1)][R = n_rollout_threads * num_agents][D] (shared_buffer.py:45-75), so a "flat row" is t*R + r.
 *   - `rows` arguments are int32 flat rows selecting the minibatch (the generators' fancy indexing,
 *     shared_buffer.py:246-286, 397-494); NULL means the identity 0..B-1.
 *   - ValueNorm state is 3 device floats {running_mean, running_mean_sq, debiasing_term} (valuenorm.py:21-23).
 */
#ifndef MAPPO_HIP_H
#define MAPPO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mappo_stream_t; /* hipStream_t */

#define MAPPO_OK 0
#define MAPPO_EINVAL (-1)    /* bad argument / unsupported shape */
#define MAPPO_ELAUNCH (-2)   /* HIP launch error */
#define MAPPO_ENOTIMPL (-3)  /* valid in the reference but not built here (e.g. popart) */

#define MAPPO_HIDDEN 64      /* hidden_size the MFMA kernels are tiled for (config.py:199 default) */
#define MAPPO_MAX_IN_DIM 512 /* obs / share_obs width */
#define MAPPO_MAX_ACTIONS 32 /* Discrete(n) with n <= 32 */
#define MAPPO_MAX_LAYER_N 2

const char *mappo_last_error(void);
int mappo_abi_version(void);

/* ---- network description -------------------------------------------------------------------------
 * One actor or critic: MLPBase (mlp.py:31-55) -> [GRU + LayerNorm (rnn.py:7-80)] -> Linear head
 * (Categorical logits, distributions.py:55-68, or v_out, r_actor_critic.py:136-142).
 * Parameters live in ONE flat float array in this order (state_dict order minus the unused fc_h):
 *   feature_norm.{w,b}[in_dim] (if use_feature_norm) | fc1.0.{W[H][in_dim], b[H]} | fc1.2.{w,b}[H] |
 *   layer_N x { fc2.i.0.{W[H][H], b[H]} | fc2.i.2.{w,b}[H] } |
 *   (if recurrent) rnn.{weight_ih[3H][H], weight_hh[3H][H], bias_ih[3H], bias_hh[3H]} | rnn.norm.{w,b}[H] |
 *   head.{W[out_dim][H], b[out_dim]}
 * mappo_net_param_count() returns the float count; offsets follow from the order above. */
typedef struct {
  int32_t in_dim;            /* obs_dim (actor) or share_obs_dim (critic) */
  int32_t hidden;            /* must be MAPPO_HIDDEN */
  int32_t out_dim;           /* n actions (actor) or 1 (critic) */
  int32_t layer_N;           /* config.py:201, 0..MAPPO_MAX_LAYER_N */
  int32_t use_relu;          /* config.py:203 (0 = tanh) */
  int32_t use_feature_norm;  /* config.py:208 */
  int32_t recurrent;         /* use_recurrent_policy || use_naive_recurrent_policy */
} mappo_net_desc;

int64_t mappo_net_param_count(const mappo_net_desc *desc /*host*/);

/* ---- K1: MPE rollout insert (mpe_runner.py:125-139, shared_buffer.py:96-112) in one launch -----------------------
 * Sources may be strided / broadcast views (strides in elements); destinations are the contiguous buffer slots
 * obs[step+1], share_obs[step+1], rewards[step], masks[step+1].  centralized != 0 builds share_obs as the thread's
 * concatenated agent observations repeated per agent (use_centralized_V). */
int mappo_insert_mpe(const float *obs, int64_t obs_stride_n, int64_t obs_stride_m, const float *rewards,
                     int64_t rew_stride_n, int64_t rew_stride_m, const uint8_t *dones /*bool bytes*/,
                     int64_t done_stride_n, int64_t done_stride_m, float *obs_dst, float *share_dst, float *rew_dst,
                     float *mask_dst, int32_t N, int32_t M, int32_t D, int32_t centralized, mappo_stream_t stream);
/* The same for recurrent policies (mpe_runner.py:126-128 + shared_buffer.py:96-97): additionally
 * rnn_dst / rnn_critic_dst [N*M][H] (slot step+1) = rnn_states / rnn_states_critic * (1 - done); H = recurrent_N * hidden. */
int mappo_insert_mpe_rnn(const float *obs, int64_t obs_stride_n, int64_t obs_stride_m, const float *rewards,
                         int64_t rew_stride_n, int64_t rew_stride_m, const uint8_t *dones, int64_t done_stride_n,
                         int64_t done_stride_m, float *obs_dst, float *share_dst, float *rew_dst, float *mask_dst, int32_t N,
                         int32_t M, int32_t D, int32_t centralized, const float *rnn_states, const float *rnn_states_critic,
                         float *rnn_dst, float *rnn_critic_dst, int32_t H, mappo_stream_t stream);

/* K1 for the SMAC runner (smac_runner.py:129-151 + shared_buffer.py:74-112) in one launch.  dones_env[n] = all_m dones[n][m];
 * masks = 1 - dones_env, active_masks = dones_env ? 1 : 1 - dones, bad_masks = 1 - bad_transition (NULL: all 1),
 * rnn slots = states * (1 - dones_env) (rnn_states NULL: skipped); obs [N*M][D], share_obs [N*M][S], avail [N*M][A] (or NULL)
 * and rewards are copied to their slots.  dones / bad_transition are bool bytes. */
int mappo_insert_smac(const float *obs, const float *share_obs, const float *avail, const float *rewards, int64_t rew_stride_n,
                      int64_t rew_stride_m, const uint8_t *dones, int64_t done_stride_n, int64_t done_stride_m,
                      const uint8_t *bad_transition, const float *rnn_states, const float *rnn_states_critic, float *obs_dst,
                      float *share_dst, float *avail_dst, float *rew_dst, float *mask_dst, float *bad_mask_dst,
                      float *active_mask_dst, float *rnn_dst, float *rnn_critic_dst, int32_t N, int32_t M, int32_t D, int32_t S,
                      int32_t A, int32_t H, mappo_stream_t stream);

/* recurrent_generator's row indices (shared_buffer.py:385-494) for all ppo epochs of one train() in one launch: perm
 * [n_epochs][data_chunks] (int64 permutations of the chunks, device), mbs = data_chunks / num_mini_batch;
 * rows [n_epochs][num_mini_batch][L*mbs] (time-major: l*mbs + j) and h0_rows [n_epochs][num_mini_batch][mbs] are buffer rows
 * (q % T)*R + q / T of flat position q = chunk*L + l in the reference's (n, m, t) order (T = episode_length, R = N*M). */
int mappo_recurrent_rows(const int64_t *perm, int32_t n_epochs, int64_t data_chunks, int32_t L, int32_t T, int32_t R,
                         int32_t num_mini_batch, int32_t *rows, int32_t *h0_rows, mappo_stream_t stream);

/* K1, after_update (shared_buffer.py:114-131): `count` (<= 16) independent fp32 device copies in one launch.
 * dst / src / n_floats are HOST arrays of device pointers / lengths. */
int mappo_copy_batch(int32_t count, float *const *dst, const float *const *src, const int64_t *n_floats, mappo_stream_t stream);

/* ---- K2: compute_returns (shared_buffer.py:168-224), all four flag branches ------------------------
 * Segmented affine-map scan over T: one wavefront per (64 series x time segment), segment composites
 * combined through LDS.  Writes value_preds[T] <- next_value (GAE branches) or returns[T] <- next_value
 * (discounted branches) exactly like the reference.  vn_state NULL => no value normaliser. */
int mappo_gae_scan(const float *rewards /*[T][R]*/, float *value_preds /*[T+1][R]*/,
                   const float *next_value /*[R]*/, const float *masks /*[T+1][R]*/,
                   const float *bad_masks /*[T+1][R]*/, float *returns /*[T+1][R]*/,
                   const float *vn_state /*[3] or NULL*/, int32_t T, int32_t R, float gamma,
                   float gae_lambda, int32_t use_gae, int32_t use_proper_time_limits,
                   mappo_stream_t stream);

/* ---- K3: advantage build + normalisation (r_mappo.py:174-182) ---------------------------------------
 * adv = returns - denorm(value_preds); moments {sum, sum_sq, count} over entries with active != 0
 * (double accumulation, deterministic two-stage reduction); adv <- (adv - mean) / (std + 1e-5).
 * The two calls are split so that a multi-GPU caller can all-reduce `moments` in between (§8e C2). */
int64_t mappo_adv_workspace_bytes(int64_t n);
int mappo_adv_moments(const float *returns /*[n]*/, const float *value_preds /*[n]*/,
                      const float *active_masks /*[n]*/, const float *vn_state /*[3] or NULL*/,
                      float *adv /*[n] out (raw)*/, double *moments /*[3] out*/, void *workspace,
                      int64_t n, mappo_stream_t stream);
int mappo_adv_normalize(float *adv /*[n] in/out*/, const double *moments /*[3]*/, int64_t n,
                        mappo_stream_t stream);

/* ---- K12: ValueNorm (valuenorm.py:37-54) + minibatch denominators ---------------------------------
 * mappo_minibatch_moments: {sum ret, sum ret^2, sum active, B} over the minibatch rows (double).
 * mappo_valuenorm_update: EMA update of vn_state from those moments (batch mean = sum/B). */
int64_t mappo_moments_workspace_bytes(int64_t B);
int mappo_minibatch_moments(const float *returns, const float *active_masks, const int32_t *rows,
                            int64_t B, double *mb_moments /*[4] out*/, void *workspace,
                            mappo_stream_t stream);
int mappo_valuenorm_update(float *vn_state /*[3] in/out*/, const double *mb_moments /*[4]*/,
                           double beta, mappo_stream_t stream);
/* n successive ValueNorm.update calls with the SAME batch moments (ppo_epoch updates on whole-buffer minibatches) in
 * one launch: states_out[e] (e = 0..n-1, 3 floats each) is the state after e+1 updates — what update e's value loss
 * normalises with — and vn_state ends as states_out[n-1].  Same fp32 rounding sequence as n single calls. */
int mappo_valuenorm_update_n(float *vn_state /*[3] in/out*/, const double *mb_moments /*[4]*/, double beta, int32_t n,
                             float *states_out /*[n][3]*/, mappo_stream_t stream);

/* ---- K5 (+K6): fused PPO loss forward+backward (r_mappo.py:52-89,124-141; act.py:154-160) ----------
 * One pass over the minibatch: availability masking, log-softmax, log-prob gather, entropy,
 * ratio / clipped surrogate, clipped Huber|MSE value error; emits d(actor objective)/d logits,
 * d(value_loss_coef * value loss)/d values and per-block double partial sums of the 4 statistics.
 * [B][A] tiles are staged through LDS so that HBM sees only full-line traffic.
 * Algorithmic bytes per sample: 4*(3A+8) with avail, 4*(2A+8) without. */
typedef struct {
  float clip_param, entropy_coef, value_loss_coef, huber_delta;
  int32_t use_huber_loss, use_clipped_value_loss, use_policy_active_masks, use_value_active_masks,
      use_valuenorm;
  /* mappo_actor_update / mappo_critic_update: != 0 ADDS this launch's per-workgroup loss sums to `partials` instead of
   * overwriting them.  With identical denominators in every update (whole-buffer minibatches) one mappo_update_stats
   * call after the last epoch then yields the sum of the per-update statistics the trainer logs. */
  int32_t accumulate_partials;
} mappo_ppo_cfg;

int64_t mappo_ppo_loss_workspace_bytes(int64_t B);
int mappo_ppo_loss_fwd_bwd(const float *logits /*[B][A] minibatch order, pre-mask*/,
                           const float *values /*[B] minibatch order*/, const int32_t *rows /*[B]|NULL*/,
                           const float *avail /*[.][A] buffer order or NULL*/, const float *actions,
                           const float *old_logp, const float *adv, const float *active,
                           const float *v_old, const float *returns /*buffer order, indexed by rows*/,
                           const float *vn_state /*[3] AFTER update, or NULL*/,
                           const double *mb_moments /*[4] from mappo_minibatch_moments*/,
                           float *dlogits /*[B][A]*/, float *dvalues /*[B]*/, double *stats /*[6] out:
                           value_loss, policy_loss, dist_entropy, ratio_mean, sum_active, B*/,
                           void *workspace, const mappo_ppo_cfg *cfg /*host*/, int64_t B, int32_t A,
                           mappo_stream_t stream);

/* ---- K7/K6/K8: fused MLP trunk + head on fp32 MFMA (mlp.py:18-28,50-55; act.py:78-81) --------------
 * Weights resident in LDS, activations of a 32-sample wave tile live in registers/LDS; LayerNorm and
 * the activation are fused between the v_mfma_f32_32x32x2_f32 chains.
 *   mappo_mlp_forward : out[B][out_dim] = head(trunk(x[rows]))                       (evaluate / get_values)
 *   mappo_actor_act   : same trunk + masking + sample|argmax + log-prob               (get_actions / act)
 *   mappo_mlp_backward: recomputes the forward per tile (no activation round trip through HBM) and
 *                       accumulates the parameter gradient of sum_b <dout[b], out[b]> into per-block
 *                       slabs [n_slabs][slab_stride] at column `slab_col0` (flat parameter order). */
int mappo_mlp_forward(const float *params, const mappo_net_desc *desc /*host*/, const float *x /*[.][in_dim]*/,
                      const int32_t *rows /*[B] or NULL*/, int64_t B, float *out /*[B][out_dim]*/,
                      mappo_stream_t stream);
int mappo_actor_act(const float *params, const mappo_net_desc *desc /*host*/, const float *obs,
                    const float *avail /*[B][A] or NULL*/, int64_t B, int32_t deterministic,
                    uint64_t seed, uint64_t counter, const uint64_t *counter_dev /*device word added to counter, or NULL*/,
                    float *actions /*[B] fp32 (buffer dtype)*/, float *logp /*[B]*/, mappo_stream_t stream);
/* One launch per rollout step (mpe_runner.py:95-139): actor get_actions + critic get_values (+ optionally the insert
 * of the env output the rows are read from).  Rows may be strided views of the env's output: with M > 0 sample i is
 * (thread n, agent m) = (i / M, i % M) and starts at base[n*stride_n + m*stride_m]; M == 0: contiguous [B][in_dim].
 * obs_dst != NULL additionally performs mappo_insert_mpe(obs, rewards, dones -> obs_dst, share_dst, rew_dst, mask_dst)
 * (N = B / M threads) inside the same launch.  actions == logp == NULL skips the actor (bootstrap value of the last
 * step: critic + insert).  Networks with in_dim <= 64 that share layer_N and the activation. */
int mappo_rollout_step(const float *actor_params, const mappo_net_desc *actor_desc /*host*/, const float *critic_params,
                       const mappo_net_desc *critic_desc /*host*/, const float *obs, int64_t obs_stride_n,
                       int64_t obs_stride_m, const float *share_obs, int64_t share_stride_n, int64_t share_stride_m,
                       int32_t M, int64_t B, const float *avail /*[B][A] or NULL*/, int32_t deterministic, uint64_t seed,
                       uint64_t counter, const uint64_t *counter_dev, float *actions /*[B]*/, float *logp /*[B]*/,
                       float *values /*[B]*/, float *obs_dst /*or NULL: no insert*/, float *share_dst, const float *rewards,
                       int64_t rew_stride_n, int64_t rew_stride_m, const uint8_t *dones, int64_t done_stride_n,
                       int64_t done_stride_m, float *rew_dst, float *mask_dst, int32_t centralized, mappo_stream_t stream);
int32_t mappo_mlp_backward_slabs(int64_t B); /* number of slabs the launch below will write */
int mappo_mlp_backward(const float *params, const mappo_net_desc *desc /*host*/, const float *x,
                       const int32_t *rows, int64_t B, const float *dout /*[B][out_dim]*/,
                       float *slabs, int64_t slab_stride, int64_t slab_col0,
                       float *wide_ws /*in_dim > 64: mappo_wide_workspace_floats(B) floats, else NULL*/,
                       mappo_stream_t stream);
/* Wide observations (64 < in_dim <= 512; reference: mlp.py:18-55 with a wide input layer): the update / backward launches
 * run layer 1 as their own MFMA kernel and leave d z1 plus the per-row LayerNorm statistics in `wide_ws`;
 * mappo_wide_l1_backward then produces the W1 and feature-norm gradient columns (one slab row per workgroup of its grid.x,
 * mappo_wide_l1_slabs(B) rows).  The workspace has one of two layouts, a PURE function of the descriptor and of which
 * entry point filled it — mappo_wide_layout(desc, producer) — and the caller hands that value to mappo_wide_l1_backward:
 * the library keeps no host-side record of workspaces (re-entrant per stream, safe across allocator address reuse). */
#define MAPPO_WIDE_LAYOUT_FEATURE_MAJOR 0   /* dz1 [64][B] | mean0 [B] | rstd0 [B]                       (K-chunked kernels)   */
#define MAPPO_WIDE_LAYOUT_BLOCKED 1         /* dz1 [tile][64][16] | padded statistics | z1 [B][64]        (16-sample-tile path) */
#define MAPPO_PRODUCER_MLP_BACKWARD 0       /* mappo_mlp_backward   */
#define MAPPO_PRODUCER_ACTOR_UPDATE 1       /* mappo_actor_update   */
#define MAPPO_PRODUCER_CRITIC_UPDATE 2      /* mappo_critic_update  */
#define MAPPO_PRODUCER_TRUNK_BACKWARD 3     /* mappo_trunk_backward */
int64_t mappo_wide_workspace_floats(int64_t B);
int32_t mappo_wide_l1_slabs(int64_t B);
int32_t mappo_wide_layout(const mappo_net_desc *desc /*host*/, int32_t producer /*MAPPO_PRODUCER_**/); /* layout id, or MAPPO_EINVAL */
int mappo_wide_l1_backward(const float *params, const mappo_net_desc *desc /*host*/, const float *x, const int32_t *rows,
                           int64_t B, const float *wide_ws, float *slabs, int64_t slab_stride, int64_t slab_col0,
                           int32_t layout /*mappo_wide_layout(desc, producer) of the launch that filled wide_ws*/,
                           mappo_stream_t stream);

/* ---- fused update kernels (K7 + K5 + K7-backward in ONE launch per network; r_mappo.py:91-164) --------------
 * The forward of a 32-sample tile, the loss gradient at the head and the backward pass run back to back in the
 * same wavefront: logits / values / d(logits) / d(values) never leave the CU.  Per-sample loss inputs are read
 * in buffer order through `rows` (NULL = identity), exactly like mappo_ppo_loss_fwd_bwd, whose arithmetic they
 * share.  Outputs: one gradient slab per workgroup (as mappo_mlp_backward) and per-workgroup double partial
 * sums `partials[n_slabs][4]` (actor: sum w*min(s1,s2), sum w*H, sum ratio, - ; critic: sum w_v*l, -, -, -),
 * turned into the 6 statistics by mappo_update_stats (actor_partials may be NULL when update_actor is off). */
int64_t mappo_update_partials_bytes(void);
int mappo_actor_update(const float *params, const mappo_net_desc *desc /*host*/, const float *obs,
                       const int32_t *rows, int64_t B, const float *avail /*or NULL*/, const float *actions,
                       const float *old_logp, const float *adv, const float *active,
                       const double *mb_moments /*[4]*/, const mappo_ppo_cfg *cfg /*host*/, float *slabs,
                       int64_t slab_stride, int64_t slab_col0, double *partials, float *wide_ws /*or NULL*/,
                       int32_t n_blocks /*0 = mappo_mlp_backward_slabs(B); else the grid size = slab rows written*/,
                       mappo_stream_t stream);
int mappo_critic_update(const float *params, const mappo_net_desc *desc /*host*/, const float *share_obs,
                        const int32_t *rows, int64_t B, const float *v_old, const float *returns,
                        const float *active, const float *vn_state /*[3] after update, or NULL*/,
                        const double *mb_moments /*[4]*/, const mappo_ppo_cfg *cfg /*host*/, float *slabs,
                        int64_t slab_stride, int64_t slab_col0, double *partials, float *wide_ws /*or NULL*/,
                        int32_t n_blocks, mappo_stream_t stream);
/* mappo_actor_update + mappo_critic_update in ONE launch (in_dim <= 64, shared layer_N / activation).  The networks run side
 * by side on disjoint CUs (shares proportional to their per-tile cost); each writes its own slab columns and loss partials.
 * mappo_dual_update_slabs = the slab rows / partial rows the caller provides PER NETWORK: every one of them is written by
 * the launch (a network with fewer workgroups has its remaining rows zero-filled), so the reduction runs over that count.
 * layer_N <= 1 and out_dim <= 16 take the one-wave-per-16-sample-tile kernel (csrc/mlp_upd16.h), the rest the pair kernel. */
int32_t mappo_dual_update_slabs(const mappo_net_desc *actor_desc /*host*/, const mappo_net_desc *critic_desc /*host*/, int64_t B);
int mappo_actor_critic_update(const float *actor_params, const mappo_net_desc *actor_desc /*host*/, const float *obs,
                              const float *critic_params, const mappo_net_desc *critic_desc /*host*/, const float *share_obs,
                              const int32_t *rows, int64_t B, const float *avail, const float *actions, const float *old_logp,
                              const float *adv, const float *active, const float *v_old, const float *returns,
                              const float *vn_state, const double *mb_moments, const mappo_ppo_cfg *cfg /*host*/, float *slabs,
                              int64_t slab_stride, int64_t actor_col0, int64_t critic_col0, double *actor_partials,
                              double *critic_partials, mappo_stream_t stream);
int mappo_update_stats(const double *actor_partials, int32_t n_actor /*workgroups that wrote them*/,
                       const double *critic_partials, int32_t n_critic, const double *mb_moments,
                       const mappo_ppo_cfg *cfg /*host*/, double *stats /*[6]*/,
                       double *acc /*[>=4] running sums of stats[0..3] over the updates of one train(), or NULL*/,
                       mappo_stream_t stream);

/* ---- K9: recurrent layer (onpolicy/algorithms/utils/rnn.py:7-80: nn.GRU(64,64) with per-step h*mask + LayerNorm) ----
 * A recurrent network = trunk (mappo_mlp_features) -> GRU -> rnn.norm -> head.  Outside the training pass (below: the
 * mappo_gru16_* entry points) features are feature-major: featT is [64][B], B = L*Nc, column t*Nc + c (time-major over Nc
 * sequences, the order of recurrent_generator's stacked chunks, shared_buffer.py:438-474); `rows` maps column -> flat buffer row.
 *   mappo_mlp_features   trunk forward, LayerNorm output of the last layer, feature-major
 *   mappo_gru_forward    L steps from h0[h0_rows] (row-major [.][64]); head_mode 0: states only, 1: head output
 *                        out[B][A] per step, 2: sample/argmax + log-prob (rollout, L = 1)
 *   mappo_trunk_backward backward of the trunk given dxT (forward recomputed per tile, as mappo_mlp_backward) */
int mappo_mlp_features(const float *params, const mappo_net_desc *desc /*host*/, const float *x, const int32_t *rows,
                       int64_t B, float *featT /*[64][B]*/, mappo_stream_t stream);
int mappo_gru_forward(const float *params, const mappo_net_desc *desc /*host*/, const float *featT, const float *h0,
                      const int32_t *h0_rows /*[Nc] or NULL*/, const float *masks, const int32_t *rows /*[B] or NULL*/,
                      int32_t L, int32_t Nc, float *h_last /*[Nc][64] or NULL*/,
                      int32_t head_mode, float *out, const float *avail /*[B][A] or NULL*/, int32_t deterministic,
                      uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions, float *logp,
                      mappo_stream_t stream);
/* One rollout step (L = 1) of a recurrent actor AND critic in one launch (r_actor_critic.py:43-70 + :146-165 on the same Nc rows):
 * actions / logp [Nc] sampled from the actor's head, values [Nc] from the critic's, next states to *_h_last [Nc][64]. */
int mappo_gru_step_dual(const float *actor_params, const mappo_net_desc *actor_desc /*host*/, const float *actor_featT /*[64][Nc]*/,
                        const float *actor_h0 /*[Nc][64]*/, float *actor_h_last, const float *critic_params,
                        const mappo_net_desc *critic_desc /*host*/, const float *critic_featT, const float *critic_h0,
                        float *critic_h_last, const float *masks /*[Nc]*/, int32_t Nc, const float *avail /*[Nc][A] or NULL*/,
                        int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions,
                        float *logp, float *values, mappo_stream_t stream);
/* The same step with the trunks included: obs / share_obs rows [Nc][in_dim] instead of precomputed features —
 * mappo_mlp_features_dual + mappo_gru_step_dual in ONE launch (r_actor_critic.py:43-70,146-165 end to end).  Both networks narrow
 * (in_dim <= 64, layer_N <= 1: every wave runs the tile's trunk from registers) or both wide (65..512, up to 16 384 rows: the
 * workgroup that ran a tile's split-K trunk goes straight on to its GRU step, the GRU operands fetched under the trunk). */
int mappo_recurrent_step_dual(const float *actor_params, const mappo_net_desc *actor_desc, const float *obs,
                              const float *actor_h0, float *actor_h_last, const float *critic_params,
                              const mappo_net_desc *critic_desc, const float *share_obs, const float *critic_h0,
                              float *critic_h_last, const float *masks, int32_t Nc, const float *avail, int32_t deterministic,
                              uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions, float *logp,
                              float *values, mappo_stream_t stream);
/* The SMAC rollout step in ONE launch (smac_runner.py:110-151 across two consecutive steps): the insert of the env output of step
 * k - 1 into buffer slot k (what mappo_insert_smac does: slot copies, masks / active_masks / bad_masks, states x (1 - env done)) AND
 * get_actions / get_values of step k, reading that env output (obs [N*M][D], share_obs [N*M][S], avail [N*M][A] or NULL, contiguous)
 * and the states the step before returned (actor_h / critic_h [N*M][64], unmasked) in place; the row mask is derived from `dones`
 * exactly as masks[k] is.  Next states go to *_h_next (not aliasing *_h); actions / logp / values are slot k's arrays. */
typedef struct mappo_smac_slot {
  float *obs, *share_obs, *available_actions /*or NULL*/, *rewards /*slot k - 1*/, *masks, *bad_masks, *active_masks, *rnn_states,
        *rnn_states_critic;
} mappo_smac_slot;
int mappo_recurrent_rollout_step(const float *actor_params, const mappo_net_desc *actor_desc /*host*/, const float *critic_params,
                                 const mappo_net_desc *critic_desc /*host*/, const float *obs, const float *share_obs, const float *avail,
                                 const float *rewards, int64_t rew_stride_n, int64_t rew_stride_m, const uint8_t *dones,
                                 int64_t done_stride_n, int64_t done_stride_m, const uint8_t *bad_transition /*[N*M] or NULL*/,
                                 const float *actor_h, const float *critic_h, float *actor_h_next, float *critic_h_next, int32_t N, int32_t M,
                                 int32_t deterministic, uint64_t seed, uint64_t counter, const uint64_t *counter_dev, float *actions,
                                 float *logp, float *values, const mappo_smac_slot *dst /*host*/, mappo_stream_t stream);
/* Trunk features of two networks (same layer_N / activation, in_dim <= 64) on the same B rows in one launch. */
int mappo_mlp_features_dual(const float *params_a, const mappo_net_desc *desc_a /*host*/, const float *x_a, float *featT_a,
                            const float *params_c, const mappo_net_desc *desc_c /*host*/, const float *x_c, float *featT_c,
                            int64_t B, mappo_stream_t stream);
int mappo_trunk_backward(const float *params, const mappo_net_desc *desc /*host*/, const float *x, const int32_t *rows,
                         int64_t B, const float *dxT /*[64][B]*/, float *slabs, int64_t slab_stride, int64_t slab_col0,
                         float *wide_ws /*or NULL*/, mappo_stream_t stream);

/* ---- K9, training pass on 16-sequence tiles (gru_train16.hip; rnn.py:25-79 inside r_mappo.py:91-164) -------------------------
 * One PPO update of a recurrent network on Nc sequences of L steps (time-major minibatch, column t*Nc + c) is
 *   mappo_mlp_features_seq   trunk features, tiled per (t, 16 sequences), BLOCKED per tile (in_dim <= 64 with layer_N <= 1, or in_dim 65..512
 *                            with Nc % 16 == 0; otherwise mappo_mlp_features, feature-major, with x_blocked = 0 below)
 *   mappo_gru16_forward_loss gi = W_ih x AND gh = W_hh (h mask) in one accumulator set, gates, h_t, then rnn.norm -> head -> PPO /
 *                            value loss -> their backward in the same registers: stores {h mask, r, z, n, W_hn h + b_hn, d h_t} per
 *                            step (gi and h_t never reach HBM), head + rnn.norm gradient columns of one slab row per workgroup,
 *                            loss partial sums [grid][4] (layout of mappo_actor_update's partials: mappo_update_stats reads them)
 *   mappo_gru16_backward     reverse time: d gates (in place over r, z, n, gh_n), carry = (W_hh^T d gh + d h z) mask, and
 *                            d x_t = W_ih^T d gi from the same registers (blocked over the d h component, or feature-major dxT)
 *   mappo_gru16_wgrad        dW_ih, dW_hh, db_ih, db_hh = sum over rows (d gates)^T (x | h mask) -> slab rows
 *   mappo_trunk_backward_seq backward of the trunk from the blocked d x (same shapes as mappo_mlp_features_seq; otherwise
 *                            mappo_trunk_backward with dxT)
 * scratch: mappo_gru16_scratch_floats(L, Nc) floats, [6][L][ceil(Nc/16)][4][64 lanes][4]: component c, step t, tile j, feature
 * block b; lane (n, q) holds features 16 b + 4 q + 0..3 of sequence 16 j + n (one contiguous KiB per wave access).  A blocked
 * feature array (x, d x) is one such component: mappo_gru16_blocked_floats(L, Nc).  mappo_gru16_slabs: slab rows written. */
int64_t mappo_gru16_scratch_floats(int32_t L, int32_t Nc);
int64_t mappo_gru16_blocked_floats(int32_t L, int32_t Nc);
int32_t mappo_gru16_slabs(int32_t L, int32_t Nc);
int mappo_mlp_features_seq(const float *params, const mappo_net_desc *desc /*host*/, const float *x, const int32_t *rows /*[L*Nc] or NULL*/,
                           int32_t L, int32_t Nc, float *out_blocked, mappo_stream_t stream);
int mappo_gru16_forward_loss(const float *params, const mappo_net_desc *desc /*host*/, const float *x, int32_t x_blocked, const float *h0,
                             const int32_t *h0_rows /*[Nc] or NULL*/, const float *masks, const int32_t *rows /*[L*Nc] or NULL*/,
                             int32_t L, int32_t Nc, int32_t head /*1 actor, 2 critic*/, const float *avail, const float *actions,
                             const float *old_logp, const float *adv, const float *active, const float *v_old, const float *returns,
                             const float *vn_state, const double *mb_moments, const mappo_ppo_cfg *cfg /*host*/, float *scratch,
                             float *slabs, int64_t slab_stride, int64_t slab_col0, double *partials, mappo_stream_t stream);
int mappo_gru16_backward(const float *params, const mappo_net_desc *desc /*host*/, const float *masks, const int32_t *rows, int32_t L,
                         int32_t Nc, float *scratch, float *dxT /*[64][L*Nc] feature-major, or NULL: blocked, in scratch component 5*/,
                         mappo_stream_t stream);
int mappo_gru16_wgrad(const mappo_net_desc *desc /*host*/, const float *x, int32_t x_blocked, const float *scratch, int32_t L, int32_t Nc,
                      float *slabs, int64_t slab_stride, int64_t slab_col0, mappo_stream_t stream);
int mappo_trunk_backward_seq(const float *params, const mappo_net_desc *desc /*host*/, const float *x, const int32_t *rows, int32_t L,
                             int32_t Nc, const float *dx_blocked, float *slabs, int64_t slab_stride, int64_t slab_col0,
                             float *wide_ws /*in_dim > 64 (then Nc % 16 == 0): mappo_wide_workspace_floats(L*Nc) floats, followed by
                                              mappo_wide_l1_backward with the layout of MAPPO_PRODUCER_TRUNK_BACKWARD; else NULL*/,
                             mappo_stream_t stream);

/* ---- K10/K11: slab reduction, global-norm clip, Adam (r_mappo.py:143-148,157-162; torch Adam) -------
 * The flat gradient covers `n_seg` parameter segments (actor, critic); norms/clip/lr are per segment.
 * opt_hyper (device floats, per segment, stride 8): {lr, beta1, beta2, eps, weight_decay, max_grad_norm,
 * use_clip, enabled}.  opt_step: device int32 per segment (Adam's `step`, incremented by the call).
 * grad_norms out: device floats per segment (pre-clip norm, what train_info logs). */
int64_t mappo_optim_workspace_bytes(int64_t P);
int mappo_slab_reduce(const float *slabs, int32_t n_slabs, int64_t slab_stride, int64_t P,
                      float *grad /*[P] out*/, mappo_stream_t stream);
int mappo_clip_adam(float *params, const float *grad, float *exp_avg, float *exp_avg_sq,
                    const int64_t *seg_bounds /*host, [n_seg+1], multiples of 256*/, int32_t n_seg,
                    const float *opt_hyper, int32_t *opt_step, float *grad_norms,
                    double *norm_acc /*[n_seg] running sums of the norms, or NULL*/, void *workspace,
                    mappo_stream_t stream);
/* mappo_slab_reduce + mappo_clip_adam in two launches instead of four (single-process training: nothing has to happen
 * between the reduction and the optimizer step).  Same arithmetic; the squared-norm partials are taken per 128
 * entries inside the reduction.  workspace: mappo_optim_workspace_bytes(P) bytes. */
int mappo_reduce_clip_adam(const float *slabs, int32_t n_slabs, int64_t slab_stride, float *params, float *grad /*out*/,
                           float *exp_avg, float *exp_avg_sq, const int64_t *seg_bounds /*host [n_seg+1]*/, int32_t n_seg,
                           const float *opt_hyper, int32_t *opt_step, float *grad_norms, double *norm_acc /*or NULL*/,
                           void *workspace, mappo_stream_t stream);

/* GPU-vectorised MPE simple_spread (csrc/mpe_env.hip): N environments x M agents x L landmarks, one launch per
 * step.  Replaces World.step / Scenario.reward / Scenario.observation / MultiAgentEnv.step + the vec-env's reset-on-done
 * (onpolicy/envs/mpe/core.py:207-322, scenarios/simple_spread.py:32-103, environment.py:117-256, env_wrappers.py:146-152).
 * State is caller-owned device memory: agent_pos / agent_vel [N][M][2] and landmark_pos [N][L][2] in float64 (the reference's
 * NumPy dtype), tstep [N] int32, episode [N] int64 (reset counter = Philox counter).  Outputs: obs [N][M][4+2L+4(M-1)] fp32,
 * rewards [N][M] fp32 (shared reward), dones [N][M] bool bytes.  action_mode 0: actions = the reference's actions_env
 * [N][M][5] (one-hot); 1: action indices [N][M] as fp32 (the buffer's own format).  An environment whose episode ends is
 * reset inside the same launch and returns the reset observation, as DummyVecEnv / SubprocVecEnv do. */
int mappo_mpe_spread_reset(double *agent_pos, double *agent_vel, double *landmark_pos, int32_t *tstep, int64_t *episode, float *obs,
                           int32_t N, int32_t M, int32_t L, uint64_t seed, mappo_stream_t stream);
int mappo_mpe_spread_step(double *agent_pos, double *agent_vel, double *landmark_pos, int32_t *tstep, int64_t *episode,
                          const float *actions, int32_t action_mode, float *obs, float *rewards, uint8_t *dones, int32_t N, int32_t M,
                          int32_t L, int32_t episode_length, uint64_t seed, mappo_stream_t stream);

/* ---- benchmark utility: the synthetic SMAC-shaped vec-env of bench.py / scripts (mappo_amd/envs/synthetic.py), one launch per
 * step.  Not a reference interface (the reference's envs are CPU processes, onpolicy/envs/starcraft2/StarCraft2_Env.py): it only
 * produces data of the SMAC shapes with agents that die and episodes that end.  obs [N][M][D], share_obs [N][M][S] ~ N(0,1);
 * avail [N][M][A] in {0,1}; rewards [N]; dead (state) / dones [N][M] bool bytes; counter_dev [34] = {counter, 33 tickets}:
 * counter keys the Philox stream and is advanced by the launch itself (last workgroup to finish), tickets must start at 0. */
/* P consecutive steps in one launch (pools [P][N][M][D] ... [P][N][M]; step p uses counter + p, the launch advances the counter by P):
 * the env hands the pool out step by step as views, as SyntheticMPEEnv does with its per-episode block. */
int mappo_synth_smac_pool(float *obs, float *share_obs, float *avail, float *rewards, uint8_t *dead, uint8_t *dones, int32_t P,
                          int32_t N, int32_t M, int32_t D, int32_t S, int32_t A, float p_death, float p_term, uint64_t seed,
                          uint64_t *counter_dev /*[34]*/, mappo_stream_t stream);
int mappo_synth_smac_step(float *obs, float *share_obs, float *avail, float *rewards, uint8_t *dead, uint8_t *dones,
                          int32_t N, int32_t M, int32_t D, int32_t S, int32_t A, float p_death, float p_term,
                          uint64_t seed, uint64_t *counter_dev /*[34]*/, mappo_stream_t stream);

/* ---- measurement hook (bench.py): the NEXT launch of the entry point's dominant kernel carries the two hipEvent_t
 * handles (hipExtLaunchKernelGGL: start / stop of that dispatch on its own stream); the hook disarms after one use. */
#define MAPPO_PROF_GAE 0
#define MAPPO_PROF_PPO_LOSS 1
#define MAPPO_PROF_MLP_FWD 2
#define MAPPO_PROF_MLP_BWD 3
#define MAPPO_PROF_SLAB_REDUCE 4
#define MAPPO_PROF_ADAM 5
#define MAPPO_PROF_ACT 6
#define MAPPO_PROF_COUNT 8
int mappo_profile_arm(int32_t kernel_id, void *ev_start /*hipEvent_t*/, void *ev_stop /*hipEvent_t*/);

/* ---- self test: fp32 MFMA operand/accumulator lane maps (used by tests/, not by the product path) --- */
int mappo_selftest_mfma(const float *A /*[32][2]*/, const float *Bm /*[2][32]*/, float *D /*[32][32]*/,
                        mappo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MAPPO_HIP_H */
