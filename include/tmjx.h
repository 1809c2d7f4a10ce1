/* include/tmjx.h — C-ABI of libtmjx_hip.so: the MI355X-native hot path of talmolab/track-mjx.
 *
 * The reference has no FFI/plugin boundary for this path (it is 100 % Python on JAX/XLA); the
 * entry points below are what a binding for the reference's Python interfaces would call.
 * Each entry cites the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - return 0 on success, a negative TMJX_E* code on failure; never throws; the message of the
 *     last failure on the calling thread is available from tmjx_last_error().
 *   - every per-env buffer is CALLER-OWNED DEVICE memory (e.g. a torch tensor's data_ptr) laid
 *     out structure-of-arrays with the ENV INDEX CONTIGUOUS:  buf[field_index * n_env + env].
 *   - the library owns only the immutable model constants and the clip table of a handle, plus ONE
 *     piece of per-handle device scratch: the physics kernel's per-env copy of the inertia matrix
 *     (4.5 KB per env for the rodent), allocated at the first launch and re-allocated only when a
 *     later launch has more envs (that one call synchronises the device; every other call enqueues only).
 *   - all work is enqueued on the caller's HIP stream; no hidden synchronisation (see above).
 *   - one handle per host thread / GPU AND per stream: launches of one handle must not overlap
 *     (thread-compatible, not thread-safe; pipelined env groups use one handle each).
 *   - `stream` is a hipStream_t passed as void* so that this header needs no HIP include.
 */
#ifndef TMJX_H
#define TMJX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMJX_OK 0
#define TMJX_EINVAL (-22)
#define TMJX_ENOMEM (-12)
#define TMJX_EHIP (-5)

typedef struct tmjx_model tmjx_model;

/* Float state layout (rows of the `state` buffer, each row n_env floats). Query with tmjx_layout(). */
typedef struct tmjx_layout_t {
  int32_t nq, nv, nu, nbody, ncon, nefc, obs_size, ref_obs_size, n_metrics, window;
  /* rows of the float state buffer */
  int32_t qpos, qvel, act, qacc_warmstart, time;          /* physics state (mjx.Data fields carried)      */
  int32_t xpos, xmat_torso, qfrc_actuator;                 /* outputs of the last mjx.forward read by K3   */
  int32_t prev_ctrl, action_buffer, done, steps_f;         /* task info (single_clip_tracking.py:227-234)  */
  int32_t first_phys, first_obs, first_prev_ctrl;          /* auto-reset snapshot (wrappers.py:93-95)       */
  int32_t state_rows;                                      /* total float rows                              */
  /* rows of the int32 state buffer */
  int32_t i_clip_idx, i_start_frame, i_buffer_index, i_nan_count, istate_rows;
  int32_t ws_rows;                                         /* float rows of the scratch workspace           */
} tmjx_layout_t;

/* Model handle from the compiled blob (model constants + env/task configuration).
 * Replaces: SingleClipTracking.__init__ (track_mjx/environment/task/single_clip_tracking.py:25-92:
 * solver options, mjcf load, mjx.put_model) and RewardConfig (task/reward.py:15-54). */
int tmjx_model_create(const void *blob, size_t nbytes, tmjx_model **out);
void tmjx_model_destroy(tmjx_model *m);
int tmjx_layout(const tmjx_model *m, tmjx_layout_t *out);

/* Episode / auto-reset wrapper semantics of the handle: wrappers.wrap(env, episode_length, ...) (track_mjx/environment/wrappers.py:18-56:
 * brax EpisodeWrapper's step counter / truncation at `episode_length`, and the (LSTM)AutoResetWrapperTracking restore on done).  Only
 * the two constants change; the clip table of the handle stays resident.  Blocking; not to be called with launches in flight. */
int tmjx_set_wrappers(tmjx_model *m, int episode_length, int auto_reset);
/* `action_repeat` of wrappers.wrap (track_mjx/environment/wrappers.py:21,43 -> brax EpisodeWrapper.step [3P]): tmjx_step then runs the
 * tracking env's own step `action_repeat` times with the same action (no termination check in between), returns the SUM of the repeats'
 * rewards, advances the episode's step counter by `action_repeat`, and takes observation / done / truncation / metrics from the last
 * repeat; the auto-reset follows the last repeat.  Default 1.  Host-side only (nothing is copied); tmjx_reward_obs, the K3-alone entry,
 * is one inner step whatever the repeat count. */
int tmjx_set_action_repeat(tmjx_model *m, int action_repeat);

/* Upload the ReferenceClip table (HOST pointers, float32, shapes (C,F,3) (C,F,4) (C,F,nq-7)
 * (C,F,nbody-1,3) (C,F,3)); the table becomes a resident device constant of the handle.
 * Replaces: the `reference_clip` constructor argument (task/multi_clip_tracking.py:16-72) whose
 * leaves are laid out by track_mjx/io/load.py:16-38,105-137. */
int tmjx_clips_upload(tmjx_model *m, const float *position, const float *quaternion, const float *joints,
                      const float *body_positions, const float *angular_velocity, int n_clips, int n_frames);

/* reset: MultiClipTracking.reset -> reset_from_clip (task/multi_clip_tracking.py:74-96,
 * task/single_clip_tracking.py:121-205) + the wrappers' reset (wrappers.py:88-102, brax
 * EpisodeWrapper.reset).  (clip_idx, start_frame, noise) are inputs because JAX's threefry stream
 * is outside this path; qpos_noise is [nq][n_env], qvel_noise [nv][n_env].  Writes obs [obs][n_env]. */
int tmjx_reset(tmjx_model *m, float *state, int32_t *istate, const int32_t *clip_idx, const int32_t *start_frame,
               const float *qpos_noise, const float *qvel_noise, float *obs, float *workspace, int n_env, void *stream);

/* step: wrappers.wrap(env).step = LSTMAutoResetWrapperTracking.step ∘ VmapWrapper ∘ EpisodeWrapper.step ∘
 * MultiClipTracking.step (wrappers.py:104-144; task/single_clip_tracking.py:207-320).
 * action [nu][n_env]; outputs obs [obs][n_env], reward/done/truncation [n_env], metrics [20][n_env]. */
int tmjx_step(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward,
              float *done, float *truncation, float *metrics, float *workspace, int n_env, void *stream);

/* The physics part of tmjx_step alone: n_frames substeps with the handle's configuration, through the same launches
 * (state -> env-major record, wave-per-env kernel, record -> state; the record lives in `workspace`).  bench.py times this. */
int tmjx_physics_step(tmjx_model *m, float *state, const float *action, float *workspace, int n_env, void *stream);

/* K2 alone: `n_substeps` x (ctrl = action; mjx.step) on the physics rows of `state`
 * (brax PipelineEnv.pipeline_step, called at task/single_clip_tracking.py:219). */
int tmjx_physics(tmjx_model *m, float *state, const float *action, int n_substeps, float *workspace, int n_env,
                 void *stream);
/* mjx.forward alone on the physics rows of `state` (pipeline_init, task/single_clip_tracking.py:163). */
int tmjx_forward(tmjx_model *m, float *state, float *workspace, int n_env, void *stream);

/* K3 alone: everything in step() except the physics substeps: frame index, clip gather, rewards,
 * observation, done/NaN guard, episode + auto-reset wrappers (task/single_clip_tracking.py:220-320,
 * task/reward.py:359-485, wrappers.py:104-144).
 * `workspace` (>= 2*nu rows) receives the per-(action dim, env) window partials; NULL computes them inline. */
int tmjx_reward_obs(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward,
                    float *done, float *truncation, float *metrics, float *workspace, int n_env, void *stream);

/* compute_tracking_rewards as the reference calls it (task/reward.py:359-366; call site task/single_clip_tracking.py:239-246): the
 * reward / termination part of K3 with the CALLER's gathered reference frame per env instead of the handle's own gather from the resident
 * clip table.  frame_* are device pointers, row-major per env: position [n][3], quaternion [n][4], joints [n][nq-7], body_positions
 * [n][nbody-1][3], angular_velocity [n][3] (ReferenceClip leaves of one frame, track_mjx/io/load.py:16-38).  Everything else as
 * tmjx_reward_obs with workspace = NULL (the observation's trajectory part still reads the resident table). */
int tmjx_reward_frame(tmjx_model *m, float *state, int32_t *istate, const float *action, const float *frame_pos,
                      const float *frame_quat, const float *frame_joints, const float *frame_bodypos, const float *frame_angvel,
                      float *obs, float *reward, float *done, float *truncation, float *metrics, int n_env, void *stream);

/* GAE reverse scan: compute_gae (track_mjx/agent/mlp_ppo/losses.py:39-100). All arrays [T][B] row-major
 * (B contiguous), bootstrap [B]; outputs vs, advantages [T][B]. */
int tmjx_gae(const float *truncation, const float *termination, const float *rewards, const float *values,
             const float *bootstrap, float lambda_, float discount, float *vs, float *advantages, int T, int B,
             void *stream);

/* PPO loss head, forward and gradients w.r.t. the network outputs in one call: compute_ppo_loss
 * (track_mjx/agent/mlp_ppo/losses.py:103-245) from the policy logits on: NormalTanh log-prob / entropy, GAE
 * (losses.py:39-100), advantage normalisation, clipped surrogate, value loss, AR(1)-prior latent KL.
 * Row-major device arrays: logits [T][B][2A], raw_action / noise [T][B][A], behaviour_logp / baseline / reward /
 * discount / truncation [T][B], bootstrap [B], fc2 = latent mean | logvar [T][B][2Z]; outputs dlogits, dbaseline,
 * dfc2 (same shapes as their inputs) = d total / d input; out[8] = total, policy_loss, v_loss, entropy_loss,
 * kl_latent_loss, advantage mean, advantage std, entropy; scratch >= tmjx_ppo_scratch_floats(T, B) floats. */
typedef struct {
  int32_t T, B, A, Z;
  float reward_scaling, discounting, gae_lambda, clip_eps, entropy_cost, kl_weight;
  int32_t normalize_advantage;
  int32_t accumulate;              /* 1: out[0..7] += this call's values (a running sum over the minibatch steps of an update), 0: out = values */
} tmjx_ppo_cfg_t;
int tmjx_ppo_scratch_floats(int T, int B);
int tmjx_ppo_loss(const tmjx_ppo_cfg_t *cfg, const float *logits, const float *raw_action, const float *behaviour_logp,
                  const float *noise, const float *baseline, const float *bootstrap, const float *reward, const float *discount,
                  const float *truncation, const float *fc2, float *dlogits, float *dbaseline, float *dfc2, float *scratch,
                  float *out, void *stream);

/* The same loss head in phases, for a caller that runs the policy and the value network on two streams: A (log-prob, entropy, KL sums) reads the policy's
 * outputs only, B (GAE, advantage statistics, value loss) the value network's only, C (surrogate + every gradient) both, D writes out[8].  Each call launches
 * the phases of its mask on its stream; the caller orders them (A, B before C; C before D).  unroll_length T <= 24.  Gradients: the bits of tmjx_ppo_loss;
 * the entropy / KL scalars are added up in another order. */
#define TMJX_PPO_PHASE_A 1
#define TMJX_PPO_PHASE_B 2
#define TMJX_PPO_PHASE_C 4
#define TMJX_PPO_PHASE_D 8
int tmjx_ppo_loss_phases(const tmjx_ppo_cfg_t *cfg, const float *logits, const float *raw_action, const float *behaviour_logp,
                         const float *noise, const float *baseline, const float *bootstrap, const float *reward, const float *discount,
                         const float *truncation, const float *fc2, float *dlogits, float *dbaseline, float *dfc2, float *scratch,
                         float *out, int phases, void *stream);

/* Dense -> SiLU -> LayerNorm epilogue of the intention network's hidden layers (intention_network.py:14-88):
 * y = LayerNorm_{gamma,beta,eps}(silu(z + bias)), z [rows][H] = the GEMM output, H in {64, 128, 256, 512, 1024}.
 * fwd also writes stats [rows][2] = mean, rstd; bwd returns dz and grads [3][H] = d_gamma | d_beta | d_bias;
 * partial >= tmjx_silu_ln_partial_floats(rows, H) floats of scratch. */
int tmjx_silu_ln_partial_floats(int rows, int H);
int tmjx_silu_ln_fwd(const float *z, const float *bias, const float *gamma, const float *beta, float *y, float *stats, int rows, int H,
                     float eps, void *stream);
int tmjx_silu_ln_bwd(const float *dy, const float *z, const float *bias, const float *gamma, const float *stats, float *dz, float *grads,
                     float *partial, int rows, int H, void *stream);
/* The same two kernels with a bf16 result (y16 / dz16: [rows][ld >= H], round to nearest even) for consumers that are bf16-operand GEMMs
 * (tmjx_bgemm_*: the 1024-wide first encoder block of the rodent-mc-intention nets, which is not one GEMM tile wide): the bits those GEMMs would have
 * made of the fp32 result when they stage it, without the fp32 round trip through memory.  Statistics and column sums stay fp32. */
int tmjx_silu_ln_fwd_bf16(const float *z, const float *bias, const float *gamma, const float *beta, uint16_t *y16, int ldy16, float *stats, int rows, int H,
                          float eps, void *stream);
int tmjx_silu_ln_bwd_bf16(const float *dy, const float *z, const float *bias, const float *gamma, const float *stats, uint16_t *dz16, int lddz16, float *grads,
                          float *partial, int rows, int H, void *stream);

/* Minibatch gather fused with the observation normaliser (ppo.py:306-311 + running_statistics.normalize):
 * out[t][b][:] = (src[t][idx[b]][:] - mean) / std; src [T][R][W], idx int64 [B], out [T][B][W], W % 4 == 0. */
int tmjx_gather_normalize(const float *src, const int64_t *idx, const float *mean, const float *std, float *out, int T, int R, int B,
                          int W, void *stream);
/* The whole minibatch of one SGD step in one launch (ppo.py:304-317: the same permutation slice indexes every leaf of the roll-out data):
 * obs_n [T][B][W] = (obs[t][idx[b]] - mean) / std, next_n [B][W] likewise from next_last [R][W], raw_action_g [T][B][A] = raw_action[t][idx[b]],
 * scalars_g [4][T][B] = (log_prob, reward, discount, truncation)[t][idx[b]].  idx: int64 [B] on the device, values in [0, R). */
int tmjx_gather_minibatch(const float *obs, const float *next_last, const float *raw_action, const float *log_prob, const float *reward, const float *discount,
                          const float *truncation, const int64_t *idx, const float *mean, const float *std, float *obs_n, float *next_n, float *raw_action_g,
                          float *scalars_g, int T, int R, int B, int W, int A, void *stream);
/* The same launch as the first kernel of a SELF-ADVANCING SGD step (a captured hipGraph replays it without any host input): the rows are
 * perm[slot B .. slot B + B) with slot = state[1]; the launch also draws the step's two N(0, 1) arrays — eps [T B Z] (the latent sample of
 * intention_network.py:78-88) and noise [T B A] (the entropy sample of losses.py:222) — from Philox4x32-10 keyed by `seed`, counter state[0],
 * and, with `advance`, adds 1 to state[0] and state[1] once every workgroup is done (state: device int64[TMJX_MINIBATCH_STATE_WORDS], {draw counter, slot, 0, 0, ...}:
 * words 2 .. are the launch's completion tickets and must be zero).
 * The host resets state[1] and refills `perm` once per epoch (ppo.py:304-311: one permutation per update over the batch). */
#define TMJX_MINIBATCH_STATE_WORDS (16 + 16 * 64)
typedef struct tmjx_minibatch_t {
  const float *obs, *next_last, *raw_action, *log_prob, *reward, *discount, *truncation;
  const int64_t *perm;
  const float *mean, *std;
  float *obs_n, *next_n, *raw_action_g, *scalars_g;
  float *eps, *noise;            /* may be NULL */
  int64_t *state;                /* may be NULL when eps, noise are NULL and advance == 0 (then slot = 0) */
  uint64_t seed;
  int32_t T, R, B, W, A, Z, advance;
} tmjx_minibatch_t;
int tmjx_minibatch_begin(const tmjx_minibatch_t *mb, void *stream);
/* The same launch with a bf16 TWIN of obs_n next to it (bf16 GEMM-input mode): obs_n16 [T B][ld16 >= W], the normalised observation rounded to
 * nearest even — the operand the first layers' bf16 GEMMs (tmjx_bgemm_*) would have made of obs_n when they stage it.  Columns W .. ld16 - 1 are
 * not written (the caller keeps them zero, so that a contraction may run to the next multiple of 64 by LDS-DMA). */
int tmjx_minibatch_begin_bf16(const tmjx_minibatch_t *mb, uint16_t *obs_n16, int ld16, void *stream);
/* Philox4x32-10 of (counter[4], key[2]) = six device words -> four device words: the generator above, exposed for its known-answer test */
int tmjx_philox4x32_10(const uint32_t *ctr_key_dev, uint32_t *out_dev, void *stream);

/* Backward of the latent sample inside tmjx_latent_concat (reparameterize, intention_network.py:78-88): from d x [n][dx_stride]
 * to d fc2 [n][2 Z] = [d mean | d logvar]. */
int tmjx_latent_concat_bwd(const float *dx, const float *eps, const float *fc2, float *dfc2, int n, int Z, int dx_stride, void *stream);
/* The same with a second gradient of fc2 summed in (`add` [n][2 Z]: the KL term's, compute_ppo_loss's kl_latent_loss, losses.py:196-237): d fc2 =
 * add + the sample's gradient, as jax.grad sums the two uses of the encoder's output (intention_network.py:128-139). */
int tmjx_latent_concat_bwd_add(const float *dx, const float *eps, const float *fc2, const float *add, float *dfc2, int n, int Z, int dx_stride,
                               void *stream);

/* Policy inference tails (ppo_networks.py:46-96, intention_network.py:78-88).
 * tmjx_latent_concat: x[i] = [ mean_i + eps_i * exp(logvar_i / 2) | obs_i[ref_w:] ], fc2 [n][2Z] = mean | logvar, eps [n][Z], obs
 *   addressed as obs[i * obs_s0 + c * obs_s1] (so the [obs][n_env] buffer of tmjx_step can be passed as is) and normalised with
 *   (. - mean[c]) / std[c] when mean != NULL, x [n][x_stride >= Z + obs_w - ref_w] (columns beyond are written as zeros).
 * tmjx_sample_action: raw = loc + (softplus(raw_scale) + 0.001) * noise; action = tanh(raw) written as [A][n] (the layout
 *   tmjx_step takes); logp = NormalTanh log-prob of the sample; logits [n][2A], noise / raw [n][A], logp [n].
 * eps == NULL / noise == NULL: the N(0, 1) draws are made on the device, Philox4x32-10 streams 2 / 3 of (seed, draw counter rng_state[0]);
 *   rng_state = device int64[2] {draw counter, 0}; tmjx_sample_action then advances the counter once all its workgroups are done, so one
 *   inference = tmjx_latent_concat ... tmjx_sample_action on the same stream uses one counter value and a captured graph of it replays
 *   with fresh noise and no host input. */
int tmjx_latent_concat(const float *fc2, const float *eps, const float *obs, float *x, int n, int Z, int obs_w, int ref_w,
                       int64_t obs_s0, int64_t obs_s1, const float *mean, const float *std, int x_stride, uint64_t seed, const int64_t *rng_state,
                       void *stream);
int tmjx_sample_action(const float *logits, const float *noise, float *raw, float *action_t, float *logp, int n, int A, uint64_t seed,
                       int64_t *rng_state, void *stream);

/* The acting policy's dense layer (brax acting.actor_step through make_inference_fn, track_mjx/agent/mlp_ppo/ppo_networks.py:46-96; layers of
 * intention_network.py:32-44,68-76) through a 20 KB LDS tile — what the CUs have free next to twelve resident physics workgroups since round 5:
 * C[M][N] = op(A) W^T (+ bias), A row-major [M][lda], W [N][ldw], op = identity or (A - mean[k]) * inv_std[k] (mean / inv_std: both or neither).
 * tmjx_linear_act_ok: 1 when the operands qualify (K % 4 == 0, 16-byte aligned rows); otherwise use tmjx_linear_nolds. */
int tmjx_linear_act_ok(const float *A, int64_t lda, const float *W, int ldw, int K);
int tmjx_linear_act(const float *A, int64_t lda, const float *W, int ldw, const float *bias, float *C, int M, int N, int K,
                    const float *mean, const float *inv_std, void *stream);
/* LDS-free dense layer (policy inference next to the physics kernel, which owns every CU's LDS):
 * C[M][N] = A W^T + bias (bias may be NULL); W [N][K] row-major; A[i][k] at A[i * sa_row + k * sa_k] with either sa_k == 1
 * (row-major activations) or sa_row == 1 (the [obs][n_env] buffer; M % 4 == 0 and 16-byte aligned columns required). */
int tmjx_linear_nolds(const float *A, int64_t sa_row, int64_t sa_k, const float *W, const float *bias, float *C, int M, int N, int K,
                      void *stream);
/* The same with the operand normalised while it is loaded: C = ((A - mean[k]) * inv_std[k]) W^T + bias (mean, inv_std: [K] device vectors) — the
 * acting policy's first layer reading the env's RAW observation buffer (brax running_statistics.normalize, track_mjx/agent/mlp_ppo/ppo_networks.py:46-60).
 * Matrix-core variant only: K % 4 == 0, 16-byte aligned operands. */
int tmjx_linear_nolds_norm(const float *A, int64_t sa_row, int64_t sa_k, const float *W, const float *bias, float *C, int M, int N, int K,
                           const float *mean, const float *inv_std, void *stream);

/* The LDS-free dense layer in bf16 GEMM-input mode (BASELINE config 5): the fp32 activations are rounded to bf16 in registers, W is the layer's
 * resident bf16 shadow [N][ldw] (tmjx_bf16_shadow: rows zero padded to ldw, a multiple of 64), fp32 accumulate on v_mfma_f32_16x16x32_bf16 — the
 * operand rounding of the learner's tmjx_bgemm_* forward pass, so that the roll-out's behaviour log-prob and the learner's first-pass log-prob
 * agree (reference: one jitted policy serves both, track_mjx/agent/mlp_ppo/ppo_networks.py:46-96).  mean / inv_std: NULL or the operand's
 * normaliser as in tmjx_linear_nolds_norm. */
int tmjx_linear_nolds_bf16(const float *A, int64_t sa_row, int64_t sa_k, const uint16_t *W, int ldw, const float *bias, float *C, int M, int N, int K,
                           const float *mean, const float *inv_std, void *stream);

/* out[width] = column sums of the row-major src[rows][width] (the bias gradient dy.sum(0) of a dense layer: flax nn.Dense's bias in
 * track_mjx/agent/mlp_ppo/intention_network.py:32-44 and brax's value MLP); `scratch`: tmjx_colsum_scratch_floats(width) floats. */
int tmjx_colsum_scratch_floats(int width);
int tmjx_colsum(const float *src, float *out, float *scratch, int rows, int width, void *stream);

/* Up to 16 independent column sums (out[width] = sum over `rows` rows of partial[rows][width]) in one launch: the per-workgroup
 * (d gamma | d beta | d bias) partials of every LayerNorm block of a backward pass (tmjx_gemm_nn_ln_bwd), reduced once behind it.
 * `problems` is a HOST array (copied into the launch). */
typedef struct tmjx_colsum_problem_t { const float *partial; float *out; int32_t rows, width; } tmjx_colsum_problem_t;
int tmjx_colsum_grouped(const tmjx_colsum_problem_t *problems, int n, void *stream);

/* optax.chain(optax.clip_by_global_norm(max_norm), optax.adam(lr)) (track_mjx/agent/mlp_ppo/ppo.py:517-520) on FLAT fp32 device
 * buffers of n elements: param -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps) with the gradient scaled by max_norm / max(max_norm,
 * *grad_norm); `grad_norm` is a device scalar (the caller's ||grad||_2 of the averaged gradient), bias_correction{1,2} = 1 - beta^t. */
int tmjx_adam_clip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *grad_norm, long long n, float lr,
                   float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, float max_norm, void *stream);
/* The same step with the global norm computed by the library itself: tmjx_adam_norm_floats() per-workgroup sums of squares (fixed
 * summation order: bit-identical on every rank for the same averaged gradient) into norm_scratch, added up inside the Adam kernel;
 * norm_out (may be NULL) receives ||grad||_2.  grad 16-byte aligned. */
int tmjx_adam_norm_floats(void);
int tmjx_adam_clip_norm(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *norm_scratch, float *norm_out, long long n, float lr,
                        float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, float max_norm, void *stream);

/* Dense layers of the learner on the matrix cores, fp32 in / fp32 accumulate (flax nn.Dense of the intention network,
 * track_mjx/agent/mlp_ppo/intention_network.py:32-44,68-76, and brax's value MLP, ppo_networks.py:180-184; the gradients are those of
 * compute_ppo_loss, losses.py:103-245).  Row-major device matrices with leading dimensions (floats); any sizes; 16-byte aligned rows
 * (pointer and leading dimension multiples of 4 floats) take the vector-load path.
 *   tmjx_gemm_nt: C[M][N] = A[M][K] W[N][K]^T + bias[N] (bias may be NULL)      y = x W^T + b
 *   tmjx_gemm_nn: C[M][N] = A[M][K] W[K][N]                                     dx = dy W
 *   tmjx_gemm_dw: dW[N][K] = dY[M][N]^T X[M][K] (dW dense, leading dimension K) and, if db != NULL, db[N] = column sums of dY;
 *                 scratch >= tmjx_gemm_dw_scratch_floats(M, N, K) floats (row-range slabs, reduced by a second launch). */
int tmjx_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc, int M, int N, int K, void *stream);
int tmjx_gemm_nn(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int M, int N, int K, void *stream);
/* One Dense -> SiLU -> LayerNorm block of the intention network (intention_network.py:32-40,68-74) forward in ONE launch, for layers exactly
 * one tile wide (N = 64, 128 or 256) with 16-byte aligned operand rows (tmjx_gemm_nt_silu_ln_ok says whether a call qualifies):
 *   Z = A W^T (no bias: the operand of tmjx_silu_ln_bwd), Y = LayerNorm(silu(Z + bias)) * gamma + beta, stats[row] = (mean, 1 / std).
 * Same arithmetic as tmjx_gemm_nt followed by tmjx_silu_ln_fwd. */
int tmjx_gemm_nt_silu_ln_ok(const float *A, int lda, const float *W, int ldw, int N);
int tmjx_gemm_nt_silu_ln(const float *A, int lda, const float *W, int ldw, const float *bias, const float *gamma, const float *beta, float *Z, float *Y,
                         int ldc, float *stats, int M, int N, int K, float eps, void *stream);
/* The input gradient of a layer whose INPUT is the output of a Dense -> SiLU -> LayerNorm block of width N = 256, with that block's LayerNorm +
 * SiLU backward applied in the epilogue (tmjx_gemm_nn followed by tmjx_silu_ln_bwd in one launch): dz[M][256] = d loss / d z of the block,
 * given dY[M][K] (gradient of this layer's output), W[K][256], and the block's saved z (without bias), bias, gamma, stats (mean, 1 / std).
 * partial (>= tmjx_gemm_nn_ln_bwd_partial_floats(M, 256) floats) receives one row of [d gamma | d beta | d bias] column sums per 80-row
 * workgroup: sum its rows (tmjx_colsum over (M + 79) / 80 rows of width 768). */
int tmjx_gemm_nn_ln_bwd_ok(const float *dY, int ldy, const float *W, int ldw, int N);
long long tmjx_gemm_nn_ln_bwd_partial_floats(int M, int N);
int tmjx_gemm_nn_ln_bwd(const float *dY, int ldy, const float *W, int ldw, const float *z, const float *bias, const float *gamma, const float *stats,
                        float *dz, float *partial, int M, int N, int K, void *stream);
/* Dense -> SiLU (brax value MLP, track_mjx/agent/mlp_ppo/ppo_networks.py:180-184: swish activations, no LayerNorm).  tmjx_gemm_nt_silu: Z[M][ldc] = A W^T
 * (WITHOUT the bias) and Y = silu(Z + bias) in one launch (operand rows 16-byte aligned: tmjx_gemm_nt_silu_ok); tmjx_silu_fwd: the activation alone;
 * tmjx_silu_bwd: dz = dy silu'(z + bias), dense [rows][N] arrays. */
int tmjx_gemm_nt_silu_ok(const float *A, int lda, const float *W, int ldw);
int tmjx_gemm_nt_silu(const float *A, int lda, const float *W, int ldw, const float *bias, float *Z, float *Y, int ldc, int M, int N, int K, void *stream);
int tmjx_silu_fwd(const float *z, const float *bias, float *y, long long rows, int N, void *stream);
int tmjx_silu_bwd(const float *dy, const float *z, const float *bias, float *dz, long long rows, int N, void *stream);
/* The backward pass of the value MLP without an element-wise launch between its GEMMs (brax make_value_network, ppo_networks.py:180-184: Dense -> swish ... Dense(1)):
 * tmjx_gemm_nn_silu_bwd: dZ[M][N] = (dY[M][K] W[K][N]) silu'(z + bias) — the input gradient of a hidden layer's consumer with THAT layer's SiLU backward on the
 *   accumulators (z, dZ dense [M][N]; operand rows 16-byte aligned: tmjx_gemm_nn_silu_bwd_ok);
 * tmjx_silu_bwd_rank1: dz[m][k] = (dy1[m] w1[k]) silu'(z[m][k] + bias[k]) — the same for the 1-wide head, whose input gradient is an outer product;
 * tmjx_head_dw: the head's gradients dw[k] = sum_m dy1[m] x[m][k], db[0] = sum_m dy1[m] (a matrix-vector product; scratch >= tmjx_head_dw_scratch_floats). */
int tmjx_gemm_nn_silu_bwd_ok(const float *dY, int ldy, const float *W, int ldw);
int tmjx_gemm_nn_silu_bwd(const float *dY, int ldy, const float *W, int ldw, const float *z, const float *bias, float *dZ, int M, int N, int K, void *stream);
int tmjx_silu_bwd_rank1(const float *dy1, const float *w1, const float *z, const float *bias, float *dz, long long rows, int N, void *stream);
long long tmjx_head_dw_scratch_floats(int M, int K);
int tmjx_head_dw(const float *dy1, const float *x, int ldx, float *dw, float *db, float *scratch, int M, int K, void *stream);
/* ---- whole-chain kernels (csrc/mlp_chain.h) for nets whose hidden layers are exactly 256 wide (BASELINE configs[1] / configs[2]: encoder, decoder and
 * critic = [256, 256]).  ONE launch runs a whole chain of the reference's layers — the encoder's Dense -> silu -> LayerNorm blocks + fc2_mean | fc2_logvar,
 * the decoder's blocks + the action head (track_mjx/agent/mlp_ppo/intention_network.py:32-44,68-76,128-139), or brax's value MLP
 * (ppo_networks.py:180-184) — on one workgroup's row tile: a layer's output goes to global memory exactly as the layer-by-layer entry points
 * leave it (tmjx_gemm_nt_silu_ln / tmjx_gemm_nt_silu: z without the bias, y, stats = (mean, 1 / std) per row) AND stays on the CU as the next layer's
 * operand.  Results are bit-identical to tmjx_gemm_nt_silu_ln / tmjx_gemm_nt_silu / tmjx_gemm_nt / tmjx_head_fwd called layer by layer.
 *   epi 1: hidden layers are Dense -> SiLU -> LayerNorm blocks;  epi 3: Dense -> SiLU layers (no gamma / beta / stats).
 *   Wf != NULL: an un-activated last layer outf[M][Nf] = y_last Wf^T + bf with Nf <= 128 (fc2, the action head); Nf == 1 (epi 3): the value
 *   head as a dot product (Wf = its weight row of 256 floats, outf[M]).  hidden[0].K = the input width (<= lda); hidden[l > 0].K = 256.
 * All rows 16-byte aligned (tmjx_chain_fwd_ok says whether a call qualifies); z, y dense [M][256]. */
#define TMJX_CHAIN_MAX_HIDDEN 4
typedef struct { const float *W, *bias, *gamma, *beta; float *z, *y, *stats; int32_t K, ldw; } tmjx_chain_layer_t;
typedef struct {
  const float *A; int32_t lda, M, n_hidden, epi;
  tmjx_chain_layer_t hidden[TMJX_CHAIN_MAX_HIDDEN];
  const float *Wf, *bf; float *outf; int32_t Nf, ldwf, ldof;
  float eps;
  int32_t rows_alloc;   /* rows every y buffer holds: >= tmjx_chain_rows(M) (a y that feeds the next layer of the launch is stored as whole row tiles) */
  /* latent tail (lat_out != NULL; epi 1, Nf = 2 lat_Z: the encoder + fc2_mean | fc2_logvar): the launch also writes the decoder's input,
   * lat_out[M][lat_ld] = [ mean + lat_eps * exp(logvar / 2) (lat_Z columns) | prop[M][prop_w] (row stride prop_ld) ] — reparameterize + the decoder-input
   * concat of intention_network.py:84-88,128-139, what tmjx_latent_concat computes as a launch of its own (same bits) */
  const float *lat_eps; float *lat_out; const float *prop; int32_t lat_Z, lat_ld, prop_w, prop_ld;
  void *prof;     /* NULL; or 16 uint64 per workgroup ((M + rows per tile - 1) / rows per tile workgroups): in-kernel clock stamps (tools/chain_stamps.py) */
} tmjx_chain_fwd_t;
int tmjx_chain_rows(int M);      /* M rounded up to the row tile (80 or 32 rows) the chain kernels take for M rows */
int tmjx_chain_fwd_ok(const tmjx_chain_fwd_t *chain);
int tmjx_chain_fwd(const tmjx_chain_fwd_t *chain, void *stream);
/* The backward pass of such a chain in ONE launch: the input-gradient GEMMs from the last layer's output gradient G[M][Kg] down to the first
 * hidden layer, each with the PRODUCING block's backward in its epilogue (what tmjx_gemm_nn_ln_bwd / tmjx_gemm_nn_silu_bwd do layer by layer),
 * d loss / d z of every hidden layer written to global memory (dz: the operand of the weight gradients, tmjx_gemm_dw_grouped) and kept on the CU as
 * the next GEMM's operand.  stage[0]: W = the LAST layer's weight [Kg][256], (z, bias, gamma, stats) = the last hidden block's saved tensors;
 * stage[s > 0]: W = the weight [256][256] of hidden layer (n - s), block = hidden layer (n - 1 - s).  epi 2: Dense -> SiLU -> LayerNorm blocks
 * (partial: >= tmjx_gemm_nn_ln_bwd_partial_floats(M, 256) floats per stage, one row of [d gamma | d beta | d bias] per workgroup: reduce with
 * tmjx_colsum_grouped);  epi 4: Dense -> SiLU layers.  Kg == 1 (epi 4): the value head — G = dy1[M], stage[0].W = the head's weight row, stage 0 has
 * no GEMM (tmjx_silu_bwd_rank1's expression).  W0 != NULL: a trailing GEMM dx[M][dx_cols] = dz_first W0[256][:dx_cols] (the decoder's first block:
 * d loss / d latent, dx_cols <= 128).  Bit-identical to the layer-by-layer entry points. */
typedef struct { const float *W; int32_t ldw; const float *z, *bias, *gamma, *stats; float *dz, *partial; } tmjx_chain_bwd_stage_t;
typedef struct {
  const float *G; int32_t ldg, Kg, M, n_stages, epi;
  tmjx_chain_bwd_stage_t stage[TMJX_CHAIN_MAX_HIDDEN];
  const float *W0; int32_t ldw0, dx_cols; float *dx; int32_t lddx;
  int32_t rows_alloc;   /* rows every dz buffer holds: >= tmjx_chain_rows(M) */
  void *prof;
} tmjx_chain_bwd_t;
int tmjx_chain_bwd_ok(const tmjx_chain_bwd_t *chain);
int tmjx_chain_bwd(const tmjx_chain_bwd_t *chain, void *stream);

/* The 1-wide head's forward pass y[m] = x[m][:K] . w + bias[0] as a matrix-vector product (bias may be NULL; K % 4 == 0, 16-byte aligned rows: tmjx_head_fwd_ok). */
int tmjx_head_fwd_ok(const float *x, int ldx, const float *w, int K);
int tmjx_head_fwd(const float *x, int ldx, const float *w, const float *bias, float *y, int M, int K, void *stream);
long long tmjx_gemm_dw_scratch_floats(int M, int N, int K);
int tmjx_gemm_dw(const float *dY, int ldy, const float *X, int ldx, float *dW, float *db, float *scratch, int M, int N, int K, void *stream);
/* All weight (+ bias) gradients of one backward pass as ONE launch + one reduction launch: up to 16 independent problems of tmjx_gemm_dw,
 * each with its own scratch (>= tmjx_gemm_dw_scratch_floats(M, N, K) floats) and a leading dimension `lddw` for dW (so the result can land
 * in a row-padded view of a flat gradient buffer); dY and X need 16-byte aligned rows.  `problems` is a HOST array (copied into the launch). */
typedef struct tmjx_dw_problem_t {
  const float *dY, *X;
  float *dW, *db, *scratch;       /* db may be NULL */
  int32_t ldy, ldx, lddw, M, N, K;
} tmjx_dw_problem_t;
int tmjx_gemm_dw_grouped(const tmjx_dw_problem_t *problems, int n, void *stream);
/* The same with the group's workgroup budget stated by the caller (0 = the default, 1024): the slab count of every problem is capped so that the
 * whole launch is about `target_wgs` workgroups — a group launched NEXT TO other kernels (the value network's weight gradients behind its own
 * backward pass, while the policy's still runs: agent/ppo.py) asks for fewer. */
int tmjx_gemm_dw_grouped_wgs(const tmjx_dw_problem_t *probs, int n, int target_wgs, void *stream);

/* ---- bf16 GEMM-input mode (BASELINE config 5 "bf16 MLP on MFMA"; layers: track_mjx/agent/mlp_ppo/intention_network.py:14-142, sizes
 * track_mjx/config/rodent-full-clips.yaml:50-57): operands bf16, accumulation and results fp32, v_mfma_f32_16x16x32_bf16 (csrc/gemm_bf16.h).
 * Activations are either fp32 (converted when they are staged: no cast pass) or bf16 written by a producing kernel; weights come from bf16
 * SHADOWS of the fp32 master parameters. */
/* Shadows of up to 24 weight matrices src[N][K] (fp32, leading dimension ld_src) in ONE launch: dst[N][ld_dst] (ld_dst >= ceil64(K), zero
 * beyond K: the forward operand) and / or dst_t[K][ld_dst_t] (ld_dst_t >= ceil64(N), zero beyond N: the input gradient's operand).  bf16 =
 * round to nearest even of the fp32 value.  `items` is a HOST array (copied into the launch). */
typedef struct tmjx_bf16_shadow_t {
  const float *src;
  uint16_t *dst, *dst_t;           /* either may be NULL */
  int32_t N, K, ld_src, ld_dst, ld_dst_t;
} tmjx_bf16_shadow_t;
int tmjx_bf16_shadow(const tmjx_bf16_shadow_t *items, int n, void *stream);
/* C[M][N] = A[M][K] . B[N][K]^T (+ bias[N]):  A fp32 (a_is_f32 = 1; lda % 4 == 0) or bf16 (lda % 8 == 0), B a bf16 shadow whose rows hold
 * ceil64(K) elements (zero beyond K; ldb % 8 == 0); C fp32.  Forward pass: B = the weight's shadow; input gradient: A = dY, B = the
 * transposed shadow (K and N exchange roles).  All base pointers 16-byte aligned. */
int tmjx_bgemm_nt(const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K,
                  void *stream);
/* The same GEMM with a block epilogue on the accumulators (the hidden activations then exist in memory only as bf16, the operand format of
 * the next GEMM, and so does z = the pre-activation WITHOUT the bias, saved for the backward pass — BASELINE config 5 is a bf16 MLP; Y and the row
 * statistics are computed from the fp32 accumulators, the backward kernels evaluate silu' / the normalised activation from the bf16 z: 2 instead
 * of 4 bytes written per element here and read, twice, there).  Row tiles are 80 rows high.  ldz in bf16 elements, a multiple of 4.
 *   tmjx_bgemm_ln_fwd   Dense -> SiLU -> LayerNorm forward (intention_network.py:32-44,68-76), N = 128, 256 or 512 (tmjx_bgemm_row_tile_ok):
 *                       Z16[M][ldz] = A B^T as bf16;  Y16 = LayerNorm(silu(A B^T + bias)) * gamma + beta as bf16;  stats[M][2] = (mean, 1 / std).
 *   tmjx_bgemm_ln_bwd   A = dY[M][K] = gradient of the CONSUMER layer's output, Bt = the consumer's transposed shadow (N = the block's width):
 *                       the tile A Bt^T is d loss / d y of the block; stores dZ16[M][lddz] = d loss / d z of the block (its LayerNorm + SiLU
 *                       backward from z, bias, gamma, stats) as bf16 and partial[(M + 79) / 80][3][N] = the row tiles' column sums
 *                       (d gamma | d beta | d bias): reduce them with tmjx_colsum_grouped (rows = (M + 79) / 80, width = 3 N).
 *   tmjx_bgemm_silu_fwd Dense -> SiLU forward (brax value MLP, ppo_networks.py:180-184), any N: Z16 = A B^T as bf16, Y16 (or Yf, fp32) = silu(A B^T + bias).
 *   tmjx_bgemm_silu_bwd its backward in the consumer's input-gradient GEMM: dZ16 = (A Bt^T) silu'(z + bias), partial[(M + 79) / 80][N] = column sums
 *                       of dZ (d bias).
 * tmjx_bgemm_partial_floats(M, N, sums): floats of `partial` for sums = 3 (ln) or 1 (silu). */
int tmjx_bgemm_row_tile_ok(int N);
/* Element size of the SAVED PRE-ACTIVATIONS these entries exchange (`Z16` / `z16` below): 2 = bf16 (the product build: BASELINE configs[4] is a
 * "bf16 MLP on MFMA"), 4 = float (a library built with -DTMJX_BF16_Z_F32: SURVEY a16's narrower "bf16 only for GEMM inputs"; the pointers then
 * point to floats, leading dimensions stay in elements). */
int tmjx_bf16_z_bytes(void);
long long tmjx_bgemm_partial_floats(int M, int N, int sums);
int tmjx_bgemm_ln_fwd(const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, const float *bias, const float *gamma, const float *beta, uint16_t *Z16, int ldz,
                      uint16_t *Y16, int ldy16, float *stats, int M, int N, int K, float eps, void *stream);
int tmjx_bgemm_ln_bwd(const void *dY, int dy_is_f32, int ldy, const uint16_t *Bt, int ldb, const uint16_t *z16, int ldz, const float *bias, const float *gamma,
                      const float *stats, uint16_t *dZ16, int lddz, float *partial, int M, int N, int K, void *stream);
int tmjx_bgemm_silu_fwd(const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, const float *bias, uint16_t *Z16, int ldz, uint16_t *Y16, int ldy16,
                        float *Yf, int ldyf, int M, int N, int K, void *stream);
int tmjx_bgemm_silu_bwd(const void *dY, int dy_is_f32, int ldy, const uint16_t *Bt, int ldb, const uint16_t *z16, int ldz, const float *bias, uint16_t *dZ16, int lddz,
                        float *partial, int M, int N, int K, void *stream);
/* The Dense -> SiLU backward alone (the block's consumer is not a bf16 GEMM): dZ16 = dY silu'(z16 + bias) as bf16 (z16: what tmjx_bgemm_silu_fwd saved), partial[(M + 79) / 80][N] = column
 * sums of dZ per 80-row tile. */
int tmjx_bf_silu_bwd(const float *dY, int ldy, const uint16_t *z16, int ldz, const float *bias, uint16_t *dZ16, int lddz, float *partial, int M, int N, void *stream);
/* The same when the block's consumer is a 1-WIDE un-activated layer (the value head, brax make_value_network: MLP(hidden..., 1)): its input gradient
 * is the outer product dy1[M] (d loss / d head output) x w1[N] (the head's weight row), formed inside the kernel — no [M][N] gradient array, no GEMM with
 * a contraction length of one.  N a multiple of 4 up to 1024, 16-byte aligned rows. */
int tmjx_bf_silu_bwd_rank1(const float *dy1, const float *w1, const uint16_t *z16, int ldz, const float *bias, uint16_t *dZ16, int lddz, float *partial, int M, int N,
                           void *stream);
/* dW[N][lddw] = dY[M][N]^T . X[M][K] and db[N] = column sums of dY (NULL: no bias gradient; sums are taken over the values AS STORED, in
 * fp32) with bf16 operands (each of dY / X fp32 or bf16 in memory, rows 16-byte aligned); scratch >= tmjx_bgemm_dw_scratch_floats(M, N, K). */
long long tmjx_bgemm_dw_scratch_floats(int M, int N, int K);
int tmjx_bgemm_dw(const void *dY, int y_is_f32, int ldy, const void *X, int x_is_f32, int ldx, float *dW, int lddw, float *db, float *scratch,
                  int M, int N, int K, void *stream);
/* All weight (+ bias) gradients of one backward pass in bf16 GEMM-input mode as ONE launch + one reduction launch: up to 24 problems of tmjx_bgemm_dw, each
 * with its own scratch (>= tmjx_bgemm_dw_scratch_floats(M, N, K) floats); the slab count of every problem is capped so that the whole launch is about
 * `target_wgs` workgroups (0: the default, 2048), in multiples of eight slabs (one per XCD).  The reference computes every layer's weight gradient inside one jax.value_and_grad
 * (track_mjx/agent/mlp_ppo/ppo.py:286-300); `problems` is a HOST array (copied into the launch). */
typedef struct tmjx_bdw_problem_t {
  const void *dY, *X;
  float *dW, *db, *scratch;       /* db may be NULL */
  int32_t y_is_f32, x_is_f32, ldy, ldx, lddw, M, N, K;
} tmjx_bdw_problem_t;
int tmjx_bgemm_dw_grouped(const tmjx_bdw_problem_t *problems, int n, int target_wgs, void *stream);

/* The stores of one env-group step into the roll-out buffers in one launch (the Transition of brax acting.actor_step, as the learner of
 * track_mjx/agent/mlp_ppo/ppo.py:330-348 collects it): obs [W][n] (env-minor) -> up to three row-major [n][W] destinations; raw [n][A],
 * logp [n], reward [n], trunc [n] copied; discount_dst = 1 - done.  Any destination may be NULL (skipped).  No LDS. */
typedef struct tmjx_rollout_store_t {
  const float *obs; float *obs_dst0, *obs_dst1;
  const float *raw; float *raw_dst;
  const float *logp; float *logp_dst;
  const float *reward; float *reward_dst;
  const float *done; float *discount_dst;
  const float *trunc; float *trunc_dst;
  int32_t n, W, A;
  float *obs_dst2;      /* a third row-major destination (the acting policy's staging copy of the new observation), may be NULL */
} tmjx_rollout_store_t;
int tmjx_rollout_store(const tmjx_rollout_store_t *s, void *stream);
/* `m` reads `owner`'s resident clip table instead of holding a copy of its own (the env groups of one rank: one upload per rank).  `owner`
 * must outlive every launch of `m`. */
int tmjx_clips_share(tmjx_model *m, const tmjx_model *owner);

/* Observation normaliser update (brax running_statistics.update as called at track_mjx/agent/mlp_ppo/ppo.py:357-361; math:
 * track_mjx/agent/masked_running_statistics.py:161-214) in one pass over src [rows][W] (W % 4 == 0):
 *   tmjx_stats_sums:  sums[0..W) = sum_rows(x - mean), sums[W..2W) = sum_rows((x - mean)^2); scratch >= tmjx_stats_scratch_floats(W).
 *   (multi-GPU: all-reduce `sums` here — replaces the reference's psums of mean_update and variance_update, same totals)
 *   tmjx_stats_apply: count += n_added; mean += S1 / count; summed_variance += S2 - (S1 / count) S1; std = clip(sqrt(max(sv, 0) / count)).
 * n_added = rows summed over all ranks. */
int tmjx_stats_scratch_floats(int W);
int tmjx_stats_sums(const float *src, const float *mean, float *sums, float *scratch, long long rows, int W, void *stream);
int tmjx_stats_apply(const float *sums, float n_added, float *count, float *mean, float *summed_variance, float *std, int W, float std_min,
                     float std_max, void *stream);

/* Debug/test access: copy a named per-env workspace/intermediate array of the last tmjx_forward /
 * tmjx_physics call into `out` (device pointer, [count][n_env]); returns count or a negative code.
 * Names: "qM" (sparse rows), "qfrc_smooth", "qacc", "qacc_smooth", "efc_D", "efc_aref", "con_dist", ...;
 * "solver_stats" = [CG iterations (mjx data.solver_niter), line-search iterations summed, constraint rows that entered the solver,
 * of which joint limits] of the last substep; "efc_in" [nefc] = 1 for the original rows that entered the solver. */
int tmjx_debug_rows(const tmjx_model *m, const char *name, int *row0, int *count);

const char *tmjx_last_error(void);
const char *tmjx_version(void);

#ifdef __cplusplus
}
#endif
#endif
