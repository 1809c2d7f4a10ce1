// csrc/tmjx_hip.hip — kernels and C-ABI of libtmjx_hip.so (gfx950 / MI355X only).
//
// The per-env bodies live in wave_physics.h (K2: one wavefront per env), env_core.h (K1 / K3: one lane per env or per (env, part), env
// index = coalesced axis of every buffer, include/tmjx.h layout rules) and ppo_kernels.h (learner); this file launches them.
// -DTMJX_LANE_IMPL additionally compiles the lane-per-env physics (tests/lane/physics_core.h; TMJX_IMPL=lane selects it): a second, independent
// HIP implementation used by the tests as a cross-check — the product library is built without it.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/tmjx.h"
#include "env_core.h"
#include "model_host.h"
#include "wave_physics.h"
#ifdef TMJX_LANE_IMPL
#include "../../tests/lane/physics_core.h"
#endif

struct tmjx_model {
  DModel h;           // host copy (clip pointers are device pointers)
  DModel *d = nullptr;  // device copy
  float *clips[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int block = 64;     // lanes per workgroup for the lane-per-env kernels (K1/K3 and the v1 physics)
  int wave = 1;       // 1: wave-per-env LDS physics kernel (default); 0: lane-per-env reference implementation
  bool rodent = false;  // dims match the compile-time specialisation of the wave kernel
  // chain layout of the wave kernel (wave_layout.h): per-env global copy of the inertia matrix M (written after "M rows", read by
  // Euler's factorisation of M + h D).  Owned by the handle, (re)allocated when a launch needs more envs than it holds — the one
  // piece of mutable per-handle scratch: a handle serves one stream at a time (include/tmjx.h)
  mutable float *mspill = nullptr;
  mutable int mspill_envs = 0;
  bool clips_owned = true;   // false: the clip table belongs to another handle of the same device (tmjx_clips_share)
  int action_repeat = 1;     // brax EpisodeWrapper's repeat count (tmjx_set_action_repeat)
};
#define WAVE_SPILL_STRIDE(m) ((((m)->h.nnz) + 63) & ~63)
// record stride: state rows qpos .. qfrc_actuator, then the action, rounded up to 16 words
#define WAVE_REC_STRIDE(m) ((((m)->h.s_prev_ctrl + (m)->h.nu) + 15) & ~15)
// workspace words in front of the record: window partials (2 nu rows) + post partials (16 rows), each n_env wide
#define WAVE_REC_OFFSET(m, n) ((size_t)(2 * (m)->h.nu + 16) * (size_t)(n))
static int launch_wave(const tmjx_model *m, float *state, const float *action, int nsub, int do_euler, float *ws, int n_env, hipStream_t stream,
                       float *rec = nullptr);

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
// the library's other translation units (tmjx_bf16.hip) record their failures in the same per-thread message
#define TMJX_SMALL_BATCH 8192      // rows up to which the acting-path element-wise kernels launch one-wave blocks
extern "C" int tmjx_internal_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(TMJX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
  } while (0)

// ----------------------------------------------------------------------------------------------- kernels
#ifdef TMJX_LANE_IMPL
__global__ void k_reset(const DModel *__restrict__ mp, float *st, int *is, const int *clip, const int *start,
                        const float *qn, const float *vn, float *obs, float *ws, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, ws, n, e};
  tm_reset_pre(m, r, is, clip[e], start[e], qn, vn);
  tm_forward(m, r);
  tm_reset_post(m, r, is, obs);
}

__global__ void k_step(const DModel *__restrict__ mp, float *st, int *is, const float *action, float *obs, float *reward,
                       float *done, float *trunc, float *metrics, float *ws, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, ws, n, e};
  tm_step_prologue(m, r);
  for (int a = 0; a < m.nu; a++) WS(m.w_ctrl, a) = action[(size_t)a * n + e];
  for (int f = 0; f < m.n_frames; f++) { tm_forward(m, r); tm_euler(m, r); }
  tm_step_post(m, r, is, action, obs, reward, done, trunc, metrics);
}

__global__ void k_physics(const DModel *__restrict__ mp, float *st, const float *action, int nsub, int do_euler, float *ws, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, ws, n, e};
  for (int a = 0; a < m.nu; a++) WS(m.w_ctrl, a) = action ? action[(size_t)a * n + e] : 0.f;
  for (int f = 0; f < nsub; f++) { tm_forward(m, r); if (do_euler) tm_euler(m, r); }
}

#endif  // TMJX_LANE_IMPL

// window statistics, one lane per (action dim, env): 38x more parallelism than lane-per-env for the 50x38 ring buffer
__global__ void k_window(const DModel *__restrict__ mp, float *st, const int *is, const float *action, float *win, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, nullptr, n, e};
  float vi, ji;
  tm_window_dim(m, r, i, is[(size_t)m.i_buffer_index * n + e], action[(size_t)i * n + e], vi, ji);
  win[(size_t)i * n + e] = vi;
  win[(size_t)(m.nu + i) * n + e] = ji;
}
__global__ void k_post(const DModel *__restrict__ mp, float *st, int *is, const float *action, float *obs, float *reward,
                       float *done, float *trunc, float *metrics, const float *win, int split, int rep, int n) {
  TM_PRIO_ACTING();
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, nullptr, n, e};
  if (rep & TM_REP_FIRST) tm_step_prologue(m, r);
  // split: the long sums were computed by k_post_parts into the workspace rows behind the 2 nu window partials
  tm_step_post(m, r, is, action, obs, reward, done, trunc, metrics, win, split != 0, split ? win + (size_t)2 * m.nu * n : nullptr, rep);
}
// k_post's inline form with the CALLER's reference frame per env (tmjx_reward_frame; env_core.h: TmFrame)
__global__ void k_post_frame(const DModel *__restrict__ mp, float *st, int *is, const float *action, float *obs, float *reward,
                             float *done, float *trunc, float *metrics, TmFrame fo, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, nullptr, n, e};
  tm_step_prologue(m, r);
  tm_step_post(m, r, is, action, obs, reward, done, trunc, metrics, nullptr, false, nullptr, TM_REP_ONE, &fo);
}
// the long reductions of the reward / termination step, one lane per (env, part) (env_core.h: tm_post_part)
__global__ void k_post_parts(const DModel *__restrict__ mp, float *st, const int *is, float *P, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  EnvRef r{st, nullptr, n, e};
  tm_post_part(*mp, r, is, blockIdx.y, P);
}
// observation, one lane per (env, part): TM_OBS_PARTS(T) = 18 pieces (env_core.h: tm_get_obs); rows of obs stay coalesced over envs
__global__ void k_obs(const DModel *__restrict__ mp, float *st, const int *is, float *__restrict__ obs, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x, part = blockIdx.y;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, nullptr, n, e};
  int clip = is[(size_t)m.i_clip_idx * n + e], start = is[(size_t)m.i_start_frame * n + e];
  tm_get_obs(m, r, clip, tm_cur_frame(m, ST(m.s_time, 0), start), obs, true, part);
}
// k_window, k_obs and k_post_parts in ONE launch (blockIdx.y picks the piece): the three read the post-physics state and write disjoint
// outputs, so a group's serial phase between two physics launches is one launch latency shorter twice over.  (Measured in round 1, when
// such launches mostly waited for wave slots: no gain; re-measured in round 3 with three env groups and one-wave blocks everywhere.)
__global__ __launch_bounds__(64) void k_step_parts(const DModel *__restrict__ mp, float *st, const int *is, const float *action, float *win,
                                                   float *__restrict__ obs, float *P, int n, int n_obs_parts) {
  TM_PRIO_ACTING();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const DModel &m = *mp;
  EnvRef r{st, nullptr, n, e};
  int y = blockIdx.y;
  if (y < m.nu) {
    float vi, ji;
    tm_window_dim(m, r, y, is[(size_t)m.i_buffer_index * n + e], action[(size_t)y * n + e], vi, ji);
    win[(size_t)y * n + e] = vi;
    win[(size_t)(m.nu + y) * n + e] = ji;
    return;
  }
  y -= m.nu;
  if (y < n_obs_parts) {
    int clip = is[(size_t)m.i_clip_idx * n + e], start = is[(size_t)m.i_start_frame * n + e];
    tm_get_obs(m, r, clip, tm_cur_frame(m, ST(m.s_time, 0), start), obs, true, y);
    return;
  }
  tm_post_part(m, r, is, y - n_obs_parts, P);
}
// auto-reset of the envs that are done: physics state, observation and prev_ctrl <- the snapshot taken at reset
// (wrappers.py:104-144), one lane per (env, block of 16 rows)
__global__ void k_autoreset(const DModel *__restrict__ mp, float *st, float *obs, const float *done, int n) {
  TM_PRIO_ACTING();
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || done[e] == 0.f) return;
  const DModel &m = *mp;
  EnvRef r{st, nullptr, n, e};
  int total = m.nphys + m.obs_size + m.nu;
  for (int k = blockIdx.y * 16; k < total && k < (int)blockIdx.y * 16 + 16; k++) {
    if (k < m.nphys) ST(m.s_qpos, k) = ST(m.s_first_phys, k);
    else if (k < m.nphys + m.obs_size) OUTROW(obs, k - m.nphys) = ST(m.s_first_obs, k - m.nphys);
    else ST(m.s_prev_ctrl, k - m.nphys - m.obs_size) = ST(m.s_first_prev_ctrl, k - m.nphys - m.obs_size);
  }
}

// Env-major physics record of K2: rec[e][0 .. s_prev_ctrl) = the state rows qpos .. qfrc_actuator of env e (physics state + the
// outputs K3 reads), rec[e][s_prev_ctrl .. + nu) = the action.  The wave-per-env kernel then reads / writes contiguous words
// (without this each of its 4-byte accesses to the [row][n_env] buffers occupied its own 32-byte sector: 131 MB of HBM traffic
// per launch for 14 MB of data, profiles/pmc_traffic.json).  The two transposes use NO LDS: with pipelined env groups they are
// launched while the other group's physics kernel owns every CU's LDS, and an LDS-tiled version sat in the queue for up to
// 1.2 ms waiting for it.  One lane = one env and 16 consecutive rows: the [row][n_env] side is coalesced across lanes, the
// record side is four float4 accesses per lane whose 64-byte lines are completed by the lane itself (L1 / L2 absorb them).
#define REC_ROWS_PER_THREAD 16
__global__ __launch_bounds__(64) void k_rec_in(const DModel *__restrict__ mp, const float *__restrict__ st, const float *__restrict__ action,
                                               float *__restrict__ rec, int n, int rs) {
  TM_PRIO_ACTING();
  const DModel &m = *mp;
  const int e = blockIdx.x * 64 + threadIdx.x, r0 = blockIdx.y * REC_ROWS_PER_THREAD;     // rows: nphys state rows, then nu action rows
  const int nrow = m.nphys + m.nu;
  if (e >= n) return;
  float v[REC_ROWS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < REC_ROWS_PER_THREAD; j++) {
    int r = r0 + j;
    v[j] = r < m.nphys ? st[(size_t)(m.s_qpos + r) * n + e] : (r < nrow ? action[(size_t)(r - m.nphys) * n + e] : 0.f);
  }
#pragma unroll
  for (int j = 0; j < REC_ROWS_PER_THREAD; j++) {
    int r = r0 + j;
    if (r < nrow) rec[(size_t)e * rs + (r < m.nphys ? m.s_qpos + r : m.s_prev_ctrl + (r - m.nphys))] = v[j];
  }
}
__global__ __launch_bounds__(64) void k_rec_out(const DModel *__restrict__ mp, float *__restrict__ st, const float *__restrict__ rec, int n, int rs) {
  TM_PRIO_ACTING();
  const DModel &m = *mp;
  const int e = blockIdx.x * 64 + threadIdx.x, r0 = blockIdx.y * REC_ROWS_PER_THREAD;
  const int nrow = m.s_prev_ctrl - m.s_qpos;      // physics state + xpos / torso xmat / qfrc_actuator
  if (e >= n) return;
  float v[REC_ROWS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < REC_ROWS_PER_THREAD; j++) { int r = r0 + j; v[j] = r < nrow ? rec[(size_t)e * rs + m.s_qpos + r] : 0.f; }
#pragma unroll
  for (int j = 0; j < REC_ROWS_PER_THREAD; j++) { int r = r0 + j; if (r < nrow) st[(size_t)(m.s_qpos + r) * n + e] = v[j]; }
}

// K2, wave-per-env (k_physics_wave, csrc/wave_physics.h) is compiled in a translation unit of its own, csrc/tmjx_wave.hip — with a compiler
// flag of its own (track_mjx_amd/hip.py) — and launched through this entry
extern "C" void tmjx_internal_launch_physics_wave(int rodent, int cnt, size_t lds, hipStream_t stream, const DModel *mp, float *st, const float *action, int nsub,
                                                  int do_euler, float *ws_dump, int n, int e0, int rs, float *spill, int spill_stride);

__global__ void k_reset_pre(const DModel *__restrict__ mp, float *st, int *is, const int *clip, const int *start, const float *qn,
                            const float *vn, float *ws, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  EnvRef r{st, ws, n, e};
  tm_reset_pre(*mp, r, is, clip[e], start[e], qn, vn);
}
__global__ void k_reset_post(const DModel *__restrict__ mp, float *st, int *is, float *obs, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  EnvRef r{st, nullptr, n, e};
  tm_reset_post(*mp, r, is, obs);
}

// compute_gae (losses.py:39-100): one lane per batch column, reverse scan over T then the advantage pass
__global__ void k_gae(const float *__restrict__ trunc, const float *__restrict__ term, const float *__restrict__ rew,
                      const float *__restrict__ val, const float *__restrict__ boot, float lam, float disc, float *vs,
                      float *adv, int T, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f, bv = boot[b], vnext = bv;
  for (int t = T - 1; t >= 0; t--) {
    size_t i = (size_t)t * B + b;
    float tm = 1.f - trunc[i], te = term[i], v = val[i];
    float delta = (rew[i] + disc * (1.f - te) * vnext - v) * tm;
    acc = delta + disc * (1.f - te) * tm * lam * acc;
    vs[i] = acc + v;
    vnext = v;
  }
  float vsn = bv;
  for (int t = T - 1; t >= 0; t--) {
    size_t i = (size_t)t * B + b;
    float tm = 1.f - trunc[i], te = term[i];
    float cur = vs[i];
    adv[i] = (rew[i] + disc * (1.f - te) * vsn - val[i]) * tm;
    vsn = cur;
  }
}

#include "ppo_kernels.h"
#include "gemm_kernels.h"

// ----------------------------------------------------------------------------------------------- C-ABI
extern "C" {

const char *tmjx_last_error(void) { return g_err.c_str(); }
const char *tmjx_version(void) { return "tmjx-hip 0.4 (gfx950, wave-per-env LDS physics)"; }

int tmjx_model_create(const void *blob, size_t nbytes, tmjx_model **out) {
  if (!blob || !out) return fail(TMJX_EINVAL, "null argument");
  tmjx_model *m = new tmjx_model();
  std::string err;
  if (!tmjx_host::build_dmodel(blob, nbytes, m->h, err)) { delete m; return fail(TMJX_EINVAL, err); }
  const char *bs = getenv("TMJX_BLOCK");
  if (bs) { int v = atoi(bs); if (v >= 1 && v <= 256) m->block = v; }
#ifdef TMJX_LANE_IMPL
  const char *impl = getenv("TMJX_IMPL");
  if (impl && !strcmp(impl, "lane")) m->wave = 0;
#endif
  if (m->wave) {
    size_t lds_bytes = (size_t)tmjx_host::make_wave_layout(m->h, false).lds_floats * sizeof(float);   // the larger (generic) layout
    if (lds_bytes > 160 * 1024) { delete m; return fail(TMJX_EINVAL, "model does not fit the 160 KiB LDS of a CU"); }
    if (lds_bytes > 64 * 1024) { delete m; return fail(TMJX_EINVAL, "model needs more than 64 KiB of LDS per env"); }
    if (m->h.nefc > 254 || m->h.nlim > 128) { delete m; return fail(TMJX_EINVAL, "more than 254 constraint rows or 128 joint limits (wave kernel: byte row map, line-search rows in 4 registers per lane)"); }
    constexpr WLayout ks(TMW_RODENT_DIMS, 1);
    const WLayout kd = tmjx_host::make_wave_layout(m->h);
    m->rodent = !getenv("TMJX_WAVE_DYNAMIC") && kd.nbody == ks.nbody && kd.njnt == ks.njnt && kd.nq == ks.nq && kd.nv == ks.nv &&
                kd.nu == ks.nu && kd.ncon == ks.ncon && kd.nlim == ks.nlim && kd.nnz == ks.nnz && kd.ngroup == ks.ngroup &&
                kd.nround_body == ks.nround_body && kd.nround_dof == ks.nround_dof && kd.lds_floats == ks.lds_floats && kd.chains == 1;
  }
  hipError_t e = hipMalloc((void **)&m->d, sizeof(DModel));
  if (e != hipSuccess) { delete m; return fail(TMJX_ENOMEM, std::string("hipMalloc(DModel): ") + hipGetErrorString(e)); }
  e = hipMemcpy(m->d, &m->h, sizeof(DModel), hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(m->d); delete m; return fail(TMJX_EHIP, std::string("hipMemcpy(DModel): ") + hipGetErrorString(e)); }
  *out = m;
  return TMJX_OK;
}

void tmjx_model_destroy(tmjx_model *m) {
  if (!m) return;
  for (int i = 0; i < 5; i++) if (m->clips[i] && m->clips_owned) hipFree(m->clips[i]);
  if (m->d) hipFree(m->d);
  if (m->mspill) hipFree(m->mspill);
  delete m;
}

int tmjx_layout(const tmjx_model *mm, tmjx_layout_t *o) {
  if (!mm || !o) return fail(TMJX_EINVAL, "null argument");
  const DModel &m = mm->h;
  o->nq = m.nq; o->nv = m.nv; o->nu = m.nu; o->nbody = m.nbody; o->ncon = m.ncon; o->nefc = m.nefc;
  o->obs_size = m.obs_size; o->ref_obs_size = m.ref_obs_size; o->n_metrics = TM_NMETRIC; o->window = m.window;
  o->qpos = m.s_qpos; o->qvel = m.s_qvel; o->act = m.s_act; o->qacc_warmstart = m.s_warm; o->time = m.s_time;
  o->xpos = m.s_xpos; o->xmat_torso = m.s_xmat_torso; o->qfrc_actuator = m.s_qfrc_actuator;
  o->prev_ctrl = m.s_prev_ctrl; o->action_buffer = m.s_action_buffer; o->done = m.s_done; o->steps_f = m.s_steps;
  o->first_phys = m.s_first_phys; o->first_obs = m.s_first_obs; o->first_prev_ctrl = m.s_first_prev_ctrl;
  o->state_rows = m.s_rows;
  o->i_clip_idx = m.i_clip_idx; o->i_start_frame = m.i_start_frame; o->i_buffer_index = m.i_buffer_index;
  o->i_nan_count = m.i_nan_count; o->istate_rows = m.i_rows; o->ws_rows = m.w_rows;
  return TMJX_OK;
}

int tmjx_set_wrappers(tmjx_model *m, int episode_length, int auto_reset) {
  if (!m) return fail(TMJX_EINVAL, "null argument");
  if (episode_length < 1) return fail(TMJX_EINVAL, "episode_length must be >= 1");
  m->h.episode_length = episode_length;
  m->h.auto_reset = auto_reset ? 1 : 0;
  // a blocking copy of the constants: called between roll-outs (wrappers.wrap, before the first reset), never while launches of this
  // handle are in flight; the clip table stays where it is (re-creating the handle re-uploaded it: 631 MB at 1024 clips)
  HIP_TRY(hipMemcpy(m->d, &m->h, sizeof(DModel), hipMemcpyHostToDevice));
  return TMJX_OK;
}

int tmjx_set_action_repeat(tmjx_model *m, int action_repeat) {
  if (!m) return fail(TMJX_EINVAL, "null argument");
  if (action_repeat < 1 || action_repeat > TM_REP_MAX) return fail(TMJX_EINVAL, "action_repeat must be in 1 .. " + std::to_string(TM_REP_MAX));
  if (action_repeat > 1 && !m->wave) return fail(TMJX_EINVAL, "action_repeat > 1 needs the wave-per-env implementation");
  m->action_repeat = action_repeat;
  return TMJX_OK;
}

int tmjx_clips_upload(tmjx_model *m, const float *position, const float *quaternion, const float *joints,
                      const float *body_positions, const float *angular_velocity, int n_clips, int n_frames) {
  if (!m || !position || !quaternion || !joints || !body_positions || !angular_velocity) return fail(TMJX_EINVAL, "null argument");
  if (n_clips < 1 || n_frames < m->h.traj_length) return fail(TMJX_EINVAL, "clip table needs >= 1 clip and >= traj_length frames");
  size_t cf = (size_t)n_clips * n_frames;
  size_t widths[5] = {3, 4, (size_t)(m->h.nq - 7), (size_t)(m->h.nbody - 1) * 3, 3};
  const float *src[5] = {position, quaternion, joints, body_positions, angular_velocity};
  for (int i = 0; i < 5; i++) {
    if (m->clips[i] && m->clips_owned) hipFree(m->clips[i]);
    m->clips[i] = nullptr;
  }
  m->clips_owned = true;
  for (int i = 0; i < 5; i++) {
    HIP_TRY(hipMalloc((void **)&m->clips[i], cf * widths[i] * sizeof(float)));
    HIP_TRY(hipMemcpy(m->clips[i], src[i], cf * widths[i] * sizeof(float), hipMemcpyHostToDevice));
  }
  m->h.clip_pos = m->clips[0]; m->h.clip_quat = m->clips[1]; m->h.clip_joints = m->clips[2];
  m->h.clip_bodypos = m->clips[3]; m->h.clip_angvel = m->clips[4];
  m->h.n_clips = n_clips; m->h.n_frames_clip = n_frames;
  HIP_TRY(hipMemcpy(m->d, &m->h, sizeof(DModel), hipMemcpyHostToDevice));
  return TMJX_OK;
}

// The env groups of one rank (pipelined roll-outs: one handle each) read ONE resident clip table: `m` takes over `owner`'s device
// arrays (631 MB at 1024 clips) without owning them.  `owner` must outlive `m`'s launches and be of the same model dimensions.
int tmjx_clips_share(tmjx_model *m, const tmjx_model *owner) {
  if (!m || !owner) return fail(TMJX_EINVAL, "null argument");
  if (m == owner) return TMJX_OK;
  if (!owner->clips[0]) return fail(TMJX_EINVAL, "the owner has no clip table");
  if (m->h.nq != owner->h.nq || m->h.nbody != owner->h.nbody || m->h.traj_length > owner->h.n_frames_clip) return fail(TMJX_EINVAL, "handles of different models cannot share a clip table");
  for (int i = 0; i < 5; i++) {
    if (m->clips[i] && m->clips_owned) hipFree(m->clips[i]);
    m->clips[i] = owner->clips[i];
  }
  m->clips_owned = false;
  m->h.clip_pos = m->clips[0]; m->h.clip_quat = m->clips[1]; m->h.clip_joints = m->clips[2];
  m->h.clip_bodypos = m->clips[3]; m->h.clip_angvel = m->clips[4];
  m->h.n_clips = owner->h.n_clips; m->h.n_frames_clip = owner->h.n_frames_clip;
  HIP_TRY(hipMemcpy(m->d, &m->h, sizeof(DModel), hipMemcpyHostToDevice));
  return TMJX_OK;
}

static int launch_wave(const tmjx_model *m, float *state, const float *action, int nsub, int do_euler, float *ws, int n_env, hipStream_t stream,
                       float *rec) {
  // the compile-time (rodent) kernel uses the chain layout, the run-time one the generic layout of the same dims
  size_t lds = (size_t)(m->rodent ? m->h.lds_floats : tmjx_host::make_wave_layout(m->h, false).lds_floats) * sizeof(float);
  if (const char *pad = getenv("TMJX_LDS_PAD_KB")) lds += (size_t)atoi(pad) * 1024;  // occupancy experiments only
  const int sstride = WAVE_SPILL_STRIDE(m);
  if (m->rodent && m->mspill_envs < n_env) {
    if (m->mspill) hipFree(m->mspill);            // (hipFree waits for the launches that may still read the old block)
    m->mspill = nullptr; m->mspill_envs = 0;
    hipError_t me = hipMalloc((void **)&m->mspill, ((size_t)n_env * sstride + 64) * sizeof(float));
    if (me != hipSuccess) {                         // nothing was launched: the caller returns this code (no silent skip of the physics)
      m->mspill = nullptr;
      (void)hipGetLastError();
      return fail(TMJX_ENOMEM, std::string("hipMalloc of the physics kernel's per-env inertia-matrix scratch (") +
                                   std::to_string(((size_t)n_env * sstride + 64) * sizeof(float)) + " bytes): " + hipGetErrorString(me));
    }
    m->mspill_envs = n_env;
  }
  float *spill = m->rodent ? m->mspill : nullptr;
  int parts = 1;
  if (const char *sp = getenv("TMJX_SPLIT_LAUNCH")) { parts = atoi(sp); if (parts < 1 || n_env % parts) parts = 1; }   // scheduling experiments
  const int rs = rec ? WAVE_REC_STRIDE(m) : 0;
  if (rec) hipLaunchKernelGGL(k_rec_in, dim3((n_env + 63) / 64, (m->h.nphys + m->h.nu + REC_ROWS_PER_THREAD - 1) / REC_ROWS_PER_THREAD), dim3(64), 0, stream, m->d, (const float *)state, action, rec, n_env, rs);
  float *st = rec ? rec : state;
  for (int p = 0; p < parts; p++) {
    int cnt = n_env / parts, e0 = p * cnt;
    tmjx_internal_launch_physics_wave(m->rodent ? 1 : 0, cnt, lds, stream, m->d, st, action, nsub, do_euler, ws, n_env, e0, rs, spill, sstride);
  }
  if (rec) hipLaunchKernelGGL(k_rec_out, dim3((n_env + 63) / 64, (m->h.s_prev_ctrl - m->h.s_qpos + REC_ROWS_PER_THREAD - 1) / REC_ROWS_PER_THREAD), dim3(64), 0, stream, m->d, state, (const float *)rec, n_env, rs);
  return TMJX_OK;
}
// env-major physics record inside the caller's workspace (behind the K3 partial rows), or nullptr = direct [row][n_env] access
// (TMJX_NO_RECORD=1, or a workspace too small for it)
static float *wave_record(const tmjx_model *m, float *workspace, int n_env) {
  if (!workspace || getenv("TMJX_NO_RECORD")) return nullptr;
  if ((size_t)m->h.w_rows * (size_t)n_env < WAVE_REC_OFFSET(m, n_env) + (size_t)WAVE_REC_STRIDE(m) * (size_t)n_env) return nullptr;
  return workspace + WAVE_REC_OFFSET(m, n_env);
}
static int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TMJX_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  return TMJX_OK;
}
#define GRID(m, n) dim3(((n) + (m)->block - 1) / (m)->block), dim3((m)->block)
#define WAVE_LDS(m) ((size_t)(m)->h.lds_floats * sizeof(float))

int tmjx_reset(tmjx_model *m, float *state, int32_t *istate, const int32_t *clip_idx, const int32_t *start_frame,
               const float *qpos_noise, const float *qvel_noise, float *obs, float *workspace, int n_env, void *stream) {
  if (!m || !state || !istate || !clip_idx || !start_frame || !qpos_noise || !qvel_noise || !obs || !workspace) return fail(TMJX_EINVAL, "null argument");
  if (n_env < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
  if (!m->h.clip_pos) return fail(TMJX_EINVAL, "tmjx_clips_upload has not been called");
  if (m->wave) {
    hipLaunchKernelGGL(k_reset_pre, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, istate, clip_idx, start_frame, qpos_noise,
                       qvel_noise, workspace, n_env);
    if (int rc = launch_wave(m, state, (const float *)nullptr, 1, 0, (float *)nullptr, n_env, (hipStream_t)stream)) return rc;
    hipLaunchKernelGGL(k_reset_post, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, istate, obs, n_env);
    return check_launch("k_reset(wave)");
  }
#ifdef TMJX_LANE_IMPL
  hipLaunchKernelGGL(k_reset, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, istate, clip_idx, start_frame, qpos_noise,
                     qvel_noise, obs, workspace, n_env);
  return check_launch("k_reset");
#else
  return fail(TMJX_EINVAL, "lane-per-env implementation not built");
#endif
}

// K3 behind the physics launch: window statistics + observation parts + reward partial sums (ONE launch, k_step_parts; TMJX_K3_SEPARATE=1:
// three), reward / termination, auto-reset copies.  64-lane workgroups throughout: next to the other env groups' physics kernel (up to 3 waves
// of 168 VGPRs per SIMD) a 256-lane workgroup had to wait for four wave slots WITH registers on one CU — 245 us on average instead of 30
static void launch_k3(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward, float *done,
                      float *truncation, float *metrics, float *workspace, int n_env, hipStream_t stream, int rep = TM_REP_ONE) {
  const DModel &h = m->h;
  static const bool merged = !getenv("TMJX_K3_SEPARATE");
  const int nobs = TM_OBS_PARTS(h.traj_length);
  float *parts = workspace + (size_t)2 * h.nu * n_env;
  if (merged) {
    hipLaunchKernelGGL(k_step_parts, dim3((n_env + 63) / 64, h.nu + nobs + TM_NPOST), dim3(64), 0, stream, m->d, state, istate, action, workspace, obs, parts, n_env, nobs);
  } else {
    hipLaunchKernelGGL(k_window, dim3((n_env + 63) / 64, h.nu), dim3(64), 0, stream, m->d, state, istate, action, workspace, n_env);
    hipLaunchKernelGGL(k_obs, dim3((n_env + 63) / 64, nobs), dim3(64), 0, stream, m->d, state, istate, obs, n_env);
    hipLaunchKernelGGL(k_post_parts, dim3((n_env + 63) / 64, TM_NPOST), dim3(64), 0, stream, m->d, state, istate, parts, n_env);
  }
  hipLaunchKernelGGL(k_post, dim3((n_env + 63) / 64), dim3(64), 0, stream, m->d, state, istate, action, obs, reward, done, truncation, metrics,
                     (const float *)workspace, 1, rep, n_env);
  if (h.auto_reset && (rep & TM_REP_LAST)) {
    int total = h.nphys + h.obs_size + h.nu;
    hipLaunchKernelGGL(k_autoreset, dim3((n_env + 63) / 64, (total + 15) / 16), dim3(64), 0, stream, m->d, state, obs, done, n_env);
  }
}

int tmjx_step(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward, float *done,
              float *truncation, float *metrics, float *workspace, int n_env, void *stream) {
  if (!m || !state || !istate || !action || !obs || !reward || !done || !truncation || !metrics || !workspace) return fail(TMJX_EINVAL, "null argument");
  if (n_env < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
  if (!m->h.clip_pos) return fail(TMJX_EINVAL, "tmjx_clips_upload has not been called");
  if (m->wave) {
    // action_repeat (brax EpisodeWrapper): the env's own step R times with the same action; K3 sums the rewards and applies the episode
    // counter / truncation / auto-reset after the last repeat only (env_core.h: tm_step_post, TM_REP_*)
    const int R = m->action_repeat;
    for (int r = 0; r < R; r++) {
      if (int rc = launch_wave(m, state, action, m->h.n_frames, 1, (float *)nullptr, n_env, (hipStream_t)stream, wave_record(m, workspace, n_env))) return rc;
      launch_k3(m, state, istate, action, obs, reward, done, truncation, metrics, workspace, n_env, (hipStream_t)stream,
                TM_REP(R, r == 0, r == R - 1));
    }
    return check_launch("k_step(wave)");
  }
#ifdef TMJX_LANE_IMPL
  hipLaunchKernelGGL(k_step, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, istate, action, obs, reward, done, truncation,
                     metrics, workspace, n_env);
  return check_launch("k_step");
#else
  return fail(TMJX_EINVAL, "lane-per-env implementation not built");
#endif
}

// the physics part of tmjx_step alone (record transposes + K2 with the configured n_frames): what bench.py brackets with HIP events
int tmjx_physics_step(tmjx_model *m, float *state, const float *action, float *workspace, int n_env, void *stream) {
  if (!m || !state || !action || !workspace) return fail(TMJX_EINVAL, "null argument");
  if (n_env < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
#ifdef TMJX_LANE_IMPL
  if (!m->wave) {
    hipLaunchKernelGGL(k_physics, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, action, m->h.n_frames, 1, workspace, n_env);
    return check_launch("k_physics");
  }
#endif
  if (int rc = launch_wave(m, state, action, m->h.n_frames, 1, (float *)nullptr, n_env, (hipStream_t)stream, wave_record(m, workspace, n_env))) return rc;
  return check_launch("k_physics_wave");
}

int tmjx_physics(tmjx_model *m, float *state, const float *action, int n_substeps, float *workspace, int n_env, void *stream) {
  if (!m || !state || (!workspace && !m->wave)) return fail(TMJX_EINVAL, "null argument");
  if (n_env < 1 || n_substeps < 0) return fail(TMJX_EINVAL, "bad n_env / n_substeps");
  if (m->wave) {
    if (int rc = launch_wave(m, state, action, n_substeps, 1, workspace, n_env, (hipStream_t)stream)) return rc;
    return check_launch("k_physics_wave");
  }
#ifdef TMJX_LANE_IMPL
  hipLaunchKernelGGL(k_physics, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, action, n_substeps, 1, workspace, n_env);
  return check_launch("k_physics");
#else
  return fail(TMJX_EINVAL, "lane-per-env implementation not built");
#endif
}

int tmjx_forward(tmjx_model *m, float *state, float *workspace, int n_env, void *stream) {
  if (!m || !state || !workspace) return fail(TMJX_EINVAL, "null argument");
  if (n_env < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
  if (m->wave) {
    if (int rc = launch_wave(m, state, (const float *)nullptr, 1, 0, workspace, n_env, (hipStream_t)stream)) return rc;
    return check_launch("k_forward(wave)");
  }
#ifdef TMJX_LANE_IMPL
  hipLaunchKernelGGL(k_physics, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, (const float *)nullptr, 1, 0, workspace, n_env);
  return check_launch("k_forward");
#else
  return fail(TMJX_EINVAL, "lane-per-env implementation not built");
#endif
}

int tmjx_reward_obs(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward, float *done,
                    float *truncation, float *metrics, float *workspace, int n_env, void *stream) {
  if (!m || !state || !istate || !action || !obs || !reward || !done || !truncation || !metrics) return fail(TMJX_EINVAL, "null argument");
  if (n_env < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
  if (!m->h.clip_pos) return fail(TMJX_EINVAL, "tmjx_clips_upload has not been called");
  if (workspace) launch_k3(m, state, istate, action, obs, reward, done, truncation, metrics, workspace, n_env, (hipStream_t)stream);
  else hipLaunchKernelGGL(k_post, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, istate, action, obs, reward, done, truncation,
                          metrics, (const float *)nullptr, 0, TM_REP_ONE, n_env);
  return check_launch("k_post");
}

int tmjx_reward_frame(tmjx_model *m, float *state, int32_t *istate, const float *action, const float *frame_pos, const float *frame_quat,
                      const float *frame_joints, const float *frame_bodypos, const float *frame_angvel, float *obs, float *reward, float *done,
                      float *truncation, float *metrics, int n_env, void *stream) {
  if (!m || !state || !istate || !action || !obs || !reward || !done || !truncation || !metrics) return fail(TMJX_EINVAL, "null argument");
  if (!frame_pos || !frame_quat || !frame_joints || !frame_bodypos || !frame_angvel) return fail(TMJX_EINVAL, "null reference-frame leaf");
  if (n_env < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
  if (!m->h.clip_pos) return fail(TMJX_EINVAL, "tmjx_clips_upload has not been called");      // (the observation's trajectory part still reads the table)
  TmFrame fo{frame_pos, frame_quat, frame_joints, frame_bodypos, frame_angvel};
  hipLaunchKernelGGL(k_post_frame, GRID(m, n_env), 0, (hipStream_t)stream, m->d, state, istate, action, obs, reward, done, truncation, metrics, fo, n_env);
  return check_launch("k_post_frame");
}

int tmjx_gae(const float *truncation, const float *termination, const float *rewards, const float *values, const float *bootstrap,
             float lambda_, float discount, float *vs, float *advantages, int T, int B, void *stream) {
  if (!truncation || !termination || !rewards || !values || !bootstrap || !vs || !advantages) return fail(TMJX_EINVAL, "null argument");
  if (T < 1 || B < 1) return fail(TMJX_EINVAL, "T and B must be >= 1");
  hipLaunchKernelGGL(k_gae, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, truncation, termination, rewards, values,
                     bootstrap, lambda_, discount, vs, advantages, T, B);
  return check_launch("k_gae");
}

int tmjx_ppo_scratch_floats(int T, int B) { return 4 * T * B + 4 * ((T * B * PPO_G + PPO_BLOCK - 1) / PPO_BLOCK) + 16 + PPO_REC * ((B + 63) / 64); }

int tmjx_ppo_loss(const tmjx_ppo_cfg_t *cfg, const float *logits, const float *raw_action, const float *behaviour_logp,
                  const float *noise, const float *baseline, const float *bootstrap, const float *reward, const float *discount,
                  const float *truncation, const float *fc2, float *dlogits, float *dbaseline, float *dfc2, float *scratch,
                  float *out, void *stream) {
  if (!cfg || !logits || !raw_action || !behaviour_logp || !noise || !baseline || !bootstrap || !reward || !discount || !truncation ||
      !fc2 || !dlogits || !dbaseline || !dfc2 || !scratch || !out) return fail(TMJX_EINVAL, "null argument");
  if (cfg->T < 1 || cfg->B < 1 || cfg->A < 1 || cfg->Z < 1) return fail(TMJX_EINVAL, "bad T / B / A / Z");
  PpoCfg c{cfg->T, cfg->B, cfg->A, cfg->Z, cfg->reward_scaling, cfg->discounting, cfg->gae_lambda, cfg->clip_eps, cfg->entropy_cost,
           cfg->kl_weight, cfg->normalize_advantage, cfg->accumulate};
  const int N = c.T * c.B, nblk = (N * PPO_G + PPO_BLOCK - 1) / PPO_BLOCK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ppo_a, dim3(nblk), dim3(PPO_BLOCK), 0, s, c, logits, raw_action, noise, fc2, scratch, nblk);
  static const bool one_block_b = getenv("TMJX_PPO_ONE_BLOCK_B") != nullptr;
  if (c.T <= PPO_TMAX && !one_block_b) {
    // GAE / advantage statistics by one 64-thread block per 64 columns, combined by the consumers (csrc/ppo_kernels.h: k_ppo_b2)
    const int nrec = (c.B + 63) / 64;
    float *rec = scratch + 4 * (size_t)N + (size_t)4 * nblk + 16;
    hipLaunchKernelGGL(k_ppo_b2, dim3(nrec), dim3(64), 0, s, c, baseline, bootstrap, reward, discount, truncation, scratch, nblk, rec);
    hipLaunchKernelGGL(k_ppo_c, dim3(nblk), dim3(PPO_BLOCK), 0, s, c, logits, raw_action, behaviour_logp, noise, baseline, fc2, dlogits, dbaseline,
                       dfc2, scratch, nblk, (const float *)rec, nrec);
    hipLaunchKernelGGL(k_ppo_d, dim3(1), dim3(256), 0, s, c, (const float *)scratch, out, nblk, (const float *)rec, nrec);
    return check_launch("k_ppo");
  }
  hipLaunchKernelGGL(k_ppo_b, dim3(1), dim3(1024), 0, s, c, baseline, bootstrap, reward, discount, truncation, scratch, nblk);
  hipLaunchKernelGGL(k_ppo_c, dim3(nblk), dim3(PPO_BLOCK), 0, s, c, logits, raw_action, behaviour_logp, noise, baseline, fc2, dlogits, dbaseline,
                     dfc2, scratch, nblk, (const float *)nullptr, 0);
  hipLaunchKernelGGL(k_ppo_d, dim3(1), dim3(256), 0, s, c, (const float *)scratch, out, nblk, (const float *)nullptr, 0);
  return check_launch("k_ppo");
}

// The loss head in phases, for a caller that runs the two networks on two streams (agent/losses.py: ppo_loss_and_output_grads): A needs the policy's outputs only,
// B the value network's only — it goes onto THAT network's stream, next to the policy's last layers — C needs both, D (the eight scalars) nothing the backward
// pass waits for: it leaves the main stream too.  Same arithmetic as tmjx_ppo_loss; the entropy / KL scalars are summed in D instead of B (a different order
// of the same additions); the gradients are the same bits.  unroll_length <= 24 only.
int tmjx_ppo_loss_phases(const tmjx_ppo_cfg_t *cfg, const float *logits, const float *raw_action, const float *behaviour_logp,
                         const float *noise, const float *baseline, const float *bootstrap, const float *reward, const float *discount,
                         const float *truncation, const float *fc2, float *dlogits, float *dbaseline, float *dfc2, float *scratch,
                         float *out, int phases, void *stream) {
  if (!cfg || !logits || !raw_action || !behaviour_logp || !noise || !baseline || !bootstrap || !reward || !discount || !truncation ||
      !fc2 || !dlogits || !dbaseline || !dfc2 || !scratch || !out) return fail(TMJX_EINVAL, "null argument");
  if (cfg->T < 1 || cfg->B < 1 || cfg->A < 1 || cfg->Z < 1) return fail(TMJX_EINVAL, "bad T / B / A / Z");
  if (cfg->T > PPO_TMAX) return fail(TMJX_EINVAL, "tmjx_ppo_loss_phases: unroll_length > 24 (use tmjx_ppo_loss)");
  if (phases < 1 || phases > 15) return fail(TMJX_EINVAL, "phases: a mask of TMJX_PPO_PHASE_A | _B | _C | _D");
  PpoCfg c{cfg->T, cfg->B, cfg->A, cfg->Z, cfg->reward_scaling, cfg->discounting, cfg->gae_lambda, cfg->clip_eps, cfg->entropy_cost,
           cfg->kl_weight, cfg->normalize_advantage, cfg->accumulate};
  const int N = c.T * c.B, nblk = (N * PPO_G + PPO_BLOCK - 1) / PPO_BLOCK, nrec = (c.B + 63) / 64;
  float *rec = scratch + 4 * (size_t)N + (size_t)4 * nblk + 16;
  hipStream_t s = (hipStream_t)stream;
  // A and C in the same call: one launch (k_ppo_ac; TMJX_PPO_AC=0: two)
  static const bool no_ac = getenv("TMJX_PPO_AC") && atoi(getenv("TMJX_PPO_AC")) == 0;
  if ((phases & TMJX_PPO_PHASE_A) && (phases & TMJX_PPO_PHASE_C) && !no_ac) {
    if (phases & TMJX_PPO_PHASE_B) {       // (B's records first: C's half reads them; A's half does not depend on B)
      hipLaunchKernelGGL(k_ppo_b2, dim3(nrec), dim3(64), 0, s, c, baseline, bootstrap, reward, discount, truncation, scratch, 0, rec);
      phases &= ~TMJX_PPO_PHASE_B;
    }
    hipLaunchKernelGGL(k_ppo_ac, dim3(nblk), dim3(PPO_BLOCK), 0, s, c, logits, raw_action, behaviour_logp, noise, baseline, fc2, dlogits, dbaseline, dfc2, scratch, nblk, (const float *)rec, nrec);
    phases &= ~(TMJX_PPO_PHASE_A | TMJX_PPO_PHASE_C);
  }
  if (phases & TMJX_PPO_PHASE_A) hipLaunchKernelGGL(k_ppo_a, dim3(nblk), dim3(PPO_BLOCK), 0, s, c, logits, raw_action, noise, fc2, scratch, nblk);
  if (phases & TMJX_PPO_PHASE_B) hipLaunchKernelGGL(k_ppo_b2, dim3(nrec), dim3(64), 0, s, c, baseline, bootstrap, reward, discount, truncation, scratch, 0, rec);   // (nblk = 0: no slice of A's partials)
  if (phases & TMJX_PPO_PHASE_C) hipLaunchKernelGGL(k_ppo_c, dim3(nblk), dim3(PPO_BLOCK), 0, s, c, logits, raw_action, behaviour_logp, noise, baseline, fc2, dlogits, dbaseline,
                                                    dfc2, scratch, nblk, (const float *)rec, nrec);
  if (phases & TMJX_PPO_PHASE_D) hipLaunchKernelGGL(k_ppo_d, dim3(1), dim3(256), 0, s, c, (const float *)scratch, out, nblk, (const float *)rec, nrec, 1);
  return check_launch("k_ppo (phases)");
}

int tmjx_silu_ln_partial_floats(int rows, int H) { return ((rows + BLK_ROWS_PER_BLOCK - 1) / BLK_ROWS_PER_BLOCK) * 3 * H; }

static int silu_ln_fwd_any(const float *z, const float *bias, const float *gamma, const float *beta, float *y, uint16_t *y16, int ldy16, float *stats, int rows, int H,
                           float eps, void *stream) {
  if (!z || !bias || !gamma || !beta || (!y && !y16) || !stats) return fail(TMJX_EINVAL, "null argument");
  if (rows < 1) return fail(TMJX_EINVAL, "rows must be >= 1");
  if (y16 && (ldy16 < H || (ldy16 & 3) || ((uintptr_t)y16 & 7))) return fail(TMJX_EINVAL, "tmjx_silu_ln_fwd_bf16: the output's rows must be 8-byte aligned and H wide");
  hipStream_t s = (hipStream_t)stream;
  // One-wave blocks for the acting policy's batches (an env group's rows): next to a GPU full of physics waves — 504 of 512 VGPRs taken on three
  // SIMDs of every CU — a CU has room for new waves on ONE SIMD only, and a 256-thread block wants all four
  const int bt = rows <= TMJX_SMALL_BATCH ? 64 : 256, wpb = bt / 64;
  int grid = (rows + wpb - 1) / wpb; if (grid > 4096) grid = 4096;
  unsigned short *o16 = (unsigned short *)y16;
#define TMJX_FWD(V) do { if (o16) hipLaunchKernelGGL((k_silu_ln_fwd<V, true>), dim3(grid), dim3(bt), 0, s, z, bias, gamma, beta, y, stats, rows, eps, o16, ldy16); \
                         else hipLaunchKernelGGL((k_silu_ln_fwd<V, false>), dim3(grid), dim3(bt), 0, s, z, bias, gamma, beta, y, stats, rows, eps, o16, 0); } while (0)
  switch (H) { case 64: TMJX_FWD(1); break; case 128: TMJX_FWD(2); break; case 256: TMJX_FWD(4); break; case 512: TMJX_FWD(8); break;
               case 1024: TMJX_FWD(16); break; default: return fail(TMJX_EINVAL, "H must be 64, 128, 256, 512 or 1024"); }
#undef TMJX_FWD
  return check_launch("k_silu_ln_fwd");
}
int tmjx_silu_ln_fwd(const float *z, const float *bias, const float *gamma, const float *beta, float *y, float *stats, int rows, int H,
                     float eps, void *stream) {
  if (!y) return fail(TMJX_EINVAL, "null argument");
  return silu_ln_fwd_any(z, bias, gamma, beta, y, nullptr, 0, stats, rows, H, eps, stream);
}
int tmjx_silu_ln_fwd_bf16(const float *z, const float *bias, const float *gamma, const float *beta, uint16_t *y16, int ldy16, float *stats, int rows, int H,
                          float eps, void *stream) {
  if (!y16) return fail(TMJX_EINVAL, "null argument");
  return silu_ln_fwd_any(z, bias, gamma, beta, nullptr, y16, ldy16, stats, rows, H, eps, stream);
}

static int silu_ln_bwd_any(const float *dy, const float *z, const float *bias, const float *gamma, const float *stats, float *dz, uint16_t *dz16, int lddz16, float *grads,
                           float *partial, int rows, int H, void *stream) {
  if (!dy || !z || !bias || !gamma || !stats || (!dz && !dz16) || !grads || !partial) return fail(TMJX_EINVAL, "null argument");
  if (rows < 1) return fail(TMJX_EINVAL, "rows must be >= 1");
  if (dz16 && (lddz16 < H || (lddz16 & 3) || ((uintptr_t)dz16 & 7))) return fail(TMJX_EINVAL, "tmjx_silu_ln_bwd_bf16: the output's rows must be 8-byte aligned and H wide");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (rows + BLK_ROWS_PER_BLOCK - 1) / BLK_ROWS_PER_BLOCK;
  unsigned short *o16 = (unsigned short *)dz16;
#define TMJX_BWD(V) do { if (o16) hipLaunchKernelGGL((k_silu_ln_bwd<V, true>), dim3(nblk), dim3(256), 0, s, dy, z, bias, gamma, stats, dz, partial, rows, o16, lddz16); \
                         else hipLaunchKernelGGL((k_silu_ln_bwd<V, false>), dim3(nblk), dim3(256), 0, s, dy, z, bias, gamma, stats, dz, partial, rows, o16, 0); } while (0)
  switch (H) { case 64: TMJX_BWD(1); break; case 128: TMJX_BWD(2); break; case 256: TMJX_BWD(4); break; case 512: TMJX_BWD(8); break;
               case 1024: TMJX_BWD(16); break; default: return fail(TMJX_EINVAL, "H must be 64, 128, 256, 512 or 1024"); }
#undef TMJX_BWD
  hipLaunchKernelGGL(k_colsum, dim3((3 * H + 31) / 32), dim3(256), 0, s, (const float *)partial, grads, nblk, 3 * H);
  return check_launch("k_silu_ln_bwd");
}
int tmjx_silu_ln_bwd(const float *dy, const float *z, const float *bias, const float *gamma, const float *stats, float *dz, float *grads,
                     float *partial, int rows, int H, void *stream) {
  if (!dz) return fail(TMJX_EINVAL, "null argument");
  return silu_ln_bwd_any(dy, z, bias, gamma, stats, dz, nullptr, 0, grads, partial, rows, H, stream);
}
int tmjx_silu_ln_bwd_bf16(const float *dy, const float *z, const float *bias, const float *gamma, const float *stats, uint16_t *dz16, int lddz16, float *grads,
                          float *partial, int rows, int H, void *stream) {
  if (!dz16) return fail(TMJX_EINVAL, "null argument");
  return silu_ln_bwd_any(dy, z, bias, gamma, stats, nullptr, dz16, lddz16, grads, partial, rows, H, stream);
}

int tmjx_gather_normalize(const float *src, const int64_t *idx, const float *mean, const float *std, float *out, int T, int R, int B,
                          int W, void *stream) {
  if (!src || !idx || !mean || !std || !out) return fail(TMJX_EINVAL, "null argument");
  if (T < 1 || R < 1 || B < 1 || W < 4 || (W & 3)) return fail(TMJX_EINVAL, "bad T / R / B / W (W must be a multiple of 4)");
  size_t total = (size_t)T * B * (W >> 2);
  int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(k_gather_normalize, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (const long long *)idx, mean, std, out, T, R, B, W);
  return check_launch("k_gather_normalize");
}

static int launch_minibatch(const MinibatchGather &g, hipStream_t s) {
  if (!g.obs || !g.next_last || !g.raw_action || !g.scalar[0] || !g.scalar[1] || !g.scalar[2] || !g.scalar[3] || !g.idx || !g.mean || !g.stdv || !g.obs_n || !g.next_n ||
      !g.raw_action_g || !g.scalars_g)
    return fail(TMJX_EINVAL, "null argument");
  if (g.T < 1 || g.R < 1 || g.B < 1 || g.A < 1 || g.W < 4 || (g.W & 3)) return fail(TMJX_EINVAL, "bad T / R / B / A / W (W must be a multiple of 4)");
  if (g.eps && g.Z < 1) return fail(TMJX_EINVAL, "Z must be >= 1 with eps");
  const size_t nact = (size_t)g.T * g.B * g.A;
  const size_t total = (size_t)g.T * g.B * (g.W >> 2) + (size_t)g.B * (g.W >> 2) + nact + (size_t)4 * g.T * g.B + (g.eps ? ((size_t)g.T * g.B * g.Z + 3) / 4 : 0) +
                       (g.noise ? (nact + 3) / 4 : 0);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(k_gather_minibatch, dim3(grid), dim3(256), 0, s, g);
  return check_launch("k_gather_minibatch");
}
int tmjx_gather_minibatch(const float *obs, const float *next_last, const float *raw_action, const float *log_prob, const float *reward, const float *discount,
                          const float *truncation, const int64_t *idx, const float *mean, const float *std, float *obs_n, float *next_n, float *raw_action_g,
                          float *scalars_g, int T, int R, int B, int W, int A, void *stream) {
  MinibatchGather g{obs, next_last, raw_action, {log_prob, reward, discount, truncation}, (const long long *)idx, mean, std, obs_n, next_n, raw_action_g, scalars_g,
                    nullptr, nullptr, nullptr, 0ull, T, R, B, W, A, 0, 0, nullptr, 0};
  return launch_minibatch(g, (hipStream_t)stream);
}
int tmjx_minibatch_begin(const tmjx_minibatch_t *m, void *stream) {
  if (!m) return fail(TMJX_EINVAL, "null argument");
  if ((m->eps || m->noise || m->advance) && !m->state) return fail(TMJX_EINVAL, "tmjx_minibatch_begin: draws / advance need the device state");
  MinibatchGather g{m->obs, m->next_last, m->raw_action, {m->log_prob, m->reward, m->discount, m->truncation}, (const long long *)m->perm, m->mean, m->std, m->obs_n,
                    m->next_n, m->raw_action_g, m->scalars_g, m->eps, m->noise, (long long *)m->state, (unsigned long long)m->seed, m->T, m->R, m->B, m->W, m->A,
                    m->Z, m->advance, nullptr, 0};
  return launch_minibatch(g, (hipStream_t)stream);
}
int tmjx_minibatch_begin_bf16(const tmjx_minibatch_t *m, uint16_t *obs_n16, int ld16, void *stream) {
  if (!m || !obs_n16) return fail(TMJX_EINVAL, "null argument");
  if ((m->eps || m->noise || m->advance) && !m->state) return fail(TMJX_EINVAL, "tmjx_minibatch_begin_bf16: draws / advance need the device state");
  if (ld16 < m->W || (ld16 & 3) || ((uintptr_t)obs_n16 & 7)) return fail(TMJX_EINVAL, "tmjx_minibatch_begin_bf16: the twin's rows must be 8-byte aligned and W wide");
  MinibatchGather g{m->obs, m->next_last, m->raw_action, {m->log_prob, m->reward, m->discount, m->truncation}, (const long long *)m->perm, m->mean, m->std, m->obs_n,
                    m->next_n, m->raw_action_g, m->scalars_g, m->eps, m->noise, (long long *)m->state, (unsigned long long)m->seed, m->T, m->R, m->B, m->W, m->A,
                    m->Z, m->advance, (unsigned short *)obs_n16, ld16};
  return launch_minibatch(g, (hipStream_t)stream);
}
int tmjx_philox4x32_10(const uint32_t *ctr_key_dev, uint32_t *out_dev, void *stream) {
  if (!ctr_key_dev || !out_dev) return fail(TMJX_EINVAL, "null argument");
  hipLaunchKernelGGL(k_philox_kat, dim3(1), dim3(64), 0, (hipStream_t)stream, ctr_key_dev, out_dev);
  return check_launch("k_philox_kat");
}

int tmjx_colsum_scratch_floats(int width) { return COLSUM_CHUNKS * width; }

int tmjx_colsum(const float *src, float *out, float *scratch, int rows, int width, void *stream) {
  if (!src || !out || !scratch) return fail(TMJX_EINVAL, "null argument");
  if (rows < 1 || width < 1) return fail(TMJX_EINVAL, "rows and width must be >= 1");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_colsum_rows, dim3((width + 31) / 32, COLSUM_CHUNKS), dim3(256), 0, s, src, scratch, rows, width);
  hipLaunchKernelGGL(k_colsum, dim3((width + 31) / 32), dim3(256), 0, s, (const float *)scratch, out, COLSUM_CHUNKS, width);
  return check_launch("k_colsum_rows");
}

int tmjx_colsum_grouped(const tmjx_colsum_problem_t *q, int n, void *stream) {
  if (!q || n < 1 || n > COLSUM_GROUP_MAX) return fail(TMJX_EINVAL, "tmjx_colsum_grouped: 1 .. 16 problems");
  ColsumGroup G;
  G.n = n;
  int blocks = 0;
  for (int i = 0; i < n; i++) {
    if (!q[i].partial || !q[i].out || q[i].rows < 1 || q[i].width < 1) return fail(TMJX_EINVAL, "tmjx_colsum_grouped: bad problem");
    G.p[i] = ColsumProblem{q[i].partial, q[i].out, q[i].rows, q[i].width, blocks};
    blocks += (q[i].width + 31) / 32;
  }
  hipLaunchKernelGGL(k_colsum_grouped, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G);
  return check_launch("k_colsum_grouped");
}

int tmjx_adam_clip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *grad_norm, long long n, float lr,
                   float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, float max_norm, void *stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !grad_norm) return fail(TMJX_EINVAL, "null argument");
  if (n < 1 || !(bias_correction1 > 0.f) || !(bias_correction2 > 0.f) || !(max_norm > 0.f)) return fail(TMJX_EINVAL, "bad n / bias corrections / max_norm");
  int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_adam_clip<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, grad_norm, n, lr, beta1, beta2, eps,
                     bias_correction1, bias_correction2, max_norm, (float *)nullptr);
  return check_launch("k_adam_clip");
}
int tmjx_adam_norm_floats(void) { return ADAM_NORM_PARTS; }
int tmjx_adam_clip_norm(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *norm_scratch, float *norm_out, long long n, float lr,
                        float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, float max_norm, void *stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !norm_scratch) return fail(TMJX_EINVAL, "null argument");
  if (n < 1 || !(bias_correction1 > 0.f) || !(bias_correction2 > 0.f) || !(max_norm > 0.f)) return fail(TMJX_EINVAL, "bad n / bias corrections / max_norm");
  if ((uintptr_t)grad & 15) return fail(TMJX_EINVAL, "tmjx_adam_clip_norm: the gradient buffer must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_grad_sumsq, dim3(ADAM_NORM_PARTS), dim3(256), 0, s, grad, n, norm_scratch);
  int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_adam_clip<true>, dim3(grid), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, (const float *)norm_scratch, n, lr, beta1, beta2, eps,
                     bias_correction1, bias_correction2, max_norm, norm_out);
  return check_launch("k_adam_clip(norm)");
}

int tmjx_latent_concat(const float *fc2, const float *eps, const float *obs, float *x, int n, int Z, int obs_w, int ref_w,
                       int64_t obs_s0, int64_t obs_s1, const float *mean, const float *std, int x_stride, uint64_t seed, const int64_t *rng_state,
                       void *stream) {
  if (!fc2 || !obs || !x) return fail(TMJX_EINVAL, "null argument");
  if (!eps && !rng_state) return fail(TMJX_EINVAL, "tmjx_latent_concat: eps == NULL needs rng_state");
  if (n < 1 || Z < 1 || ref_w < 0 || obs_w < ref_w || x_stride < Z + obs_w - ref_w) return fail(TMJX_EINVAL, "bad sizes");
  size_t total = (size_t)n * x_stride;
  const size_t bt = n <= TMJX_SMALL_BATCH ? 64 : 256;        // (one-wave blocks for an env group's rows: see tmjx_silu_ln_fwd)
  int grid = (int)((total + bt - 1) / bt < 4096 ? (total + bt - 1) / bt : 4096);
  hipLaunchKernelGGL(k_latent_concat, dim3(grid), dim3((unsigned)bt), 0, (hipStream_t)stream, fc2, eps, obs, x, n, Z, obs_w, ref_w, (long long)obs_s0, (long long)obs_s1, mean, std, x_stride,
                     (unsigned long long)seed, (const long long *)rng_state);
  return check_launch("k_latent_concat");
}

static int launch_latent_concat_bwd(const float *dx, const float *eps, const float *fc2, const float *add, float *dfc2, int n, int Z, int dx_stride, void *stream) {
  if (!dx || !eps || !fc2 || !dfc2) return fail(TMJX_EINVAL, "null argument");
  if (n < 1 || Z < 1 || dx_stride < Z) return fail(TMJX_EINVAL, "bad sizes");
  size_t total = (size_t)n * Z;
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_latent_concat_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, dx, eps, fc2, dfc2, n, Z, dx_stride, add);
  return check_launch("k_latent_concat_bwd");
}
int tmjx_latent_concat_bwd(const float *dx, const float *eps, const float *fc2, float *dfc2, int n, int Z, int dx_stride, void *stream) {
  return launch_latent_concat_bwd(dx, eps, fc2, nullptr, dfc2, n, Z, dx_stride, stream);
}
int tmjx_latent_concat_bwd_add(const float *dx, const float *eps, const float *fc2, const float *add, float *dfc2, int n, int Z, int dx_stride, void *stream) {
  if (!add) return fail(TMJX_EINVAL, "null argument");
  return launch_latent_concat_bwd(dx, eps, fc2, add, dfc2, n, Z, dx_stride, stream);
}

int tmjx_sample_action(const float *logits, const float *noise, float *raw, float *action_t, float *logp, int n, int A, uint64_t seed,
                       int64_t *rng_state, void *stream) {
  if (!logits || !raw || !action_t || !logp) return fail(TMJX_EINVAL, "null argument");
  if (!noise && !rng_state) return fail(TMJX_EINVAL, "tmjx_sample_action: noise == NULL needs rng_state");
  if (n < 1 || A < 1) return fail(TMJX_EINVAL, "bad sizes");
  const int sbt = n <= TMJX_SMALL_BATCH ? 64 : PPO_BLOCK;      // (one-wave blocks for an env group's rows: see tmjx_silu_ln_fwd)
  hipLaunchKernelGGL(k_sample_action, dim3((n * PPO_G + sbt - 1) / sbt), dim3(sbt), 0, (hipStream_t)stream, logits, noise, raw, action_t, logp, n, A,
                     (unsigned long long)seed, (long long *)(noise ? nullptr : rng_state));
  return check_launch("k_sample_action");
}

// The same with the operand normalised on the fly: (A - mean[k]) * inv_std[k] (the acting policy's first layer reading the RAW observation).
// Only the matrix-core variant: K % 4 == 0, 16-byte aligned W / mean / inv_std, A row-major with aligned rows or K-major.
int tmjx_linear_nolds_norm(const float *A, int64_t sa_row, int64_t sa_k, const float *W, const float *bias, float *C, int M, int N, int K,
                           const float *mean, const float *inv_std, void *stream) {
  if (!A || !W || !C || !mean || !inv_std) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1) return fail(TMJX_EINVAL, "bad sizes");
  const bool kmajor = sa_row == 1 && sa_k != 1;
  if (!kmajor && sa_k != 1) return fail(TMJX_EINVAL, "A must be row-major (sa_k == 1) or K-major (sa_row == 1)");
  if ((K & 3) || ((uintptr_t)W & 15) || ((uintptr_t)mean & 15) || ((uintptr_t)inv_std & 15) || (!kmajor && ((sa_row & 3) || ((uintptr_t)A & 15))))
    return fail(TMJX_EINVAL, "tmjx_linear_nolds_norm: K % 4 == 0 and 16-byte aligned operands");
  hipStream_t s = (hipStream_t)stream;
  dim3 g2((M + 31) / 32, (N + 31) / 32);
  if (kmajor) hipLaunchKernelGGL((k_linear_nolds_mfma<true>), g2, dim3(64), 0, s, A, (long long)sa_row, (long long)sa_k, W, bias, C, M, N, K, mean, inv_std);
  else hipLaunchKernelGGL((k_linear_nolds_mfma<false>), g2, dim3(64), 0, s, A, (long long)sa_row, (long long)sa_k, W, bias, C, M, N, K, mean, inv_std);
  return check_launch("k_linear_nolds_mfma(norm)");
}
// The acting policy's layer through a 20 KB LDS tile (k_linear_act, csrc/ppo_kernels.h): C[M][N] = op(A)[M][K] W[N][ldw]^T (+ bias), op = identity or
// (A - mean) * inv_std.  Returns TMJX_EINVAL for operands it does not take (the caller then uses tmjx_linear_nolds): tmjx_linear_act_ok says which.
int tmjx_linear_act_ok(const float *A, int64_t lda, const float *W, int ldw, int K) {
  return !(K & 3) && !((uintptr_t)A & 15) && !(lda & 3) && !((uintptr_t)W & 15) && !(ldw & 3) && lda >= K && ldw >= K;
}
int tmjx_linear_act(const float *A, int64_t lda, const float *W, int ldw, const float *bias, float *C, int M, int N, int K, const float *mean,
                    const float *inv_std, void *stream) {
  if (!A || !W || !C) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1) return fail(TMJX_EINVAL, "bad sizes");
  if (!tmjx_linear_act_ok(A, lda, W, ldw, K)) return fail(TMJX_EINVAL, "tmjx_linear_act: K % 4 == 0, row-major operands with 16-byte aligned rows");
  if ((mean || inv_std) && (!mean || !inv_std || (((uintptr_t)mean | (uintptr_t)inv_std) & 15))) return fail(TMJX_EINVAL, "tmjx_linear_act: mean and inv_std together, 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // 32-row tiles while 64-row ones would leave CUs without a workgroup (the acting policy's 1 365 rows: 88 -> 172 workgroups); TMJX_ACT_ROWS=64 / 32 forces one
  static const int force = getenv("TMJX_ACT_ROWS") ? atoi(getenv("TMJX_ACT_ROWS")) : 0;
  const int ncol = (N + ACT_BM - 1) / ACT_BM;
  const bool small = force ? force == 32 : ((M + 63) / 64) * ncol < 128;
  if (small) {
    dim3 grid((M + 31) / 32, ncol);
    if (mean) hipLaunchKernelGGL((k_linear_act<true, 32>), grid, dim3(256), 0, s, A, (long long)lda, W, ldw, bias, C, M, N, K, mean, inv_std);
    else hipLaunchKernelGGL((k_linear_act<false, 32>), grid, dim3(256), 0, s, A, (long long)lda, W, ldw, bias, C, M, N, K, (const float *)nullptr, (const float *)nullptr);
  } else {
    dim3 grid((M + ACT_BM - 1) / ACT_BM, ncol);
    if (mean) hipLaunchKernelGGL((k_linear_act<true, 64>), grid, dim3(256), 0, s, A, (long long)lda, W, ldw, bias, C, M, N, K, mean, inv_std);
    else hipLaunchKernelGGL((k_linear_act<false, 64>), grid, dim3(256), 0, s, A, (long long)lda, W, ldw, bias, C, M, N, K, (const float *)nullptr, (const float *)nullptr);
  }
  return check_launch("k_linear_act");
}
int tmjx_linear_nolds(const float *A, int64_t sa_row, int64_t sa_k, const float *W, const float *bias, float *C, int M, int N, int K,
                      void *stream) {
  if (!A || !W || !C) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1) return fail(TMJX_EINVAL, "bad sizes");
  const bool kmajor = sa_row == 1 && sa_k != 1;
  if (!kmajor && sa_k != 1) return fail(TMJX_EINVAL, "A must be row-major (sa_k == 1) or K-major (sa_row == 1)");
  if (kmajor && ((M & 3) || (sa_k & 3) || ((uintptr_t)A & 15))) return fail(TMJX_EINVAL, "K-major A needs M % 4 == 0 and 16-byte aligned columns");
  // widest vector along K that keeps every row of W (and of a row-major A) aligned
  int V = 1;
  if (!(K & 3) && !((uintptr_t)W & 15) && (kmajor || (!(sa_row & 3) && !((uintptr_t)A & 15)))) V = 4;
  else if (!(K & 1) && !((uintptr_t)W & 7) && (kmajor || (!(sa_row & 1) && !((uintptr_t)A & 7)))) V = 2;
  hipStream_t s = (hipStream_t)stream;
  if (V == 4 && !getenv("TMJX_NOLDS_VALU")) {      // matrix-core variant: float4 operands along K
    dim3 g2((M + 31) / 32, (N + 31) / 32);
    if (kmajor) hipLaunchKernelGGL((k_linear_nolds_mfma<true>), g2, dim3(64), 0, s, A, (long long)sa_row, (long long)sa_k, W, bias, C, M, N, K, (const float *)nullptr, (const float *)nullptr);
    else hipLaunchKernelGGL((k_linear_nolds_mfma<false>), g2, dim3(64), 0, s, A, (long long)sa_row, (long long)sa_k, W, bias, C, M, N, K, (const float *)nullptr, (const float *)nullptr);
    return check_launch("k_linear_nolds_mfma");
  }
  dim3 grid((M + 63) / 64, (N + 31) / 32), block(128);
#define TMJX_NL(VV, KM) hipLaunchKernelGGL((k_linear_nolds<VV, KM>), grid, block, 0, s, A, (long long)sa_row, (long long)sa_k, W, bias, C, M, N, K)
  if (kmajor) { if (V == 4) TMJX_NL(4, true); else if (V == 2) TMJX_NL(2, true); else TMJX_NL(1, true); }
  else { if (V == 4) TMJX_NL(4, false); else if (V == 2) TMJX_NL(2, false); else TMJX_NL(1, false); }
#undef TMJX_NL
  return check_launch("k_linear_nolds");
}

}  // extern "C" (the templates below need C++ linkage)
// ---- dense layers of the learner on the matrix cores (csrc/gemm_kernels.h)
// vector (dwordx4) loads of a row-major operand: 16-byte aligned rows (pointer and leading dimension) — then ld >= cols rounded up to 4
// holds by itself (ld is a multiple of 4 and >= cols), i.e. the last float4 of a row stays inside the row's allocation
static bool aligned16(const void *p, long long ld) { return !((uintptr_t)p & 15) && !(ld & 3); }
// Rows per workgroup tile (16 MT): 80 where that fills the chip (20 480 rows = 256 workgroups, one per CU), 32 where 80-row tiles would leave most
// CUs idle — 5 120 rows (one rank's share of the 8-GPU configuration: batch_size 2048 / 8 x unroll_length 20) are 64 workgroups of 80 rows but
// 160 of 32.  Cost model: rounds of 256 workgroups x the tile's MFMA time (a 32-row tile streams the same weight tile per K step as an 80-row one,
// ~ 10 % over its 2 / 5 share).  TMJX_GEMM_MT=5 / 2 forces either (tuning, tests).
static int gemm_mt(int M, int col_tiles) {
  static const int forced = getenv("TMJX_GEMM_MT") ? atoi(getenv("TMJX_GEMM_MT")) : 0;
  if (forced == 5 || forced == 2) return forced;
  const long long w5 = (long long)((M + 79) / 80) * col_tiles, w2 = (long long)((M + 31) / 32) * col_tiles;
  const double c5 = (double)((w5 + 255) / 256) * 5.0, c2 = (double)((w2 + 255) / 256) * 2.2;
  return c2 < c5 ? 2 : 5;
}
extern "C" int tmjx_internal_gemm_mt(int M, int col_tiles) { return gemm_mt(M, col_tiles); }      // (tmjx_chain.hip: the chain kernels take the same row tile)
template <int NIW, bool BT, bool AVEC, bool WVEC, int MT = 5, int EPI = 0>
static int launch_gemm_act(const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc, int M, int N, int K, hipStream_t s, GemmLN ln = GemmLN{}) {
  constexpr int BN = 64 * NIW, BM = 16 * MT;
  constexpr size_t lds = 2 * sizeof(float) * (size_t)((BM + 8) * GEMM_LDA + (BT ? BN * GEMM_LDA : GEMM_BK * (BN + 4)));
  static bool attr_set = false;            // > 64 KiB of dynamic LDS needs the attribute once per kernel
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_gemm_act<NIW, BT, AVEC, WVEC, EPI, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_gemm_act): ") + hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipLaunchKernelGGL((k_gemm_act<NIW, BT, AVEC, WVEC, EPI, MT>), grid, dim3(GemmCfg<NIW>::THREADS), lds, s, A, lda, W, ldw, bias, C, ldc, M, N, K, ln);
  return check_launch("k_gemm_act");
}
// the input gradient of a Dense -> SiLU layer's consumer with that layer's SiLU backward in the epilogue (k_gemm_act<.., EPI = 4>; aligned operands only)
template <int NIW>
static int launch_gemm_nn_silu_bwd(const float *dY, int ldy, const float *W, int ldw, const float *bias, float *dZ, int M, int N, int K, const float *z, hipStream_t s) {
  GemmLN ln{};
  ln.z = z;
  if (gemm_mt(M, (N + 64 * NIW - 1) / (64 * NIW)) == 2) return launch_gemm_act<NIW, false, true, true, 2, 4>(dY, ldy, W, ldw, bias, dZ, N, M, N, K, s, ln);
  return launch_gemm_act<NIW, false, true, true, 5, 4>(dY, ldy, W, ldw, bias, dZ, N, M, N, K, s, ln);
}
template <int NIW, bool BT>
static int gemm_act_vec(const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc, int M, int N, int K, hipStream_t s) {
  const bool av = aligned16(A, lda), wv = aligned16(W, ldw);
  if (av && wv) {        // (the 32-row tile exists for the aligned form only: every operand of the learner is)
    if (gemm_mt(M, (N + 64 * NIW - 1) / (64 * NIW)) == 2) return launch_gemm_act<NIW, BT, true, true, 2>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
    return launch_gemm_act<NIW, BT, true, true>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
  }
  if (av) return launch_gemm_act<NIW, BT, true, false>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
  if (wv) return launch_gemm_act<NIW, BT, false, true>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
  return launch_gemm_act<NIW, BT, false, false>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
}
template <bool BT>
static int gemm_act(const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc, int M, int N, int K, void *stream) {
  if (!A || !W || !C) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1 || lda < K || ldc < N || ldw < (BT ? K : N)) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  hipStream_t s = (hipStream_t)stream;
  if (N <= 64) return gemm_act_vec<1, BT>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
  if (N <= 128) return gemm_act_vec<2, BT>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
  return gemm_act_vec<4, BT>(A, lda, W, ldw, bias, C, ldc, M, N, K, s);
}
// Dense -> SiLU -> LayerNorm forward in one launch (k_gemm_act<.., LN = true>): the layer must be exactly one tile wide
template <int NIW, int MT>
static int launch_gemm_ln(const float *A, int lda, const float *W, int ldw, const float *bias, float *Z, int ldc, int M, int N, int K, GemmLN ln, hipStream_t s) {
  constexpr int BN = 64 * NIW, BM = 16 * MT;
  constexpr size_t lds = 2 * sizeof(float) * (size_t)((BM + 8) * GEMM_LDA + BN * GEMM_LDA);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_gemm_act<NIW, true, true, true, 1, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_gemm_act LN): ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm_act<NIW, true, true, true, 1, MT>), dim3((M + BM - 1) / BM, 1), dim3(GemmCfg<NIW>::THREADS), lds, s, A, lda, W, ldw, bias, Z, ldc, M, N, K, ln);
  return check_launch("k_gemm_act(LN)");
}
template <bool YVEC, bool XVEC>
static int launch_gemm_dw(const float *dY, int ldy, const float *X, int ldx, float *scratch, int M, int N, int K, int with_bias, int rps, int S, int ld, hipStream_t s) {
  constexpr size_t lds = 2 * sizeof(float) * 2 * DW_BM * DW_LD;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_gemm_dw<YVEC, XVEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_gemm_dw): ") + hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid((N + DW_BT - 1) / DW_BT, (K + DW_BT - 1) / DW_BT, S);
  hipLaunchKernelGGL((k_gemm_dw<YVEC, XVEC>), grid, dim3(512), lds, s, dY, ldy, X, ldx, scratch, M, N, K, with_bias, rps, ld);
  return TMJX_OK;
}
template <int NIW, int MT>
static int launch_gemm_silu(const float *A, int lda, const float *W, int ldw, const float *bias, float *Z, float *Y, int ldc, int M, int N, int K, hipStream_t s) {
  constexpr int BN = 64 * NIW, BM = 16 * MT;
  constexpr size_t lds = 2 * sizeof(float) * (size_t)((BM + 8) * GEMM_LDA + BN * GEMM_LDA);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_gemm_act<NIW, true, true, true, 3, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_gemm_act SiLU): ") + hipGetErrorString(e));
    attr_set = true;
  }
  GemmLN ln{nullptr, nullptr, Y, nullptr, 0.f, nullptr, nullptr};
  hipLaunchKernelGGL((k_gemm_act<NIW, true, true, true, 3, MT>), dim3((M + BM - 1) / BM, (N + BN - 1) / BN), dim3(GemmCfg<NIW>::THREADS), lds, s, A, lda, W, ldw, bias, Z, ldc, M, N, K, ln);
  return check_launch("k_gemm_act(SiLU)");
}
extern "C" {
int tmjx_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias, float *C, int ldc, int M, int N, int K, void *stream) {
  return gemm_act<true>(A, lda, W, ldw, bias, C, ldc, M, N, K, stream);
}
int tmjx_gemm_nt_silu_ln_ok(const float *A, int lda, const float *W, int ldw, int N) {
  return (N == 64 || N == 128 || N == 256) && aligned16(A, lda) && aligned16(W, ldw);
}
int tmjx_gemm_nt_silu_ln(const float *A, int lda, const float *W, int ldw, const float *bias, const float *gamma, const float *beta, float *Z, float *Y,
                         int ldc, float *stats, int M, int N, int K, float eps, void *stream) {
  if (!A || !W || !bias || !gamma || !beta || !Z || !Y || !stats) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || K < 1 || lda < K || ldc < N || ldw < K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  if (!tmjx_gemm_nt_silu_ln_ok(A, lda, W, ldw, N)) return fail(TMJX_EINVAL, "tmjx_gemm_nt_silu_ln: N must be 64, 128 or 256 and the operands' rows 16-byte aligned");
  GemmLN ln{gamma, beta, Y, stats, eps, nullptr, nullptr};
  hipStream_t s = (hipStream_t)stream;
  if (gemm_mt(M, 1) == 2) {
    if (N == 64) return launch_gemm_ln<1, 2>(A, lda, W, ldw, bias, Z, ldc, M, N, K, ln, s);
    if (N == 128) return launch_gemm_ln<2, 2>(A, lda, W, ldw, bias, Z, ldc, M, N, K, ln, s);
    return launch_gemm_ln<4, 2>(A, lda, W, ldw, bias, Z, ldc, M, N, K, ln, s);
  }
  if (N == 64) return launch_gemm_ln<1, 5>(A, lda, W, ldw, bias, Z, ldc, M, N, K, ln, s);
  if (N == 128) return launch_gemm_ln<2, 5>(A, lda, W, ldw, bias, Z, ldc, M, N, K, ln, s);
  return launch_gemm_ln<4, 5>(A, lda, W, ldw, bias, Z, ldc, M, N, K, ln, s);
}
// Dense -> SiLU forward in one launch (k_gemm_act<.., EPI = 3>): Z = A W^T (without the bias), Y = silu(Z + bias); any N, 16-byte aligned operand rows
int tmjx_gemm_nt_silu_ok(const float *A, int lda, const float *W, int ldw) { return aligned16(A, lda) && aligned16(W, ldw); }
int tmjx_gemm_nt_silu(const float *A, int lda, const float *W, int ldw, const float *bias, float *Z, float *Y, int ldc, int M, int N, int K, void *stream) {
  if (!A || !W || !bias || !Z || !Y) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1 || lda < K || ldc < N || ldw < K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  if (!tmjx_gemm_nt_silu_ok(A, lda, W, ldw)) return fail(TMJX_EINVAL, "tmjx_gemm_nt_silu: the operands' rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int niw = N <= 64 ? 1 : N <= 128 ? 2 : 4;
  if (gemm_mt(M, (N + 64 * niw - 1) / (64 * niw)) == 2) {
    if (niw == 1) return launch_gemm_silu<1, 2>(A, lda, W, ldw, bias, Z, Y, ldc, M, N, K, s);
    if (niw == 2) return launch_gemm_silu<2, 2>(A, lda, W, ldw, bias, Z, Y, ldc, M, N, K, s);
    return launch_gemm_silu<4, 2>(A, lda, W, ldw, bias, Z, Y, ldc, M, N, K, s);
  }
  if (niw == 1) return launch_gemm_silu<1, 5>(A, lda, W, ldw, bias, Z, Y, ldc, M, N, K, s);
  if (niw == 2) return launch_gemm_silu<2, 5>(A, lda, W, ldw, bias, Z, Y, ldc, M, N, K, s);
  return launch_gemm_silu<4, 5>(A, lda, W, ldw, bias, Z, Y, ldc, M, N, K, s);
}
int tmjx_silu_fwd(const float *z, const float *bias, float *y, long long rows, int N, void *stream) {
  if (!z || !bias || !y) return fail(TMJX_EINVAL, "null argument");
  if (rows < 1 || N < 1) return fail(TMJX_EINVAL, "bad sizes");
  const long long total = rows * N;
  hipLaunchKernelGGL(k_silu_fwd_f32, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, z, bias, y, total, N);
  return check_launch("k_silu_fwd_f32");
}
// dZ[M][N] = (dY[M][K] W[K][N]) silu'(z + bias): the consumer's input gradient and the producing Dense -> SiLU layer's backward in ONE launch; z, dZ dense [M][N]
int tmjx_gemm_nn_silu_bwd_ok(const float *dY, int ldy, const float *W, int ldw) { return aligned16(dY, ldy) && aligned16(W, ldw); }
int tmjx_gemm_nn_silu_bwd(const float *dY, int ldy, const float *W, int ldw, const float *z, const float *bias, float *dZ, int M, int N, int K, void *stream) {
  if (!dY || !W || !z || !bias || !dZ) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1 || ldy < K || ldw < N) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  if (!tmjx_gemm_nn_silu_bwd_ok(dY, ldy, W, ldw)) return fail(TMJX_EINVAL, "tmjx_gemm_nn_silu_bwd: the operands' rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (N <= 64) return launch_gemm_nn_silu_bwd<1>(dY, ldy, W, ldw, bias, dZ, M, N, K, z, s);
  if (N <= 128) return launch_gemm_nn_silu_bwd<2>(dY, ldy, W, ldw, bias, dZ, M, N, K, z, s);
  return launch_gemm_nn_silu_bwd<4>(dY, ldy, W, ldw, bias, dZ, M, N, K, z, s);
}
// dz[m][k] = (dy1[m] w1[k]) silu'(z[m][k] + bias[k]): the SiLU backward of the last hidden layer of an MLP with a 1-wide head (N % 4 == 0, 16-byte aligned arrays)
int tmjx_silu_bwd_rank1(const float *dy1, const float *w1, const float *z, const float *bias, float *dz, long long rows, int N, void *stream) {
  if (!dy1 || !w1 || !z || !bias || !dz) return fail(TMJX_EINVAL, "null argument");
  if (rows < 1 || N < 4 || (N & 3) || (((uintptr_t)w1 | (uintptr_t)z | (uintptr_t)bias | (uintptr_t)dz) & 15)) return fail(TMJX_EINVAL, "tmjx_silu_bwd_rank1: N % 4 == 0 and 16-byte aligned arrays");
  const long long total = rows * N;
  hipLaunchKernelGGL(k_silu_bwd_rank1_f32, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, dy1, w1, z, bias, dz, total, N);
  return check_launch("k_silu_bwd_rank1_f32");
}
// the 1-wide head's gradients: dw[k] = sum_m dy1[m] x[m][k], db[0] = sum_m dy1[m] (db may be NULL); scratch >= tmjx_head_dw_scratch_floats(M, K) floats
long long tmjx_head_dw_scratch_floats(int M, int K) { return (M < 1 || K < 1) ? 0 : (long long)((M + HEAD_DW_ROWS - 1) / HEAD_DW_ROWS) * (K + 1); }
int tmjx_head_dw(const float *dy1, const float *x, int ldx, float *dw, float *db, float *scratch, int M, int K, void *stream) {
  if (!dy1 || !x || !dw || !scratch) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || K < 1 || ldx < K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  const int slabs = (M + HEAD_DW_ROWS - 1) / HEAD_DW_ROWS;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_head_dw, dim3(slabs), dim3(256), 0, s, dy1, x, ldx, scratch, M, K);
  hipLaunchKernelGGL(k_head_dw_reduce, dim3((K + 1 + 31) / 32), dim3(256), 0, s, (const float *)scratch, dw, db, slabs, K);
  return check_launch("k_head_dw");
}
// y[m] = x[m][:K] . w[:K] + bias[0] (bias may be NULL): the 1-wide head's forward pass; K % 4 == 0, x rows and w 16-byte aligned
int tmjx_head_fwd_ok(const float *x, int ldx, const float *w, int K) { return !(K & 3) && aligned16(x, ldx) && !((uintptr_t)w & 15); }
int tmjx_head_fwd(const float *x, int ldx, const float *w, const float *bias, float *y, int M, int K, void *stream) {
  if (!x || !w || !y) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || K < 1 || ldx < K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  if (!tmjx_head_fwd_ok(x, ldx, w, K)) return fail(TMJX_EINVAL, "tmjx_head_fwd: K % 4 == 0 and 16-byte aligned rows");
  hipLaunchKernelGGL(k_head_fwd, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias, y, M, K);
  return check_launch("k_head_fwd");
}
int tmjx_silu_bwd(const float *dy, const float *z, const float *bias, float *dz, long long rows, int N, void *stream) {
  if (!dy || !z || !bias || !dz) return fail(TMJX_EINVAL, "null argument");
  if (rows < 1 || N < 1) return fail(TMJX_EINVAL, "bad sizes");
  const long long total = rows * N;
  hipLaunchKernelGGL(k_silu_bwd_f32, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, dy, z, bias, dz, total, N);
  return check_launch("k_silu_bwd_f32");
}
}  // extern "C"
template <int MT>
static int launch_gemm_ln_bwd(const float *dY, int ldy, const float *W, int ldw, const float *bias, float *dz, int M, int N, int K, GemmLN ln, hipStream_t s) {
  constexpr int BM = 16 * MT;
  constexpr size_t lds = 2 * sizeof(float) * (size_t)((BM + 8) * GEMM_LDA + GEMM_BK * (256 + 4));
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_gemm_act<4, false, true, true, 2, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_gemm_act LN bwd): ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm_act<4, false, true, true, 2, MT>), dim3((M + BM - 1) / BM, 1), dim3(GemmCfg<4>::THREADS), lds, s, dY, ldy, W, ldw, bias, dz, N, M, N, K, ln);
  return check_launch("k_gemm_act(LN bwd)");
}
extern "C" {
int tmjx_gemm_nn_ln_bwd_ok(const float *dY, int ldy, const float *W, int ldw, int N) { return N == 256 && aligned16(dY, ldy) && aligned16(W, ldw); }
// (one partial row of 3 N column sums per workgroup of the launch: the row tile is gemm_mt's)
long long tmjx_gemm_nn_ln_bwd_partial_floats(int M, int N) { const int bm = 16 * gemm_mt(M, 1); return (long long)((M + bm - 1) / bm) * 3 * N; }
int tmjx_gemm_nn_ln_bwd(const float *dY, int ldy, const float *W, int ldw, const float *z, const float *bias, const float *gamma, const float *stats,
                        float *dz, float *partial, int M, int N, int K, void *stream) {
  if (!dY || !W || !z || !bias || !gamma || !stats || !dz || !partial) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || K < 1 || ldy < K || ldw < N) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  if (!tmjx_gemm_nn_ln_bwd_ok(dY, ldy, W, ldw, N)) return fail(TMJX_EINVAL, "tmjx_gemm_nn_ln_bwd: N must be 256 and the operands' rows 16-byte aligned");
  GemmLN ln{gamma, nullptr, nullptr, const_cast<float *>(stats), 0.f, z, partial};
  // (the kernel's contraction length is its "K" = the next layer's width; its "N" = this block's width = 256; z and dz are dense [M][256])
  if (gemm_mt(M, 1) == 2) return launch_gemm_ln_bwd<2>(dY, ldy, W, ldw, bias, dz, M, N, K, ln, (hipStream_t)stream);
  return launch_gemm_ln_bwd<5>(dY, ldy, W, ldw, bias, dz, M, N, K, ln, (hipStream_t)stream);
}
int tmjx_gemm_nn(const float *A, int lda, const float *W, int ldw, float *C, int ldc, int M, int N, int K, void *stream) {
  return gemm_act<false>(A, lda, W, ldw, nullptr, C, ldc, M, N, K, stream);
}
// rows of M per slab and number of slabs so that tiles x slabs is about the number of CUs (256); `max_slabs` > 0 caps the slab count (the
// grouped launch shares the chip between its problems: tmjx_gemm_dw_grouped)
static void dw_split(int M, int N, int K, int *rows_per_split, int *S, int *ld_slab, int max_slabs = 0) {
  const int tiles = ((N + DW_BT - 1) / DW_BT) * ((K + DW_BT - 1) / DW_BT);
  static const int target = getenv("TMJX_DW_WGS") ? atoi(getenv("TMJX_DW_WGS")) : 256;      // tuning knob
  int want = (target + tiles - 1) / tiles;      // one workgroup per CU: twice as many slabs (two per CU) ran the kernel no faster and doubled the reduction's traffic
  if (max_slabs > 0 && want > max_slabs) want = max_slabs;
  if (want < 1) want = 1;
  int rps = (((M + want - 1) / want) + DW_BM - 1) / DW_BM * DW_BM;
  if (rps < DW_BM) rps = DW_BM;
  *rows_per_split = rps;
  *S = (M + rps - 1) / rps;
  *ld_slab = ((K + DW_BT - 1) / DW_BT) * DW_BT + 4;
}
long long tmjx_gemm_dw_scratch_floats(int M, int N, int K) {
  if (M < 1 || N < 1 || K < 1) return 0;
  int rps, S, ld;
  dw_split(M, N, K, &rps, &S, &ld);
  return (long long)S * N * ld;
}
int tmjx_gemm_dw(const float *dY, int ldy, const float *X, int ldx, float *dW, float *db, float *scratch, int M, int N, int K, void *stream) {
  if (!dY || !X || !dW || !scratch) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1 || ldy < N || ldx < K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  int rps, S, ld;
  dw_split(M, N, K, &rps, &S, &ld);
  hipStream_t s = (hipStream_t)stream;
  const bool yv = aligned16(dY, ldy), xv = aligned16(X, ldx);
  int rc;
  if (yv && xv) rc = launch_gemm_dw<true, true>(dY, ldy, X, ldx, scratch, M, N, K, db ? 1 : 0, rps, S, ld, s);
  else if (yv) rc = launch_gemm_dw<true, false>(dY, ldy, X, ldx, scratch, M, N, K, db ? 1 : 0, rps, S, ld, s);
  else if (xv) rc = launch_gemm_dw<false, true>(dY, ldy, X, ldx, scratch, M, N, K, db ? 1 : 0, rps, S, ld, s);
  else rc = launch_gemm_dw<false, false>(dY, ldy, X, ldx, scratch, M, N, K, db ? 1 : 0, rps, S, ld, s);
  if (rc) return rc;
  const long long total = (long long)N * (K + (db ? 1 : 0));
  hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float *)scratch, dW, db, S, N, K, db ? 1 : 0, ld, K);
  return check_launch("k_gemm_dw");
}

int tmjx_gemm_dw_grouped(const tmjx_dw_problem_t *probs, int n, void *stream) { return tmjx_gemm_dw_grouped_wgs(probs, n, 0, stream); }
int tmjx_gemm_dw_grouped_wgs(const tmjx_dw_problem_t *probs, int n, int target_wgs, void *stream) {
  if (!probs) return fail(TMJX_EINVAL, "null argument");
  if (n < 1 || n > DW_GROUP_MAX) return fail(TMJX_EINVAL, "1 .. 16 problems per group");
  DwGroup G;
  G.n = n;
  int wg = 0, red = 0;
  // The problems of a group run side by side in ONE launch: together they should fill the chip about four times (TMJX_DW_GROUP_WGS = 1024 workgroups: two
  // 74 KB workgroups fit a CU), not once EACH — nine problems split for 256 workgroups apiece were 2 300 workgroups writing and re-reading
  // 64 slabs per weight matrix (160 MB per backward pass at 20 480 rows, 130 MB at 5 120).  Never more slabs than the problem's own split
  // (the caller sized the scratch by tmjx_gemm_dw_scratch_floats).
  static const int group_env = getenv("TMJX_DW_GROUP_WGS") ? atoi(getenv("TMJX_DW_GROUP_WGS")) : 0;
  const int group_target = target_wgs > 0 ? target_wgs : group_env;           // (a group that runs NEXT TO other kernels asks for fewer workgroups)
  int all_tiles = 0;
  for (int i = 0; i < n; i++) all_tiles += ((probs[i].N + DW_BT - 1) / DW_BT) * ((probs[i].K + DW_BT - 1) / DW_BT);
  if (all_tiles < 1) all_tiles = 1;
  // Round 6: the workgroup count has to land just UNDER a multiple of the 256 CUs.  The 2 x 256 nets' group is 42 tiles; per minibatch step at 20 480 /
  // 5 120 rows (cfg2 / cfg3, ms): 6 slabs = 252 workgroups 0.842 / 0.381, 12 = 504 0.846 / 0.391, 10 = 420 0.882 / 0.395, 8 = 336 0.942 / 0.405,
  // 25 = 1 050 (the old "about 1 024") 0.865 / 0.408 — a count like 336 puts a second workgroup on 80 of the CUs, which then run both at half speed while
  // the rest wait; 1 050 is two full rounds of two per CU plus 26 stragglers.  So: as many slabs as keep tiles x slabs within ONE workgroup per CU (fewest
  // slabs = least slab traffic and reduction work among the good counts).  Groups of more than 256 tiles (the rodent-mc-intention nets) are flat in
  // this knob (5.06 - 5.09 ms): about 1 024 workgroups as before.  TMJX_DW_GROUP_WGS / target_wgs: an explicit workgroup budget instead.
  const int max_slabs = group_target > 0 ? (group_target + all_tiles - 1) / all_tiles : (all_tiles <= 256 ? 256 / all_tiles : (1024 + all_tiles - 1) / all_tiles);
  for (int i = 0; i < n; i++) {
    const tmjx_dw_problem_t &q = probs[i];
    if (!q.dY || !q.X || !q.dW || !q.scratch) return fail(TMJX_EINVAL, "null pointer in a problem");
    if (q.M < 1 || q.N < 1 || q.K < 1 || q.ldy < q.N || q.ldx < q.K || q.lddw < q.K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions in a problem");
    if (!aligned16(q.dY, q.ldy) || !aligned16(q.X, q.ldx)) return fail(TMJX_EINVAL, "grouped weight gradients need 16-byte aligned rows of dY and X");
    DwProblem &P = G.p[i];
    P.dY = q.dY; P.X = q.X; P.dW = q.dW; P.db = q.db; P.slabs = q.scratch;
    P.ldy = q.ldy; P.ldx = q.ldx; P.lddw = q.lddw; P.M = q.M; P.N = q.N; P.K = q.K;
    dw_split(q.M, q.N, q.K, &P.rows_per_split, &P.S, &P.ld_slab, max_slabs);
    P.tiles_n = (q.N + DW_BT - 1) / DW_BT; P.tiles_k = (q.K + DW_BT - 1) / DW_BT;
    P.wg_begin = wg; wg += P.tiles_n * P.tiles_k * P.S;
    P.red_begin = red; red += (int)(((long long)q.N * (q.K + (q.db ? 1 : 0)) + 255) / 256);
  }
  constexpr size_t lds = 2 * sizeof(float) * 2 * DW_BM * DW_LD;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_gemm_dw_grouped, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_gemm_dw_grouped): ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gemm_dw_grouped, dim3(wg), dim3(512), lds, s, G);
  hipLaunchKernelGGL(k_dw_reduce_grouped, dim3(red), dim3(256), 0, s, G);
  return check_launch("k_gemm_dw_grouped");
}

int tmjx_rollout_store(const tmjx_rollout_store_t *q, void *stream) {
  if (!q) return fail(TMJX_EINVAL, "null argument");
  if (q->n < 1 || q->W < 0 || q->A < 0) return fail(TMJX_EINVAL, "bad sizes");
  if ((q->obs_dst0 || q->obs_dst1 || q->obs_dst2) && (!q->obs || q->W < 1)) return fail(TMJX_EINVAL, "observation destination without a source");
  if ((q->raw_dst && !q->raw) || (q->logp_dst && !q->logp) || (q->reward_dst && !q->reward) || (q->discount_dst && !q->done) || (q->trunc_dst && !q->trunc))
    return fail(TMJX_EINVAL, "destination without a source");
  RolloutStore s{q->obs, q->obs_dst0, q->obs_dst1, q->raw, q->raw_dst, q->logp, q->logp_dst, q->reward, q->reward_dst, q->done, q->discount_dst,
                 q->trunc, q->trunc_dst, q->n, q->W, q->A, q->obs_dst2};
  const int ny = (q->obs_dst0 || q->obs_dst1 || q->obs_dst2) ? (q->W + 15) / 16 : 0;
  hipLaunchKernelGGL(k_rollout_store, dim3((q->n + 63) / 64, ny + 1), dim3(64), 0, (hipStream_t)stream, s);
  return check_launch("k_rollout_store");
}

int tmjx_stats_scratch_floats(int W) { return STATS_SLABS * 2 * W; }

int tmjx_stats_sums(const float *src, const float *mean, float *sums, float *scratch, long long rows, int W, void *stream) {
  if (!src || !mean || !sums || !scratch) return fail(TMJX_EINVAL, "null argument");
  if (rows < 1 || W < 4 || (W & 3)) return fail(TMJX_EINVAL, "rows must be >= 1 and W a positive multiple of 4");
  if (((uintptr_t)src | (uintptr_t)mean | (uintptr_t)scratch) & 15) return fail(TMJX_EINVAL, "src, mean and scratch must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_stats_partial, dim3(((W >> 2) + 63) / 64, STATS_SLABS), dim3(256), 0, s, src, mean, scratch, rows, W);
  hipLaunchKernelGGL(k_colsum, dim3((2 * W + 31) / 32), dim3(256), 0, s, (const float *)scratch, sums, STATS_SLABS, 2 * W);
  return check_launch("k_stats_partial");
}

int tmjx_stats_apply(const float *sums, float n_added, float *count, float *mean, float *summed_variance, float *std, int W, float std_min,
                     float std_max, void *stream) {
  if (!sums || !count || !mean || !summed_variance || !std) return fail(TMJX_EINVAL, "null argument");
  if (W < 1 || !(n_added > 0.f)) return fail(TMJX_EINVAL, "W must be >= 1 and n_added > 0");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_stats_finalize, dim3((W + 255) / 256), dim3(256), 0, s, sums, n_added, count, mean, summed_variance, std, W, std_min, std_max);
  hipLaunchKernelGGL(k_stats_count, dim3(1), dim3(1), 0, s, count, n_added);
  return check_launch("k_stats_finalize");
}

int tmjx_debug_rows(const tmjx_model *m, const char *name, int *row0, int *count) {
  if (!m || !name || !row0 || !count) return fail(TMJX_EINVAL, "null argument");
  // "k2_kernel": which physics kernel this handle launches — *row0 = 1 the compile-time (rodent chain) specialisation, 0 the generic one;
  // *count = its dynamic LDS bytes per env (tests pin both: a model that silently fell back to the generic kernel ran 2.4 x slower)
  if (!strcmp(name, "k2_kernel")) {
    *row0 = m->rodent ? 1 : 0;
    *count = (int)((m->rodent ? m->h.lds_floats : tmjx_host::make_wave_layout(m->h, false).lds_floats) * sizeof(float));
    return 0;
  }
  for (const auto &e : tmjx_host::debug_rows(m->h))
    if (!strcmp(e.name, name)) { *row0 = e.row0; *count = e.count; return e.in_state ? 1 : 0; }
  return fail(TMJX_EINVAL, std::string("unknown debug array: ") + name);
}

}  // extern "C"
