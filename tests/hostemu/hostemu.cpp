#include <cstdlib>
// tests/hostemu/hostemu.cpp — TEST-ONLY host emulation of the HIP kernel bodies.
//
// Compiles the very same per-env device functions (tests/lane/physics_core.h, csrc/env_core.h) with g++ and
// runs them lane-by-lane on host memory, so that the kernel source can be unit-tested against the
// oracle in the GPU-less build container.  It is NOT a fallback: nothing in track_mjx_amd/ loads
// this library and the product path fails loudly without libtmjx_hip.so.
#define TM_HOST_EMU 1
#define TM_DEV static inline
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../track_mjx_amd/csrc/env_core.h"
#include "../lane/physics_core.h"
#include "../../track_mjx_amd/csrc/model_host.h"
#include "../../track_mjx_amd/csrc/wave_physics.h"
#include <algorithm>
#include <vector>

struct EmuModel { DModel h; float *clips[5]; };
static std::string g_err;

extern "C" {
const char *emu_last_error() { return g_err.c_str(); }
EmuModel *emu_model_create(const void *blob, size_t n) {
  EmuModel *m = new EmuModel();
  memset(m->clips, 0, sizeof(m->clips));
  if (!tmjx_host::build_dmodel(blob, n, m->h, g_err)) { delete m; return nullptr; }
  return m;
}
void emu_model_destroy(EmuModel *m) { if (!m) return; for (int i = 0; i < 5; i++) free(m->clips[i]); delete m; }
void emu_layout(const EmuModel *m, int *out) {
  out[0] = m->h.s_rows; out[1] = m->h.i_rows; out[2] = m->h.w_rows; out[3] = m->h.obs_size; out[4] = m->h.nphys;
}
int emu_rows(const EmuModel *m, const char *name, int *row0, int *count) {
  for (const auto &e : tmjx_host::debug_rows(m->h)) if (!strcmp(e.name, name)) { *row0 = e.row0; *count = e.count; return e.in_state ? 1 : 0; }
  return -1;
}
void emu_clips(EmuModel *m, const float *p, const float *q, const float *j, const float *b, const float *a, int nc, int nf) {
  size_t cf = (size_t)nc * nf, w[5] = {3, 4, (size_t)(m->h.nq - 7), (size_t)(m->h.nbody - 1) * 3, 3};
  const float *src[5] = {p, q, j, b, a};
  for (int i = 0; i < 5; i++) { free(m->clips[i]); m->clips[i] = (float *)malloc(cf * w[i] * 4); memcpy(m->clips[i], src[i], cf * w[i] * 4); }
  m->h.clip_pos = m->clips[0]; m->h.clip_quat = m->clips[1]; m->h.clip_joints = m->clips[2]; m->h.clip_bodypos = m->clips[3]; m->h.clip_angvel = m->clips[4];
  m->h.n_clips = nc; m->h.n_frames_clip = nf;
}
void emu_reset(EmuModel *mm, float *st, int *is, const int *clip, const int *start, const float *qn, const float *vn, float *obs, float *ws, int n) {
  const DModel &m = mm->h;
  for (int e = 0; e < n; e++) { EnvRef r{st, ws, n, e}; tm_reset_pre(m, r, is, clip[e], start[e], qn, vn); tm_forward(m, r); tm_reset_post(m, r, is, obs); }
}
void emu_step(EmuModel *mm, float *st, int *is, const float *action, float *obs, float *rew, float *done, float *trunc, float *metrics, float *ws, int n) {
  const DModel &m = mm->h;
  for (int e = 0; e < n; e++) {
    EnvRef r{st, ws, n, e};
    tm_step_prologue(m, r);
    for (int a = 0; a < m.nu; a++) WS(m.w_ctrl, a) = action[(size_t)a * n + e];
    for (int f = 0; f < m.n_frames; f++) { tm_forward(m, r); tm_euler(m, r); }
    tm_step_post(m, r, is, action, obs, rew, done, trunc, metrics);
  }
}
// tmjx_step with action_repeat = R (tmjx_hip.hip): per repeat the physics, then K3 told which repeat it is
void emu_step_repeat(EmuModel *mm, float *st, int *is, const float *action, float *obs, float *rew, float *done, float *trunc, float *metrics, float *ws, int n, int R) {
  const DModel &m = mm->h;
  for (int e = 0; e < n; e++) {
    EnvRef r{st, ws, n, e};
    for (int k = 0; k < R; k++) {
      if (k == 0) tm_step_prologue(m, r);
      for (int a = 0; a < m.nu; a++) WS(m.w_ctrl, a) = action[(size_t)a * n + e];
      for (int f = 0; f < m.n_frames; f++) { tm_forward(m, r); tm_euler(m, r); }
      tm_step_post(m, r, is, action, obs, rew, done, trunc, metrics, nullptr, false, nullptr, TM_REP(R, k == 0, k == R - 1));
    }
  }
}
void emu_physics(EmuModel *mm, float *st, const float *action, int nsub, int do_euler, float *ws, int n) {
  const DModel &m = mm->h;
  for (int e = 0; e < n; e++) {
    EnvRef r{st, ws, n, e};
    for (int a = 0; a < m.nu; a++) WS(m.w_ctrl, a) = action ? action[(size_t)a * n + e] : 0.f;
    for (int f = 0; f < nsub; f++) { tm_forward(m, r); if (do_euler) tm_euler(m, r); }
  }
}
// wave-per-env kernel body (csrc/wave_physics.h): one emulated 64-lane wavefront + an LDS image per env
void emu_physics_wave(EmuModel *mm, float *st, const float *action, int nsub, int do_euler, float *ws_dump, int n) {
  std::vector<float> lds(std::max(tmjx_host::make_wave_layout(mm->h, true).lds_floats, tmjx_host::make_wave_layout(mm->h, false).lds_floats) + 64);
  // TMJX_EMU_POISON=nan (or a number): what a wave finds in LDS, in its registers and in its scratch when it starts is whatever the previous wave on
  // that CU left there — NaNs, if that env had blown up.  The kernel body must not let such words reach a result (0 x NaN is NaN: a masked product
  // needs the select on BOTH operands or on the product); the default image is zeros, under which such a read goes unnoticed.
  const char *poison_s = getenv("TMJX_EMU_POISON");
  const float poison = poison_s ? (!strcmp(poison_s, "nan") ? NAN : (float)atof(poison_s)) : 0.f;
  for (int e = 0; e < n; e++) {
    for (auto &v : lds) v = poison;
    WCtx c{&mm->h, lds.data(), st, n, e, 0, nullptr, 0ull, ws_dump};
    if (poison_s) {
      for (int l = 0; l < TMW_NL; l++) {
        c.qfs0[l] = c.qfs1[l] = c.dg0[l] = c.dg1[l] = c.wp0[l] = c.wp1[l] = poison;
#ifdef TMW_EMU_HAS_QA
        c.qa0[l] = c.qa1[l] = c.ma0[l] = c.ma1[l] = poison;
#endif
      }
    }
    std::vector<float> spill(mm->h.nnz + mm->h.nv + 64, poison);
    c.mspill = spill.data() + 64;
    c.action = action;
    const WLayout K = tmjx_host::make_wave_layout(mm->h, !getenv("TMJX_EMU_GENERIC"));
    float time = tmw_load_state(c, K, action);
    for (int f = 0; f < nsub; f++) { tmw_forward(c, K, f == nsub - 1); if (do_euler) time = tmw_euler(c, K, time); }
    if (ws_dump) tmw_dump(c, K, ws_dump);
    tmw_store_state(c, K, time);
  }
}
int emu_lds_floats(const EmuModel *m) { return m->h.lds_floats; }
void emu_post(EmuModel *mm, float *st, int *is, const float *action, float *obs, float *rew, float *done, float *trunc, float *metrics, int n) {
  const DModel &m = mm->h;
  for (int e = 0; e < n; e++) { EnvRef r{st, nullptr, n, e}; tm_step_prologue(m, r); tm_step_post(m, r, is, action, obs, rew, done, trunc, metrics); }
}
}
