// csrc/tm_common.h — what every per-env kernel body of the hot path shares: the env-minor buffer accessors, small vector / quaternion /
// spatial-algebra helpers, and the two pieces of MJX arithmetic used verbatim by both physics formulations (plane-sphere contact,
// contact frame, solref / solimp -> stiffness, damping, impedance).  Paths of the reference call sites: see wave_physics.h.
#pragma once
#include <math.h>
#include <stddef.h>

#include "dmodel.h"

#ifndef TM_DEV
#define TM_DEV __device__ __forceinline__
#endif

#define TM_MINVAL 1e-15f
#define TM_MINIMP 0.0001f
#define TM_MAXIMP 0.9999f

struct EnvRef {
  float *st;   // float state buffer  [rows][n]
  float *ws;   // workspace           [rows][n]
  int n;       // number of envs = row stride
  int e;       // this lane's env
};
#define ST(off, i) r.st[(size_t)((off) + (i)) * (size_t)r.n + (size_t)r.e]
#define WS(off, i) r.ws[(size_t)((off) + (i)) * (size_t)r.n + (size_t)r.e]

// ------------------------------------------------------------------------------------------ small math
TM_DEV float tm_dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
TM_DEV void tm_cross(float *o, const float *a, const float *b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
TM_DEV void tm_rotate(float *o, const float *v, const float *q) {
  float s = q[0], uv = q[1] * v[0] + q[2] * v[1] + q[3] * v[2], uu = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float c[3];
  tm_cross(c, q + 1, v);
  float w = s * s - uu;
  o[0] = 2.f * (uv * q[1]) + w * v[0] + 2.f * s * c[0];
  o[1] = 2.f * (uv * q[2]) + w * v[1] + 2.f * s * c[1];
  o[2] = 2.f * (uv * q[3]) + w * v[2] + 2.f * s * c[2];
}
TM_DEV void tm_quat_mul(float *o, const float *a, const float *b) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
TM_DEV void tm_quat_to_mat(float *m, const float *q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2.f * (x * y - w * z); m[2] = 2.f * (x * z + w * y);
  m[3] = 2.f * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2.f * (y * z - w * x);
  m[6] = 2.f * (x * z - w * y); m[7] = 2.f * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
TM_DEV float tm_normalize3(float *v) {
  float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  float d = n + (n == 0.f ? 1e-6f : 0.f);
  v[0] /= d; v[1] /= d; v[2] /= d;
  return n;
}
TM_DEV void tm_normalize4(float *v) {
  float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  float d = n + (n == 0.f ? 1e-6f : 0.f);
  v[0] /= d; v[1] /= d; v[2] /= d; v[3] /= d;
}
// spatial: motion vector = [ang, lin]; cinert = [xx,yy,zz,xy,xz,yz, m*off(3), m]
TM_DEV void tm_inert_mul(float *o, const float *I, const float *v) {
  float c1[3], c2[3];
  tm_cross(c1, I + 6, v + 3);
  tm_cross(c2, I + 6, v);
  o[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] + c1[0];
  o[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + c1[1];
  o[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] + c1[2];
  o[3] = I[9] * v[3] - c2[0]; o[4] = I[9] * v[4] - c2[1]; o[5] = I[9] * v[5] - c2[2];
}
TM_DEV void tm_motion_cross(float *o, const float *u, const float *v) {
  float a[3], b[3], c[3];
  tm_cross(a, u, v); tm_cross(b, u, v + 3); tm_cross(c, u + 3, v);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
TM_DEV void tm_motion_cross_force(float *o, const float *v, const float *f) {
  float a[3], b[3], c[3];
  tm_cross(a, v, f); tm_cross(b, v + 3, f + 3); tm_cross(c, v, f + 3);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
#define TM_LD(dst, MAC, off, base, cnt) for (int k_ = 0; k_ < (cnt); k_++) (dst)[k_] = MAC(off, (base) + k_)
#define TM_SV(MAC, off, base, src, cnt) for (int k_ = 0; k_ < (cnt); k_++) MAC(off, (base) + k_) = (src)[k_]

// ------------------------------------------------------------------------------------------ collision / constraint helpers
TM_DEV void tm_plane_sphere(const float *n, const float *pp, const float *sp, float rad, float &dist, float *pos) {
  float df[3] = {sp[0] - pp[0], sp[1] - pp[1], sp[2] - pp[2]};
  dist = tm_dot3(df, n) - rad;
  for (int k = 0; k < 3; k++) pos[k] = sp[k] - n[k] * (rad + 0.5f * dist);
}
TM_DEV void tm_make_frame(const float *a_in, float *fr) {
  float a[3] = {a_in[0], a_in[1], a_in[2]}, b[3], c[3];
  tm_normalize3(a);
  bool yy = (-0.5f < a[1]) && (a[1] < 0.5f);
  b[0] = 0.f; b[1] = yy ? 1.f : 0.f; b[2] = yy ? 0.f : 1.f;
  float ab = tm_dot3(a, b);
  for (int k = 0; k < 3; k++) b[k] -= a[k] * ab;
  tm_normalize3(b);
  tm_cross(c, a, b);
  for (int k = 0; k < 3; k++) { fr[k] = a[k]; fr[3 + k] = b[k]; fr[6 + k] = c[k]; }
}
TM_DEV void tm_kbi(float timestep, const float *solref, const float *solimp, float pos, float &k, float &b, float &imp) {
  float timeconst = fmaxf(solref[0], 2.f * timestep), dampratio = solref[1];
  float dmin = fminf(fmaxf(solimp[0], TM_MINIMP), TM_MAXIMP), dmax = fminf(fmaxf(solimp[1], TM_MINIMP), TM_MAXIMP);
  float width = fmaxf(TM_MINVAL, solimp[2]), mid = fminf(fmaxf(solimp[3], TM_MINIMP), TM_MAXIMP), power = fmaxf(1.f, solimp[4]);
  k = 1.f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  b = 2.f / (dmax * timeconst);
  if (solref[0] <= 0.f) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0.f) b = -solref[1] / dmax;
  float x = fabsf(pos) / width;
  // power == 2 (MuJoCo's default solimp, every joint and geom of the rodent): squares instead of four powf calls (~150 VALU
  // instructions each on the GPU; powf(x, 2) is x * x up to the last bit anyway)
  float ia, ib;
  if (power == 2.f) {
    ia = (1.f / mid) * (x * x);
    ib = 1.f - (1.f / (1.f - mid)) * ((1.f - x) * (1.f - x));
  } else {
    ia = (1.f / powf(mid, power - 1.f)) * powf(x, power);
    ib = 1.f - (1.f / powf(1.f - mid, power - 1.f)) * powf(1.f - x, power);
  }
  float y = x < mid ? ia : ib;
  float im = dmin + y * (dmax - dmin);
  im = fminf(fmaxf(im, dmin), dmax);
  if (x > 1.f) im = dmax;
  imp = im;
}
TM_DEV void tm_kbi(const DModel &m, const float *solref, const float *solimp, float pos, float &k, float &b, float &imp) {
  tm_kbi(m.timestep, solref, solimp, pos, k, b, imp);
}

// Kernels of an env group's SERIAL phase (K3, roll-out store, acting policy: between two physics launches of the group) run next to the other groups'
// physics waves, which are older and therefore win the SIMD's issue arbitration at equal priority (MI355X_MICROARCH.md "priority, then age"): these
// short, latency-critical waves raise their own priority for their whole life (round 5; -DTMJX_NO_ACT_PRIO builds without it for A/B runs).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TMJX_NO_ACT_PRIO)
#define TM_PRIO_ACTING() __builtin_amdgcn_s_setprio(3)
#else
#define TM_PRIO_ACTING() do { } while (0)
#endif
