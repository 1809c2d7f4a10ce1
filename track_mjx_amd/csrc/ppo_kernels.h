// csrc/ppo_kernels.h — the PPO loss head as four kernels: everything of compute_ppo_loss
// (track_mjx/agent/mlp_ppo/losses.py:103-245) between the network outputs and the scalar loss, forward AND the
// gradients w.r.t. the network outputs (the loss is a scalar, so the backward pass only scales what is stored here).
// Replaces ~170 tiny element-wise / reduction launches per minibatch step.  Deterministic: block partials + a final
// single-block reduction, no atomics.
//
//   logits  [T][B][2A]  loc | raw_scale of the NormalTanh policy (brax NormalTanhDistribution, min_std 0.001)
//   fc2     [T][B][2Z]  latent_mean | latent_logvar of the intention encoder
//   scratch [4 T B + 4 nblk + 16] floats, nblk = ceil(8 T B / 256)
//   out     [8]: total, policy_loss, v_loss, entropy_loss, kl_latent_loss, adv_mean, adv_std, entropy
#pragma once
#include "silu_math.h"
#include <hip/hip_runtime.h>
#include <math.h>

struct PpoCfg {
  int T, B, A, Z;
  float reward_scaling, discounting, gae_lambda, clip_eps, entropy_cost, kl_weight;
  int normalize_advantage;
  int accumulate;
};

#define PPO_BLOCK 256
#define PPO_ALPHA 0.95f
__device__ __forceinline__ float ppo_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float ppo_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float ppo_fldj(float x) { return 2.f * (0.69314718055994531f - x - ppo_softplus(-2.f * x)); }

// block-wide sum of NV values per thread; result valid in thread 0
template <int NV>
__device__ __forceinline__ void ppo_block_sum(float *v, float *lds) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    float x = v[k];
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    if (lane == 0) lds[k * 16 + w] = x;
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) { float s = 0.f; for (int i = 0; i < nw; i++) s += lds[k * 16 + i]; v[k] = s; }
  }
  __syncthreads();
}

// A: per (t, b): log-prob of the stored action, entropy sample, latent-KL sums.  PPO_G = 8 lanes share one (t, b): lane s takes
// the action / latent dims s, s + 8, .. (8 consecutive floats per load instead of one float per 304-byte row per lane)
#define PPO_G 8
__device__ __forceinline__ float ppo_group_sum(float x) { x += __shfl_xor(x, 4); x += __shfl_xor(x, 2); x += __shfl_xor(x, 1); return x; }
__global__ __launch_bounds__(PPO_BLOCK) void k_ppo_a(PpoCfg c, const float *__restrict__ logits, const float *__restrict__ raw_action,
                                                     const float *__restrict__ noise, const float *__restrict__ fc2, float *scratch, int nblk) {
  __shared__ float lds[64];
  const int N = c.T * c.B, gid = blockIdx.x * PPO_BLOCK + threadIdx.x, i = gid / PPO_G, sub = gid % PPO_G;
  float acc[3] = {0.f, 0.f, 0.f};   // entropy, kl0 inner sum, klt inner sum
  if (i < N) {
    const int t = i / c.B;
    const float *lg = logits + (size_t)i * 2 * c.A, *xa = raw_action + (size_t)i * c.A, *nz = noise + (size_t)i * c.A;
    float logp = 0.f, ent = 0.f;
    for (int a = sub; a < c.A; a += PPO_G) {
      float loc = lg[a], scale = ppo_softplus(lg[c.A + a]) + 0.001f, x = xa[a], d = (x - loc) / scale;
      logp += -0.5f * d * d - logf(scale) - 0.91893853320467274f - ppo_fldj(x);
      float xs = loc + scale * nz[a];
      ent += 0.5f + 0.91893853320467274f + logf(scale) + ppo_fldj(xs);
    }
    logp = ppo_group_sum(logp);
    if (sub == 0) scratch[i] = logp;
    acc[0] = ent;
    const float *f = fc2 + (size_t)i * 2 * c.Z;
    const float pv = 1.f - PPO_ALPHA * PPO_ALPHA;
    float s = 0.f;
    if (t == 0) {
      for (int z = sub; z < c.Z; z += PPO_G) { float m = f[z], lv = f[c.Z + z]; s += 1.f + lv - m * m - expf(lv); }
      acc[1] = s;
    } else {
      const float *fp = f - (size_t)c.B * 2 * c.Z;
      for (int z = sub; z < c.Z; z += PPO_G) { float m = f[z], lv = f[c.Z + z], e = PPO_ALPHA * fp[z] - m; s += expf(lv) / pv + e * e / pv - 1.f + logf(pv) - lv; }
      acc[2] = s;
    }
  }
  ppo_block_sum<3>(acc, lds);
  if (threadIdx.x == 0) { float *p = scratch + (size_t)4 * N + (size_t)blockIdx.x * 4; p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2]; }
}

// B: one block: GAE per column (losses.py:39-100), advantage statistics, value loss, reduction of A's partials
__global__ __launch_bounds__(1024) void k_ppo_b(PpoCfg c, const float *__restrict__ baseline, const float *__restrict__ bootstrap,
                                                const float *__restrict__ reward, const float *__restrict__ discount,
                                                const float *__restrict__ truncation, float *scratch, int nblk) {
  __shared__ float lds[96];
  const int N = c.T * c.B;
  float *vs = scratch + N, *adv = scratch + 2 * (size_t)N, *part = scratch + 4 * (size_t)N, *scal = part + (size_t)4 * nblk;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // sum adv, sum adv^2 (shifted later), sum v_err^2, ent, kl0, klt
  constexpr int TMAX = 24;     // unroll_length <= TMAX: the whole column lives in registers (all loads of a column in flight together;
                               // the two scans below were chains of 2 T memory latencies before)
  for (int b = threadIdx.x; b < c.B; b += blockDim.x) {
    float bv = bootstrap[b];
    if (c.T <= TMAX) {
      float rw[TMAX], te[TMAX], tm[TMAX], vv[TMAX], vsr[TMAX];
#pragma unroll
      for (int t = 0; t < TMAX; t++) {
        size_t i = (size_t)(t < c.T ? t : 0) * c.B + b;
        float tr = truncation[i];
        rw[t] = reward[i] * c.reward_scaling; te[t] = (1.f - discount[i]) * (1.f - tr); tm[t] = 1.f - tr; vv[t] = baseline[i];
      }
      float a = 0.f, vnext = bv;
#pragma unroll
      for (int t = TMAX - 1; t >= 0; t--) {
        if (t < c.T) {
          float delta = (rw[t] + c.discounting * (1.f - te[t]) * vnext - vv[t]) * tm[t];
          a = delta + c.discounting * (1.f - te[t]) * tm[t] * c.gae_lambda * a;
          vsr[t] = a + vv[t];
          vs[(size_t)t * c.B + b] = vsr[t];
          vnext = vv[t];
        }
      }
      float vsn = bv;
#pragma unroll
      for (int t = TMAX - 1; t >= 0; t--) {
        if (t < c.T) {
          float ad = (rw[t] + c.discounting * (1.f - te[t]) * vsn - vv[t]) * tm[t];
          adv[(size_t)t * c.B + b] = ad;
          acc[0] += ad;
          float ve = vsr[t] - vv[t];
          acc[2] += ve * ve;
          vsn = vsr[t];
        }
      }
      continue;
    }
    float a = 0.f, vnext = bv;
    for (int t = c.T - 1; t >= 0; t--) {
      size_t i = (size_t)t * c.B + b;
      float tr = truncation[i], te = (1.f - discount[i]) * (1.f - tr), tm = 1.f - tr, v = baseline[i];
      float delta = (reward[i] * c.reward_scaling + c.discounting * (1.f - te) * vnext - v) * tm;
      a = delta + c.discounting * (1.f - te) * tm * c.gae_lambda * a;
      vs[i] = a + v;
      vnext = v;
    }
    float vsn = bv;
    for (int t = c.T - 1; t >= 0; t--) {
      size_t i = (size_t)t * c.B + b;
      float tr = truncation[i], te = (1.f - discount[i]) * (1.f - tr), tm = 1.f - tr, cur = vs[i], v = baseline[i];
      float ad = (reward[i] * c.reward_scaling + c.discounting * (1.f - te) * vsn - v) * tm;
      adv[i] = ad;
      acc[0] += ad;
      float ve = cur - v;
      acc[2] += ve * ve;
      vsn = cur;
    }
  }
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) { acc[3] += part[4 * k]; acc[4] += part[4 * k + 1]; acc[5] += part[4 * k + 2]; }
  ppo_block_sum<6>(acc, lds);
  __shared__ float mean_s;
  if (threadIdx.x == 0) mean_s = acc[0] / (float)N;
  __syncthreads();
  float mean = mean_s, ss[1] = {0.f};
  for (int b = threadIdx.x; b < c.B; b += blockDim.x)
    for (int t = 0; t < c.T; t++) { float d = adv[(size_t)t * c.B + b] - mean; ss[0] += d * d; }
  ppo_block_sum<1>(ss, lds);
  if (threadIdx.x == 0) {
    const float Nf = (float)N;
    scal[0] = mean;                                   // advantage mean
    scal[1] = sqrtf(ss[0] / Nf);                      // population std
    scal[2] = acc[2] / Nf * 0.25f;                    // v_loss = mean(err^2) * 0.5 * 0.5
    scal[3] = acc[3] / Nf;                            // entropy
    float kl0 = -0.5f * acc[4] / (float)(c.B * c.Z);
    float kl = kl0;
    if (c.T > 1) { float klt = 0.5f * acc[5] / (float)((c.T - 1) * c.B * c.Z); kl = (kl0 + klt * (float)(c.T - 1)) / (float)c.T; }
    scal[4] = c.kl_weight * kl;
  }
}

// B, many blocks (the single 1024-thread block above took 31 us of every minibatch step: one CU, two block-wide reductions and a second pass over the
// advantages in global memory): one 64-thread block per 64 columns.  Each block leaves a RECORD {n, mean, M2, sum v_err^2, ent, kl0, klt} of its
// columns (mean / M2 of the advantages two-pass within the block, from registers) and of its slice of A's partial sums; C and D combine the
// records (Chan's update for the variance).  T <= PPO_TMAX (24) only; longer unrolls take k_ppo_b.
#define PPO_TMAX 24
#define PPO_REC 8
__device__ __forceinline__ float ppo_wave_sum(float x) { for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off); return x; }
__global__ __launch_bounds__(64) void k_ppo_b2(PpoCfg c, const float *__restrict__ baseline, const float *__restrict__ bootstrap,
                                               const float *__restrict__ reward, const float *__restrict__ discount,
                                               const float *__restrict__ truncation, float *scratch, int nblk, float *rec) {
  const int N = c.T * c.B, b = blockIdx.x * 64 + threadIdx.x;
  float *vs = scratch + N, *adv = scratch + 2 * (size_t)N;
  const float *part = scratch + 4 * (size_t)N;
  float ad[PPO_TMAX], s_ad = 0.f, s_ve = 0.f;
#pragma unroll
  for (int t = 0; t < PPO_TMAX; t++) ad[t] = 0.f;
  const bool on = b < c.B;
  if (on) {
    const float bv = bootstrap[b];
    float rw[PPO_TMAX], te[PPO_TMAX], tm[PPO_TMAX], vv[PPO_TMAX], vsr[PPO_TMAX];
#pragma unroll
    for (int t = 0; t < PPO_TMAX; t++) {
      size_t i = (size_t)(t < c.T ? t : 0) * c.B + b;
      float tr = truncation[i];
      rw[t] = reward[i] * c.reward_scaling; te[t] = (1.f - discount[i]) * (1.f - tr); tm[t] = 1.f - tr; vv[t] = baseline[i];
    }
    float a = 0.f, vnext = bv;
#pragma unroll
    for (int t = PPO_TMAX - 1; t >= 0; t--) {
      if (t < c.T) {
        float delta = (rw[t] + c.discounting * (1.f - te[t]) * vnext - vv[t]) * tm[t];
        a = delta + c.discounting * (1.f - te[t]) * tm[t] * c.gae_lambda * a;
        vsr[t] = a + vv[t];
        vs[(size_t)t * c.B + b] = vsr[t];
        vnext = vv[t];
      }
    }
    float vsn = bv;
#pragma unroll
    for (int t = PPO_TMAX - 1; t >= 0; t--) {
      if (t < c.T) {
        ad[t] = (rw[t] + c.discounting * (1.f - te[t]) * vsn - vv[t]) * tm[t];
        adv[(size_t)t * c.B + b] = ad[t];
        s_ad += ad[t];
        float ve = vsr[t] - vv[t];
        s_ve += ve * ve;
        vsn = vsr[t];
      }
    }
  }
  const int cols = min(64, c.B - (int)blockIdx.x * 64);
  const float n_b = (float)(cols * c.T);
  const float mean_b = ppo_wave_sum(s_ad) / n_b;
  float m2 = 0.f;
  if (on) {
#pragma unroll
    for (int t = 0; t < PPO_TMAX; t++) if (t < c.T) { float d = ad[t] - mean_b; m2 += d * d; }
  }
  m2 = ppo_wave_sum(m2);
  s_ve = ppo_wave_sum(s_ve);
  // this block's slice of A's per-block partial sums (entropy, kl0, klt)
  const int k0 = (int)((long long)nblk * blockIdx.x / gridDim.x), k1 = (int)((long long)nblk * (blockIdx.x + 1) / gridDim.x);
  float e0 = 0.f, e1 = 0.f, e2 = 0.f;
  for (int k = k0 + threadIdx.x; k < k1; k += 64) { e0 += part[4 * k]; e1 += part[4 * k + 1]; e2 += part[4 * k + 2]; }
  e0 = ppo_wave_sum(e0); e1 = ppo_wave_sum(e1); e2 = ppo_wave_sum(e2);
  if (threadIdx.x == 0) {
    float *r = rec + (size_t)PPO_REC * blockIdx.x;
    r[0] = n_b; r[1] = mean_b; r[2] = m2; r[3] = s_ve; r[4] = e0; r[5] = e1; r[6] = e2; r[7] = 0.f;
  }
}
// the records of k_ppo_b2 -> the five scalars k_ppo_b leaves in scal[] (one thread)
__device__ __forceinline__ void ppo_combine_records(const PpoCfg &c, const float *rec, int nrec, float *scal) {
  float n = 0.f, sm = 0.f, ve = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f;
  for (int k = 0; k < nrec; k++) { const float *r = rec + PPO_REC * k; n += r[0]; sm += r[0] * r[1]; ve += r[3]; e0 += r[4]; e1 += r[5]; e2 += r[6]; }
  const float mean = sm / n;
  float m2 = 0.f;
  for (int k = 0; k < nrec; k++) { const float *r = rec + PPO_REC * k; const float d = r[1] - mean; m2 += r[2] + r[0] * d * d; }
  scal[0] = mean;
  scal[1] = sqrtf(m2 / n);
  scal[2] = ve / n * 0.25f;
  scal[3] = e0 / n;
  float kl0 = -0.5f * e1 / (float)(c.B * c.Z), kl = kl0;
  if (c.T > 1) { float klt = 0.5f * e2 / (float)((c.T - 1) * c.B * c.Z); kl = (kl0 + klt * (float)(c.T - 1)) / (float)c.T; }
  scal[4] = c.kl_weight * kl;
}

// C: per (t, b) (PPO_G lanes each): surrogate loss term and every gradient w.r.t. logits, baseline, fc2
__global__ __launch_bounds__(PPO_BLOCK) void k_ppo_c(PpoCfg c, const float *__restrict__ logits, const float *__restrict__ raw_action,
                                                     const float *__restrict__ behaviour_logp, const float *__restrict__ noise,
                                                     const float *__restrict__ baseline, const float *__restrict__ fc2, float *dlogits,
                                                     float *dbaseline, float *dfc2, float *scratch, int nblk, const float *rec = nullptr, int nrec = 0) {
  __shared__ float lds[32];
  __shared__ float scal_s[8];
  const int N = c.T * c.B, gid = blockIdx.x * PPO_BLOCK + threadIdx.x, i = gid / PPO_G, sub = gid % PPO_G;
  const float *vs = scratch + N, *adv = scratch + 2 * (size_t)N, *scal = scratch + 4 * (size_t)N + (size_t)4 * nblk;
  if (rec) {       // k_ppo_b2's records: every block combines them itself (a few dozen words)
    if (threadIdx.x == 0) ppo_combine_records(c, rec, nrec, scal_s);
    __syncthreads();
    scal = scal_s;
  }
  float acc[1] = {0.f};
  if (i < N) {
    const float Nf = (float)N;
    float ad = adv[i];
    if (c.normalize_advantage) ad = (ad - scal[0]) / (scal[1] + 1e-8f);
    float rho = expf(scratch[i] - behaviour_logp[i]), lo = 1.f - c.clip_eps, hi = 1.f + c.clip_eps;
    float rc = fminf(fmaxf(rho, lo), hi), s1 = rho * ad, s2 = rc * ad;
    if (sub == 0) acc[0] = fminf(s1, s2);
    // d(-mean min(s1, s2)) / d logp: s1 taken (ties included: both sides then have the same derivative) or unclipped s2
    float glp = (s1 <= s2 || (rho >= lo && rho <= hi)) ? -ad * rho / Nf : 0.f;
    const float ce = -c.entropy_cost / Nf;   // entropy_loss = -entropy_cost * mean(ent)
    const float *lg = logits + (size_t)i * 2 * c.A, *xa = raw_action + (size_t)i * c.A, *nz = noise + (size_t)i * c.A;
    float *dl = dlogits + (size_t)i * 2 * c.A;
    for (int a = sub; a < c.A; a += PPO_G) {
      float loc = lg[a], raw = lg[c.A + a], scale = ppo_softplus(raw) + 0.001f, sig = ppo_sigmoid(raw), x = xa[a], d = x - loc;
      float inv = 1.f / scale, dlp_loc = d * inv * inv, dlp_scale = d * d * inv * inv * inv - inv;
      float n = nz[a], th = tanhf(loc + scale * n), de_loc = -2.f * th, de_scale = inv - 2.f * th * n;
      dl[a] = glp * dlp_loc + ce * de_loc;
      dl[c.A + a] = (glp * dlp_scale + ce * de_scale) * sig;
    }
    if (sub == 0) dbaseline[i] = -0.5f * (vs[i] - baseline[i]) / Nf;
    // latent KL
    const int t = i / c.B;
    const float pv = 1.f - PPO_ALPHA * PPO_ALPHA, kw = c.kl_weight;
    const float c0 = kw / (float)c.T / (float)(c.B * c.Z), c1 = c.T > 1 ? kw * (float)(c.T - 1) / (float)c.T / (float)((c.T - 1) * c.B * c.Z) : 0.f;
    const float *f = fc2 + (size_t)i * 2 * c.Z, *fp = f - (size_t)c.B * 2 * c.Z, *fn = f + (size_t)c.B * 2 * c.Z;
    float *df = dfc2 + (size_t)i * 2 * c.Z;
    for (int z = sub; z < c.Z; z += PPO_G) {
      float m = f[z], lv = f[c.Z + z], dm, dlv;
      if (t == 0) { dm = c0 * m; dlv = -0.5f * c0 * (1.f - expf(lv)); }
      else { float e = PPO_ALPHA * fp[z] - m; dm = -c1 * e / pv; dlv = 0.5f * c1 * (expf(lv) / pv - 1.f); }
      if (t + 1 < c.T) { float e2 = PPO_ALPHA * m - fn[z]; dm += c1 * PPO_ALPHA * e2 / pv; }
      df[z] = dm; df[c.Z + z] = dlv;
    }
  }
  ppo_block_sum<1>(acc, lds);
  if (threadIdx.x == 0) scratch[(size_t)4 * N + (size_t)blockIdx.x * 4 + 3] = acc[0];
}

// A and C in ONE launch (tmjx_ppo_loss_phases with both phases in the mask; round 6): the learner's main stream ran A (18 us), waited for B's records and ran
// C (22 us) — two latency-bound launches of the same (t, b) x PPO_G lane groups over the same logits / actions / fc2 rows between the policy's forward
// and backward chains, with the chip otherwise idle.  The bodies of k_ppo_a and k_ppo_c back to back, expression for expression (the log-prob stays in
// its lane group's registers; it is still written to scratch[i]); the four per-block partial sums land where A and C put them.
__global__ __launch_bounds__(PPO_BLOCK) void k_ppo_ac(PpoCfg c, const float *__restrict__ logits, const float *__restrict__ raw_action,
                                                      const float *__restrict__ behaviour_logp, const float *__restrict__ noise,
                                                      const float *__restrict__ baseline, const float *__restrict__ fc2, float *dlogits,
                                                      float *dbaseline, float *dfc2, float *scratch, int nblk, const float *rec, int nrec) {
  __shared__ float lds[64];
  __shared__ float scal_s[8];
  const int N = c.T * c.B, gid = blockIdx.x * PPO_BLOCK + threadIdx.x, i = gid / PPO_G, sub = gid % PPO_G;
  const float *vs = scratch + N, *adv = scratch + 2 * (size_t)N;
  if (threadIdx.x == 0) ppo_combine_records(c, rec, nrec, scal_s);
  __syncthreads();
  const float *scal = scal_s;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // entropy, kl0 inner sum, klt inner sum | surrogate
  if (i < N) {
    const int t = i / c.B;
    const float *lg = logits + (size_t)i * 2 * c.A, *xa = raw_action + (size_t)i * c.A, *nz = noise + (size_t)i * c.A;
    // ---- A
    float logp = 0.f, ent = 0.f;
    for (int a = sub; a < c.A; a += PPO_G) {
      float loc = lg[a], scale = ppo_softplus(lg[c.A + a]) + 0.001f, x = xa[a], d = (x - loc) / scale;
      logp += -0.5f * d * d - logf(scale) - 0.91893853320467274f - ppo_fldj(x);
      float xs = loc + scale * nz[a];
      ent += 0.5f + 0.91893853320467274f + logf(scale) + ppo_fldj(xs);
    }
    logp = ppo_group_sum(logp);
    if (sub == 0) scratch[i] = logp;
    acc[0] = ent;
    const float *f = fc2 + (size_t)i * 2 * c.Z;
    const float pv = 1.f - PPO_ALPHA * PPO_ALPHA;
    {
      float s = 0.f;
      if (t == 0) {
        for (int z = sub; z < c.Z; z += PPO_G) { float m = f[z], lv = f[c.Z + z]; s += 1.f + lv - m * m - expf(lv); }
        acc[1] = s;
      } else {
        const float *fp = f - (size_t)c.B * 2 * c.Z;
        for (int z = sub; z < c.Z; z += PPO_G) { float m = f[z], lv = f[c.Z + z], e = PPO_ALPHA * fp[z] - m; s += expf(lv) / pv + e * e / pv - 1.f + logf(pv) - lv; }
        acc[2] = s;
      }
    }
    // ---- C
    const float Nf = (float)N;
    float ad = adv[i];
    if (c.normalize_advantage) ad = (ad - scal[0]) / (scal[1] + 1e-8f);
    float rho = expf(logp - behaviour_logp[i]), lo = 1.f - c.clip_eps, hi = 1.f + c.clip_eps;
    float rc = fminf(fmaxf(rho, lo), hi), s1 = rho * ad, s2 = rc * ad;
    if (sub == 0) acc[3] = fminf(s1, s2);
    float glp = (s1 <= s2 || (rho >= lo && rho <= hi)) ? -ad * rho / Nf : 0.f;
    const float ce = -c.entropy_cost / Nf;
    float *dl = dlogits + (size_t)i * 2 * c.A;
    for (int a = sub; a < c.A; a += PPO_G) {
      float loc = lg[a], raw = lg[c.A + a], scale = ppo_softplus(raw) + 0.001f, sig = ppo_sigmoid(raw), x = xa[a], d = x - loc;
      float inv = 1.f / scale, dlp_loc = d * inv * inv, dlp_scale = d * d * inv * inv * inv - inv;
      float n = nz[a], th = tanhf(loc + scale * n), de_loc = -2.f * th, de_scale = inv - 2.f * th * n;
      dl[a] = glp * dlp_loc + ce * de_loc;
      dl[c.A + a] = (glp * dlp_scale + ce * de_scale) * sig;
    }
    if (sub == 0) dbaseline[i] = -0.5f * (vs[i] - baseline[i]) / Nf;
    const float kw = c.kl_weight;
    const float c0 = kw / (float)c.T / (float)(c.B * c.Z), c1 = c.T > 1 ? kw * (float)(c.T - 1) / (float)c.T / (float)((c.T - 1) * c.B * c.Z) : 0.f;
    const float *fp = f - (size_t)c.B * 2 * c.Z, *fn = f + (size_t)c.B * 2 * c.Z;
    float *df = dfc2 + (size_t)i * 2 * c.Z;
    for (int z = sub; z < c.Z; z += PPO_G) {
      float m = f[z], lv = f[c.Z + z], dm, dlv;
      if (t == 0) { dm = c0 * m; dlv = -0.5f * c0 * (1.f - expf(lv)); }
      else { float e = PPO_ALPHA * fp[z] - m; dm = -c1 * e / pv; dlv = 0.5f * c1 * (expf(lv) / pv - 1.f); }
      if (t + 1 < c.T) { float e2 = PPO_ALPHA * m - fn[z]; dm += c1 * PPO_ALPHA * e2 / pv; }
      df[z] = dm; df[c.Z + z] = dlv;
    }
  }
  ppo_block_sum<4>(acc, lds);
  if (threadIdx.x == 0) { float *p = scratch + (size_t)4 * N + (size_t)blockIdx.x * 4; p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2]; p[3] = acc[3]; }
}

// D: final scalars.  sum_parts (tmjx_ppo_loss_phases): the records do not carry A's entropy / KL sums (B ran next to A, on the value network's
// stream) — D adds A's per-block partials up itself
__global__ void k_ppo_d(PpoCfg c, const float *scratch, float *out, int nblk, const float *rec = nullptr, int nrec = 0, int sum_parts = 0) {
  __shared__ float lds[64];
  __shared__ float scal_s[8];
  const int N = c.T * c.B;
  const float *part = scratch + 4 * (size_t)N, *scal = part + (size_t)4 * nblk;
  if (rec) {
    if (threadIdx.x == 0) ppo_combine_records(c, rec, nrec, scal_s);
    __syncthreads();
    scal = scal_s;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) {
    acc[0] += part[4 * k + 3];
    if (sum_parts) { acc[1] += part[4 * k]; acc[2] += part[4 * k + 1]; acc[3] += part[4 * k + 2]; }
  }
  ppo_block_sum<4>(acc, lds);
  if (threadIdx.x == 0) {
    if (sum_parts) {      // (the arithmetic of ppo_combine_records on the three sums)
      const float n = (float)N;
      scal_s[3] = acc[1] / n;
      float kl0 = -0.5f * acc[2] / (float)(c.B * c.Z), kl = kl0;
      if (c.T > 1) { float klt = 0.5f * acc[3] / (float)((c.T - 1) * c.B * c.Z); kl = (kl0 + klt * (float)(c.T - 1)) / (float)c.T; }
      scal_s[4] = c.kl_weight * kl;
    }
    float policy = -acc[0] / (float)N, v = scal[2], entl = -c.entropy_cost * scal[3], kl = scal[4];
    const float o[8] = {policy + v + entl + kl, policy, v, entl, kl, scal[0], scal[1], scal[3]};
    // accumulate: `out` is a running sum over the minibatch steps of an update (the learner's metric accumulator: no add launch per step)
    for (int i = 0; i < 8; i++) out[i] = c.accumulate ? out[i] + o[i] : o[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Dense -> SiLU -> LayerNorm block epilogue (intention_network.py:14-88: flax Dense, nn.silu, nn.LayerNorm eps 1e-6):
//   y = gamma * (a - mean(a)) * rstd(a) + beta,  a = silu(z + bias),  z = x W^T from the library GEMM.
// One wavefront per row, VPT = H / 64 consecutive columns per lane (H = 256: one float4), row statistics by wave
// shuffles.  The backward kernel also produces the column sums d_gamma, d_beta, d_bias as per-block partials (rows are
// dealt to blocks in contiguous slabs) that a second small kernel adds up: deterministic, no atomics.
#define BLK_ROWS_PER_BLOCK 32
__device__ __forceinline__ float wave_sum(float x) { for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off); return x; }

// Column of register k of a lane: the lane's VPT columns are dealt in groups of four CONSECUTIVE ones, group j of lane l at column 4 (64 j + l):
// a wave instruction then reads 64 adjacent 16-byte pieces = whole cache lines (VPT consecutive columns per lane — 32 / 64 bytes apart at H = 512 /
// 1024 — used a quarter of every line per instruction: 2.3 TB/s; up to H = 256 the two mappings are the same).
template <int VPT> __device__ __forceinline__ int silu_ln_col(int lane, int k) { return VPT >= 4 ? 4 * (64 * (k >> 2) + lane) + (k & 3) : lane * VPT + k; }
// two floats -> two bf16 in one dword (lo in the low half), round to nearest even (v_cvt_pk_bf16_f32: the rounding csrc/gemm_bf16.h applies
// when it stages an fp32 operand, so a bf16 output of these kernels feeds the bf16 GEMMs the bits they would have made themselves)
typedef __bf16 slb2 __attribute__((ext_vector_type(2)));
typedef float slf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned silu_ln_pack(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(slf2{lo, hi}, slb2)); }
// store the VPT values of a lane to row `o` (fp32) or `o16` (bf16), in 16- / 8-byte pieces where the mapping has them
template <int VPT, bool OUT16>
__device__ __forceinline__ void silu_ln_store(const float (&v)[VPT], float *__restrict__ o, unsigned short *__restrict__ o16, int lane) {
  if constexpr (VPT >= 4) {
#pragma unroll
    for (int j = 0; j < VPT / 4; j++) {
      const int c = 4 * (64 * j + lane);
      if constexpr (OUT16) *reinterpret_cast<uint2 *>(o16 + c) = uint2{silu_ln_pack(v[4 * j], v[4 * j + 1]), silu_ln_pack(v[4 * j + 2], v[4 * j + 3])};
      else *reinterpret_cast<float4 *>(o + c) = float4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    }
  } else {
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      if constexpr (OUT16) o16[lane * VPT + k] = (unsigned short)(silu_ln_pack(v[k], 0.f) & 0xffffu);
      else o[lane * VPT + k] = v[k];
    }
  }
}
template <int VPT>
__device__ __forceinline__ void silu_ln_load(float (&v)[VPT], const float *__restrict__ src, int lane) {
  if constexpr (VPT >= 4) {
#pragma unroll
    for (int j = 0; j < VPT / 4; j++) {
      const float4 q = *reinterpret_cast<const float4 *>(src + 4 * (64 * j + lane));
      v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < VPT; k++) v[k] = src[lane * VPT + k];
  }
}

// OUT16: y as bf16 to y16 (leading dimension ldy16) instead of fp32 to y
template <int VPT, bool OUT16 = false>
__global__ __launch_bounds__(256) void k_silu_ln_fwd(const float *__restrict__ z, const float *__restrict__ bias, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float *__restrict__ y, float *__restrict__ stats, int rows,
                                                     float eps, unsigned short *__restrict__ y16 = nullptr, int ldy16 = 0) {
  TM_PRIO_ACTING();
  constexpr int H = VPT * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float b[VPT], g[VPT], be[VPT];
  silu_ln_load<VPT>(b, bias, lane); silu_ln_load<VPT>(g, gamma, lane); silu_ln_load<VPT>(be, beta, lane);
  const int wpb = blockDim.x >> 6;         // waves (= rows in flight) per block: 4, or 1 for the acting policy (tmjx_silu_ln_fwd)
  for (int r = blockIdx.x * wpb + w; r < rows; r += gridDim.x * wpb) {
    float a[VPT], s = 0.f;
    silu_ln_load<VPT>(a, z + (size_t)r * H, lane);
#pragma unroll
    for (int k = 0; k < VPT; k++) { float v = a[k] + b[k]; a[k] = tm_silu(v); s += a[k]; }
    float mean = wave_sum(s) / (float)H, q = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; k++) { float d = a[k] - mean; q += d * d; }
    float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int k = 0; k < VPT; k++) a[k] = (a[k] - mean) * rstd * g[k] + be[k];
    silu_ln_store<VPT, OUT16>(a, OUT16 ? nullptr : y + (size_t)r * H, OUT16 ? y16 + (size_t)r * ldy16 : nullptr, lane);
    if (lane == 0) { stats[2 * (size_t)r] = mean; stats[2 * (size_t)r + 1] = rstd; }
  }
}

// OUT16: d loss / d z as bf16 to dz16 (leading dimension lddz16) instead of fp32 to dz — the column sums are taken from the fp32 values either way
template <int VPT, bool OUT16 = false>
__global__ __launch_bounds__(256) void k_silu_ln_bwd(const float *__restrict__ dy, const float *__restrict__ z, const float *__restrict__ bias,
                                                     const float *__restrict__ gamma, const float *__restrict__ stats, float *__restrict__ dz,
                                                     float *__restrict__ partial, int rows, unsigned short *__restrict__ dz16 = nullptr, int lddz16 = 0) {
  constexpr int H = VPT * 64;
  __shared__ float lds[3 * H * 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float b[VPT], g[VPT], sg[VPT], sb[VPT], sz[VPT];
  silu_ln_load<VPT>(b, bias, lane); silu_ln_load<VPT>(g, gamma, lane);
#pragma unroll
  for (int k = 0; k < VPT; k++) { sg[k] = 0.f; sb[k] = 0.f; sz[k] = 0.f; }
  const int r0 = blockIdx.x * BLK_ROWS_PER_BLOCK, r1 = min(rows, r0 + BLK_ROWS_PER_BLOCK);
  for (int r = r0 + w; r < r1; r += 4) {
    const float mean = stats[2 * (size_t)r], rstd = stats[2 * (size_t)r + 1];
    float v[VPT], sig[VPT], ah[VPT], da[VPT], m1 = 0.f, m2 = 0.f;
    silu_ln_load<VPT>(v, z + (size_t)r * H, lane); silu_ln_load<VPT>(da, dy + (size_t)r * H, lane);
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      v[k] += b[k];
      sig[k] = tm_sigmoid(v[k]);
      ah[k] = (v[k] * sig[k] - mean) * rstd;
      const float d = da[k];
      sg[k] += d * ah[k]; sb[k] += d;
      da[k] = d * g[k];
      m1 += da[k]; m2 += da[k] * ah[k];
    }
    m1 = wave_sum(m1) / (float)H; m2 = wave_sum(m2) / (float)H;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      float dact = rstd * (da[k] - m1 - ah[k] * m2);
      float o = dact * (sig[k] * (1.f + v[k] * (1.f - sig[k])));
      da[k] = o;
      sz[k] += o;
    }
    silu_ln_store<VPT, OUT16>(da, OUT16 ? nullptr : dz + (size_t)r * H, OUT16 ? dz16 + (size_t)r * lddz16 : nullptr, lane);
  }
#pragma unroll
  for (int k = 0; k < VPT; k++) { int c = silu_ln_col<VPT>(lane, k); lds[(0 * 4 + w) * H + c] = sg[k]; lds[(1 * 4 + w) * H + c] = sb[k]; lds[(2 * 4 + w) * H + c] = sz[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * H; i += 256) {
    int which = i / H, c = i - which * H;
    partial[(size_t)blockIdx.x * 3 * H + i] = lds[(which * 4 + 0) * H + c] + lds[(which * 4 + 1) * H + c] + lds[(which * 4 + 2) * H + c] + lds[(which * 4 + 3) * H + c];
  }
}
// column sums of the per-block partials -> [3][H] = d_gamma | d_beta | d_bias.  Block = 32 columns x 8 row slices (the first
// version, one thread per column walking all partial rows, took 74 us for 320 x 768 floats: three workgroups, serial loads)
__global__ __launch_bounds__(256) void k_colsum(const float *__restrict__ partial, float *__restrict__ out, int nblk, int width) {
  __shared__ float lds[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, col = blockIdx.x * 32 + cx;
  // eight independent loads in flight per thread: with two the 80 loads of a thread (640 partial rows) were a chain of memory latencies
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < width) {
    int k = ry;
    for (; k + 56 < nblk; k += 64) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] += partial[(size_t)(k + 8 * u) * width + col];
    }
    for (; k < nblk; k += 8) acc[0] += partial[(size_t)k * width + col];
  }
  lds[ry][cx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (ry == 0 && col < width) { float s = 0.f; for (int j = 0; j < 8; j++) s += lds[j][cx]; out[col] = s; }
}

// The same contraction on the MATRIX cores, still without LDS: v_mfma_f32_16x16x4_f32 takes A[i][k] in lane 16 k + i and B[k][j] in
// lane 16 k + j, so a lane's float4 along K (k = 4 (lane / 16) + s, s = 0..3) feeds four MFMAs straight from global memory — any
// assignment of the 16 k of a step to the four MFMAs works as long as A and B use the same one.  One wave per 32 x 32 output tile
// (2 x 2 MFMA tiles, 16 accumulator registers), operands of the next K step in flight while the 16 MFMAs of this one issue.  Next
// to the other env group's physics kernel this matters twice: the fp32 MFMA rate equals the vector FMA rate, but the matrix pipe
// is idle there while the vector ALU is what the physics kernel is bound by; and a one-wave workgroup fits any free wave slot.
// __launch_bounds__(64, 5) (round 5): at most 96 registers.  Left to itself the allocator took 108 VGPRs + 32 AGPRs = 140 — and three physics waves of 136
// allocated registers leave 104 of a SIMD's 512: the layer's waves could only start where a physics wave had retired.  Capped, the kernel needs 77, no spill.
template <bool A_KMAJOR>
__global__ __launch_bounds__(64, 5) void k_linear_nolds_mfma(const float *__restrict__ A, long long sa_row, long long sa_k, const float *__restrict__ W,
                                                          const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                                          const float *__restrict__ mean = nullptr, const float *__restrict__ inv_std = nullptr) {
  TM_PRIO_ACTING();
  // mean / inv_std (optional, [K]): the operand is (A - mean) * inv_std — the observation normaliser applied while the raw observation is
  // loaded (the first layer of the LDS-free inference: no element-wise launch in front of it)
  typedef float __attribute__((ext_vector_type(4))) f4;
  const int lane = threadIdx.x, li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  int ar[2], wc[2];
#pragma unroll
  for (int t = 0; t < 2; t++) { int r = row0 + 16 * t + li; ar[t] = r < M ? r : M - 1; int c = col0 + 16 * t + li; wc[t] = c < N ? c : N - 1; }
  f4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  struct Frag { float4 a[2], w[2]; };
  auto load = [&](Frag &f, int k0) {
    const int k = k0 + 4 * kq;
    const bool ok = k < K;                       // K is a multiple of 4 (host check): a float4 is inside or outside as a whole
    const int kc = ok ? k : 0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      float4 v;
      if (A_KMAJOR) {
        v.x = A[(long long)kc * sa_k + ar[t]]; v.y = A[(long long)(kc + 1) * sa_k + ar[t]];
        v.z = A[(long long)(kc + 2) * sa_k + ar[t]]; v.w = A[(long long)(kc + 3) * sa_k + ar[t]];
      } else {
        v = *reinterpret_cast<const float4 *>(A + (long long)ar[t] * sa_row + kc);
      }
      if (mean) {
        const float4 mu = *reinterpret_cast<const float4 *>(mean + kc), is = *reinterpret_cast<const float4 *>(inv_std + kc);
        v = float4{(v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w};
      }
      f.a[t] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
      float4 w = *reinterpret_cast<const float4 *>(W + (size_t)wc[t] * K + kc);
      f.w[t] = ok ? w : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto mma = [&](const Frag &f) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[a].x, f.w[b].x, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[a].y, f.w[b].y, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[a].z, f.w[b].z, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[a].w, f.w[b].w, acc[a][b], 0, 0, 0);
      }
  };
  Frag f0, f1;
  load(f0, 0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    if (k0 + 16 < K) load(f1, k0 + 16);
    mma(f0);
    if (k0 + 16 >= K) break;
    if (k0 + 32 < K) load(f0, k0 + 32);
    mma(f1);
  }
  // accumulator register r of lane l holds C[4 (l / 16) + r][l % 16] of its 16 x 16 tile
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int c = col0 + 16 * b + li;
      const float bv = (bias && c < N) ? bias[c] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = row0 + 16 * a + 4 * kq + r;
        if (row < M && c < N) C[(size_t)row * N + c] = acc[a][b][r] + bv;
      }
    }
}

// The acting policy's dense layer with a SMALL LDS tile (round 5).  The physics kernel's env image fell to 9 of a CU's 128 LDS granules, so twelve
// resident envs leave 20 granules = 25 600 bytes free on every CU, and at ~130 VGPRs three physics waves leave room for one more wave of <= 104
// registers on every SIMD: a 4-wave workgroup with <= 20 480 bytes of LDS fits next to a full house of physics workgroups at any time.  Against
// the LDS-free kernel above (one wave per 32 x 32 tile, every operand word fetched from L2 by the wave that uses it, the K loop a chain of
// exposed L2 latencies): 64 x 64 tiles staged through LDS (operands re-read 4 x less), the next K step's global loads in flight in registers
// while this step's 32 MFMAs per wave issue.  Row-major A with 16-byte aligned rows, K % 4 == 0, W [N][ldw] with 16-byte aligned rows.
// NORM: the operand is (A - mean[k]) * inv_std[k] (the observation normaliser, first layer), applied when the staged registers go to LDS.
#define ACT_BM 64
#define ACT_BK 32
// __launch_bounds__(256, 5): at most 96 registers (accumulators included) — three physics waves of 136 allocated registers leave 104 of a SIMD's 512
#define ACT_LD 40          // floats per LDS row (32 + 8: b128 fragment reads of 16 consecutive rows hit 16 distinct slots, as in gemm_kernels.h)
// BMR = 64: a 64 x 64 tile, every wave 32 x 32 (2 x 2 MFMA tiles).  BMR = 32 (round 6): a 32 x 64 tile, every wave 32 x 16 (2 x 1) — half the MFMAs per
// K step and workgroup and twice the workgroups: at the acting policy's 1 365 rows a layer is ONE workgroup's time (88 resp. 172 workgroups all run at
// once on 256 CUs), and of that time the 32 matrix instructions per wave and step were the larger half (12 -> ~8 us per 256-wide layer).  Same k order
// per output element: bit-identical results.
template <bool NORM, int BMR>
__global__ __launch_bounds__(256, 5) void k_linear_act(const float *__restrict__ A, long long lda, const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                    float *__restrict__ C, int M, int N, int K, const float *__restrict__ mean, const float *__restrict__ inv_std) {
  TM_PRIO_ACTING();
  typedef float __attribute__((ext_vector_type(4))) f4;
  constexpr int NA = BMR / 32, NB = BMR == 64 ? 2 : 1;      // float4 of A per thread and K step; MFMA column tiles per wave
  __shared__ __attribute__((aligned(16))) float sA[BMR * ACT_LD], sW[ACT_BM * ACT_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * BMR, col0 = blockIdx.y * ACT_BM;
  const int wr = BMR == 64 ? (wave >> 1) * 32 : 0, wc = BMR == 64 ? (wave & 1) * 32 : wave * 16;
  const float *pa[2], *pw[2];
  int sr[2], sc[2];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int f = t + 256 * p, r = f >> 3, c4 = f & 7;
    sr[p] = r; sc[p] = 4 * c4;
    pa[p] = A + (long long)min(row0 + min(r, BMR - 1), M - 1) * lda;   // rows / columns outside the matrix: clamped (their products are never stored)
    pw[p] = W + (size_t)min(col0 + r, N - 1) * ldw;
  }
  f4 ra[2], rw[2];
  auto load = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const bool ok = k0 + sc[p] < K;                // K % 4 == 0: a float4 lies inside or outside as a whole
      const int kc = ok ? k0 + sc[p] : 0;            // (outside: the row's FIRST float4, always readable — K >= 4 — and zeroed below)
      const f4 w = *reinterpret_cast<const f4 *>(pw[p] + kc);
      rw[p] = ok ? w : f4{0.f, 0.f, 0.f, 0.f};
      if (p < NA) {
        f4 a = *reinterpret_cast<const f4 *>(pa[p] + kc);
        if (NORM) {
          const f4 mu = *reinterpret_cast<const f4 *>(mean + kc), is = *reinterpret_cast<const f4 *>(inv_std + kc);
          a = (a - mu) * is;
        }
        ra[p] = ok ? a : f4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  f4 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < NB; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  load(0);
  for (int k0 = 0; k0 < K; k0 += ACT_BK) {
    __syncthreads();                                  // every wave has read the previous step's fragments
#pragma unroll
    for (int p = 0; p < 2; p++) {
      if (p < NA) *reinterpret_cast<f4 *>(sA + sr[p] * ACT_LD + sc[p]) = ra[p];
      *reinterpret_cast<f4 *>(sW + sr[p] * ACT_LD + sc[p]) = rw[p];
    }
    __syncthreads();
    if (k0 + ACT_BK < K) load(k0 + ACT_BK);           // in flight while this step's MFMAs issue
#pragma unroll
    for (int c = 0; c < 2; c++) {
      f4 fa[2], fw[NB];
#pragma unroll
      for (int a = 0; a < 2; a++) fa[a] = *reinterpret_cast<const f4 *>(sA + (wr + 16 * a + li) * ACT_LD + 16 * c + 4 * kq);
#pragma unroll
      for (int b = 0; b < NB; b++) fw[b] = *reinterpret_cast<const f4 *>(sW + (wc + 16 * b + li) * ACT_LD + 16 * c + 4 * kq);
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < NB; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][e], fw[b][e], acc[a][b], 0, 0, 0);
    }
  }
  // accumulator register r of lane l holds C[4 (l / 16) + r][l % 16] of its 16 x 16 tile
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int c = col0 + wc + 16 * b + li;
      const float bv = (bias && c < N) ? bias[c] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = row0 + wr + 16 * a + 4 * kq + r;
        if (row < M && c < N) C[(size_t)row * N + c] = acc[a][b][r] + bv;
      }
    }
}

// minibatch gather + observation normalisation in one pass: out[t][b][:] = (src[t][idx[b]][:] - mean) / std
// (index_select + two element-wise passes over 57 MB otherwise); float4 lanes, W = obs width (multiple of 4)
__global__ __launch_bounds__(256) void k_gather_normalize(const float *__restrict__ src, const long long *__restrict__ idx, const float *__restrict__ mean,
                                                          const float *__restrict__ stdv, float *__restrict__ out, int T, int R, int B, int W) {
  const int w4 = W >> 2;
  const size_t total = (size_t)T * B * w4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int c = (int)(i % w4);
    size_t tb = i / w4;
    int b = (int)(tb % B), t = (int)(tb / B);
    const float4 v = reinterpret_cast<const float4 *>(src + ((size_t)t * R + (size_t)idx[b]) * W)[c];
    const float4 m = reinterpret_cast<const float4 *>(mean)[c], s = reinterpret_cast<const float4 *>(stdv)[c];
    float4 o = {(v.x - m.x) / s.x, (v.y - m.y) / s.y, (v.z - m.z) / s.z, (v.w - m.w) / s.w};
    reinterpret_cast<float4 *>(out + tb * W)[c] = o;
  }
}

// The whole minibatch of one SGD step in ONE launch (ppo.py:304-317: every leaf of the roll-out data is indexed by the same permutation slice):
// normalised observations [T][B][W] and bootstrap observations [B][W] (float4 lanes), raw actions [T][B][A] and the four per-step scalars
// (log-prob, reward, discount, truncation) [4][T][B] — seven index_select / gather launches of 4-20 us each otherwise, serialised in front of
// the first GEMM of the step.
#define TM_MB_SUBTICKETS 64
#define TM_MB_STATE_HEAD 16      // int64 words in front of the sub-tickets: {draw counter, slot, top ticket, ...}
#define TM_MB_STATE_STRIDE 16    // one 128-byte line per sub-ticket
struct MinibatchGather {
  const float *obs, *next_last, *raw_action, *scalar[4];
  const long long *idx;       // the minibatch rows are idx[slot * B .. + B), slot = state[1] (0 without state)
  const float *mean, *stdv;
  float *obs_n, *next_n, *raw_action_g, *scalars_g;
  float *eps, *noise;         // N(0, 1) draws [T B Z] / [T B A] (either may be null)
  long long *state;           // device int64[TM_MB_STATE_HEAD + TM_MB_STATE_STRIDE * TM_MB_SUBTICKETS]: {draw counter, slot, ticket, .. | sub-tickets} or null
  unsigned long long seed;
  int T, R, B, W, A, Z, advance;
  unsigned short *obs_n16; int ld16;      // optional bf16 twin of obs_n: [T B][ld16 >= W] (columns beyond W are the caller's: zeros)
};
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 constants)
__device__ __forceinline__ void tm_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned *out) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// four N(0, 1) draws for elements 4 q .. 4 q + 3 of stream `sid` at draw counter `ctr` (Box-Muller on 24-bit uniforms in (0, 1))
__device__ __forceinline__ void tm_normal4(unsigned long long seed, unsigned long long ctr, unsigned sid, unsigned q, float *n) {
  unsigned x[4];
  tm_philox4x32_10(q, sid, (unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)seed, (unsigned)(seed >> 32), x);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float u1 = ((float)(x[2 * h] >> 8) + 0.5f) * (1.f / 16777216.f), u2 = ((float)(x[2 * h + 1] >> 8) + 0.5f) * (1.f / 16777216.f);
    const float r = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincospif(2.f * u2, &sn, &cs);
    n[2 * h] = r * cs; n[2 * h + 1] = r * sn;
  }
}
__global__ __launch_bounds__(256) void k_gather_minibatch(MinibatchGather g) {
  const int w4 = g.W >> 2;
  // every block reads the state before it can have been advanced: the advance is made by the block that FINISHES last (ticket below)
  const unsigned long long ctr = g.state ? (unsigned long long)g.state[0] : 0ull;
  const long long *idx = g.idx + (g.state ? g.state[1] * (long long)g.B : 0ll);
  const size_t n_obs = (size_t)g.T * g.B * w4, n_next = (size_t)g.B * w4, n_act = (size_t)g.T * g.B * g.A, n_sc = (size_t)4 * g.T * g.B;
  const size_t q_eps = g.eps ? ((size_t)g.T * g.B * g.Z + 3) / 4 : 0, q_noise = g.noise ? (n_act + 3) / 4 : 0;
  const size_t e0 = n_obs + n_next, e1 = e0 + n_act, e2 = e1 + n_sc, e3 = e2 + q_eps, total = e3 + q_noise;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    if (i < e0) {
      const bool nx = i >= n_obs;
      const size_t j = nx ? i - n_obs : i;
      const int c = (int)(j % w4);
      const size_t tb = j / w4;
      const int b = (int)(tb % g.B), t = (int)(tb / g.B);
      const float *src = nx ? g.next_last : g.obs;
      const float4 v = reinterpret_cast<const float4 *>(src + ((size_t)t * g.R + (size_t)idx[b]) * g.W)[c];
      const float4 m = reinterpret_cast<const float4 *>(g.mean)[c], s = reinterpret_cast<const float4 *>(g.stdv)[c];
      const float4 o = {(v.x - m.x) / s.x, (v.y - m.y) / s.y, (v.z - m.z) / s.z, (v.w - m.w) / s.w};
      reinterpret_cast<float4 *>((nx ? g.next_n : g.obs_n) + tb * g.W)[c] = o;
      if (!nx && g.obs_n16) *reinterpret_cast<uint2 *>(g.obs_n16 + tb * g.ld16 + 4 * c) = uint2{silu_ln_pack(o.x, o.y), silu_ln_pack(o.z, o.w)};
    } else if (i < e1) {
      const size_t j = i - e0;
      const int c = (int)(j % g.A);
      const size_t tb = j / g.A;
      const int b = (int)(tb % g.B), t = (int)(tb / g.B);
      g.raw_action_g[j] = g.raw_action[((size_t)t * g.R + (size_t)idx[b]) * g.A + c];
    } else if (i < e2) {
      const size_t j = i - e1;
      const int b = (int)(j % g.B);
      const size_t kt = j / g.B;                      // k * T + t
      const int t = (int)(kt % g.T), k = (int)(kt / g.T);
      g.scalars_g[j] = g.scalar[k][(size_t)t * g.R + (size_t)idx[b]];
    } else {
      const bool second = i >= e3;
      const size_t q = second ? i - e3 : i - e2, n = second ? n_act : (size_t)g.T * g.B * g.Z;
      float v[4];
      tm_normal4(g.seed, ctr, second ? 1u : 0u, (unsigned)q, v);
      float *dst = second ? g.noise : g.eps;
#pragma unroll
      for (int k = 0; k < 4; k++) if (4 * q + k < n) dst[4 * q + k] = v[k];
    }
  }
  if (g.state && g.advance) {
    // No fences: a block's reads of state[0 .. 1] are complete before its loop ends (their values were used), hence before its ticket; the
    // write below happens after every ticket, and the kernel boundary publishes it.  (An agent-scope __threadfence() per block — an L2
    // write-back on gfx950 — made this 20 us kernel take 370 us.)
    // Two-level ticket: 8192 atomics on ONE address serialise in its L2 channel (~10 ns each: 80 us); 64 sub-tickets on separate cache
    // lines take them in parallel, the last block of each sub-group then takes one of 64 top-level tickets.
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned sub = blockIdx.x % TM_MB_SUBTICKETS, nsub = min((unsigned)gridDim.x, (unsigned)TM_MB_SUBTICKETS);
      const unsigned members = (gridDim.x - sub + TM_MB_SUBTICKETS - 1) / TM_MB_SUBTICKETS;       // blocks with this residue
      unsigned long long *st = (unsigned long long *)g.state, *subt = st + TM_MB_STATE_HEAD + (size_t)TM_MB_STATE_STRIDE * sub;
      if (__hip_atomic_fetch_add(subt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)members - 1ull) {
        __hip_atomic_store(subt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(st + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)nsub - 1ull) {
          __hip_atomic_store(st + 2, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every block has finished, i.e. has read state[0 .. 1]
          g.state[0] += 1; g.state[1] += 1;
        }
      }
    }
  }
}
// raw Philox words for the known-answer test (tests/test_gpu_parity.py)
__global__ void k_philox_kat(const unsigned *ctr_key, unsigned *out) {
  if (threadIdx.x == 0) tm_philox4x32_10(ctr_key[0], ctr_key[1], ctr_key[2], ctr_key[3], ctr_key[4], ctr_key[5], out);
}

// ---- policy inference tails (make_inference_fn, ppo_networks.py:46-96; reparameterize, intention_network.py:78-88)
// latent sample + decoder input in one pass:  x[i] = [ mean + eps * exp(logvar / 2)  |  obs[i][ref:] ]
__global__ __launch_bounds__(256) void k_latent_concat(const float *__restrict__ fc2, const float *__restrict__ eps, const float *__restrict__ obs,
                                                       float *__restrict__ x, int n, int Z, int obs_w, int ref_w, long long obs_s0, long long obs_s1,
                                                       const float *__restrict__ mean, const float *__restrict__ stdv, int x_stride,
                                                       unsigned long long seed, const long long *__restrict__ rng_state) {
  TM_PRIO_ACTING();
  // eps == nullptr: the latent noise is drawn here, Philox stream 2 of (seed, draw counter rng_state[0]) — the acting policy's graph then
  // holds no torch generator (two state fills per replay) and no normal_ launch; the pad columns [W, x_stride) are written as zeros
  const int W = Z + obs_w - ref_w;
  const unsigned long long ctr = rng_state ? (unsigned long long)rng_state[0] : 0ull;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * x_stride; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % x_stride); size_t e = i / x_stride;
    float v = 0.f;
    if (c < Z) {
      float ep;
      if (eps) ep = eps[e * Z + c];
      else { float v4[4]; const size_t idx = e * Z + c; tm_normal4(seed, ctr, 2u, (unsigned)(idx >> 2), v4); ep = v4[idx & 3]; }
      v = fmaf(ep, expf(0.5f * fc2[e * 2 * Z + Z + c]), fc2[e * 2 * Z + c]);       // (written as the fused multiply-add it compiles to: the encoder chain's latent tail, csrc/mlp_chain.h, forms the same)
    } else if (c < W) { int k = ref_w + c - Z; v = obs[(long long)e * obs_s0 + (long long)k * obs_s1]; if (mean) v = (v - mean[k]) / stdv[k]; }
    x[e * (size_t)x_stride + c] = v;
  }
}
// gradient of the latent sample w.r.t. the two halves of fc2 = [mean | logvar] from d x (the decoder-input gradient, row stride
// dx_stride): d mean = dx[:, :Z];  d logvar = dx[:, :Z] * eps * exp(logvar / 2) / 2   (reparameterize, intention_network.py:78-88)
// `add` (optional, [n][2 Z]): another gradient of fc2 — the KL term's, from the loss head — summed in here, so that autograd has ONE gradient for
// fc2 and launches no element-wise add for the two (7.5 us per minibatch step, the last torch element-wise kernel of the captured SGD step)
__global__ __launch_bounds__(256) void k_latent_concat_bwd(const float *__restrict__ dx, const float *__restrict__ eps, const float *__restrict__ fc2,
                                                           float *__restrict__ dfc2, int n, int Z, int dx_stride, const float *__restrict__ add) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n * Z; i += (size_t)gridDim.x * 256) {
    int c = (int)(i % Z); size_t e = i / Z;
    float g = dx[e * (size_t)dx_stride + c];
    float gm = g, gv = g * eps[e * Z + c] * (0.5f * expf(0.5f * fc2[e * 2 * Z + Z + c]));
    if (add) { gm = add[e * 2 * Z + c] + gm; gv = add[e * 2 * Z + Z + c] + gv; }
    dfc2[e * 2 * Z + c] = gm;
    dfc2[e * 2 * Z + Z + c] = gv;
  }
}
// action sample, tanh post-processing and log-prob of the sample: one lane group of PPO_G per env
__global__ __launch_bounds__(PPO_BLOCK) void k_sample_action(const float *__restrict__ logits, const float *__restrict__ noise, float *__restrict__ raw,
                                                             float *__restrict__ action_t, float *__restrict__ logp, int n, int A,
                                                             unsigned long long seed, long long *__restrict__ rng_state) {
  TM_PRIO_ACTING();
  // noise == nullptr: drawn here (Philox stream 3 of (seed, rng_state[0])); the workgroup that finishes last then advances the draw counter
  // for the next inference (every workgroup has read it by then; ticket in rng_state[1])
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, e = gid / PPO_G, sub = gid % PPO_G;
  const unsigned long long ctr = rng_state ? (unsigned long long)rng_state[0] : 0ull;
  float lp = 0.f;
  if (e < n) {
    const float *lg = logits + (size_t)e * 2 * A;
    for (int a = sub; a < A; a += PPO_G) {
      float nz;
      if (noise) nz = noise[(size_t)e * A + a];
      else { float v4[4]; const size_t idx = (size_t)e * A + a; tm_normal4(seed, ctr, 3u, (unsigned)(idx >> 2), v4); nz = v4[idx & 3]; }
      float loc = lg[a], scale = ppo_softplus(lg[A + a]) + 0.001f, x = loc + scale * nz, d = (x - loc) / scale;
      raw[(size_t)e * A + a] = x;
      action_t[(size_t)a * n + e] = tanhf(x);          // [A][n]: the env-minor layout tmjx_step takes
      lp += -0.5f * d * d - logf(scale) - 0.91893853320467274f - ppo_fldj(x);
    }
  }
  lp = ppo_group_sum(lp);
  if (e < n && sub == 0) logp[e] = lp;
  if (rng_state) {
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long *st = (unsigned long long *)rng_state;
      if (__hip_atomic_fetch_add(st + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1ull) {
        __hip_atomic_store(st + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rng_state[0] += 1;
      }
    }
  }
}

// ---- LDS-free dense layer for the policy inference that runs NEXT TO the physics kernel --------------------------------------
// C[M][N] = A' W^T + bias, A'[i][k] = (A[i * sa_row + k * sa_k] - mean[k]) / std[k] (mean == nullptr: A as is), W [N][K] row-major.
// The wave-per-env physics kernel owns all 160 KB of LDS of every CU for ~2 ms at a time; a library GEMM (LDS tiles) launched
// on another stream cannot start until physics workgroups retire.  This kernel keeps its 4 x 4 register tile per thread in
// VGPRs and reads operands through L1/L2 only, so it co-runs with the physics kernel and its time disappears behind it.
// Throughput is modest (plain v_fma, ~10 TFLOP/s) and irrelevant: the whole inference is 1.7 GFLOP per 2048 envs.
#define NL_TM 4
#define NL_TN 4
// V = vector width of the loads along K (largest of 4 / 2 / 1 dividing K, with 4 V-byte aligned rows); A_KMAJOR: A[i][k] at
// A[k * sa_k + i] (the env-minor observation buffer): one float4 then fetches the thread's four consecutive rows.
template <int V> struct NlVec;
template <> struct NlVec<4> { typedef float4 T; };
template <> struct NlVec<2> { typedef float2 T; };
template <> struct NlVec<1> { typedef float T; };
template <int V> __device__ __forceinline__ void nl_unpack(const typename NlVec<V>::T &v, float *o);
template <> __device__ __forceinline__ void nl_unpack<4>(const float4 &v, float *o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void nl_unpack<2>(const float2 &v, float *o) { o[0] = v.x; o[1] = v.y; }
template <> __device__ __forceinline__ void nl_unpack<1>(const float &v, float *o) { o[0] = v; }

template <int V, bool A_KMAJOR>
__global__ __launch_bounds__(128) void k_linear_nolds(const float *__restrict__ A, long long sa_row, long long sa_k, const float *__restrict__ W,
                                                      const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K) {
  TM_PRIO_ACTING();
  typedef typename NlVec<V>::T VT;
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;                   // 16 x 8 threads: a 64-row x 32-column tile per workgroup
  const int row0 = blockIdx.x * 64 + ty * NL_TM, col0 = blockIdx.y * 32 + tx * NL_TN;
  float acc[NL_TM][NL_TN];
#pragma unroll
  for (int i = 0; i < NL_TM; i++)
#pragma unroll
    for (int j = 0; j < NL_TN; j++) acc[i][j] = 0.f;
  int cc[NL_TN];
#pragma unroll
  for (int j = 0; j < NL_TN; j++) cc[j] = col0 + j < N ? col0 + j : N - 1;      // clamp: out-of-range tiles compute duplicates, never store
  const int rbase = row0 + NL_TM <= M ? row0 : (M >= NL_TM ? M - NL_TM : 0);     // K-major loads need 4 valid consecutive rows
  int rr[NL_TM];
#pragma unroll
  for (int i = 0; i < NL_TM; i++) rr[i] = row0 + i < M ? row0 + i : M - 1;
  // register prefetch ring, two steps deep: the operands of steps k + V and k + 2V are in flight while step k is multiplied (one
  // or two waves per SIMD is all this kernel gets next to the physics kernel, so nothing else hides the L2 latency: the K loop is
  // a chain of load -> use round trips, and a ring one step deep left one full latency per step exposed)
  struct Stage { VT an[NL_TM], wn[NL_TN]; float4 akn[V]; };
  auto load = [&](Stage &st, int k) {
    if (A_KMAJOR) {
#pragma unroll
      for (int v = 0; v < V; v++) st.akn[v] = *reinterpret_cast<const float4 *>(A + (long long)(k + v) * sa_k + rbase);
    } else {
#pragma unroll
      for (int i = 0; i < NL_TM; i++) st.an[i] = *reinterpret_cast<const VT *>(A + (long long)rr[i] * sa_row + k);
    }
#pragma unroll
    for (int j = 0; j < NL_TN; j++) st.wn[j] = *reinterpret_cast<const VT *>(W + (size_t)cc[j] * K + k);
  };
  auto fma_stage = [&](const Stage &st) {
    float a[NL_TM][V], w[NL_TN][V];
    if (A_KMAJOR) {
#pragma unroll
      for (int v = 0; v < V; v++) { a[0][v] = st.akn[v].x; a[1][v] = st.akn[v].y; a[2][v] = st.akn[v].z; a[3][v] = st.akn[v].w; }
    } else {
#pragma unroll
      for (int i = 0; i < NL_TM; i++) nl_unpack<V>(st.an[i], a[i]);
    }
#pragma unroll
    for (int j = 0; j < NL_TN; j++) nl_unpack<V>(st.wn[j], w[j]);
#pragma unroll
    for (int v = 0; v < V; v++)
#pragma unroll
      for (int i = 0; i < NL_TM; i++)
#pragma unroll
        for (int j = 0; j < NL_TN; j++) acc[i][j] = fmaf(a[i][v], w[j][v], acc[i][j]);
  };
  Stage s0, s1, s2;
  load(s0, 0);
  if (V < K) load(s1, V);
  for (int k = 0; k < K; k += 3 * V) {          // the ring as three named stages (register arrays cannot be indexed dynamically)
    if (k + 2 * V < K) load(s2, k + 2 * V);
    fma_stage(s0);
    if (k + V >= K) break;
    if (k + 3 * V < K) load(s0, k + 3 * V);
    fma_stage(s1);
    if (k + 2 * V >= K) break;
    if (k + 4 * V < K) load(s1, k + 4 * V);
    fma_stage(s2);
  }
  const int rout = A_KMAJOR ? rbase : row0;
#pragma unroll
  for (int i = 0; i < NL_TM; i++)
#pragma unroll
    for (int j = 0; j < NL_TN; j++)
      if (rout + i < M && col0 + j < N && (!A_KMAJOR || rout == row0 || rout + i >= row0)) C[(size_t)(rout + i) * N + col0 + j] = acc[i][j] + (bias ? bias[col0 + j] : 0.f);
}

// ---- optimiser: optax.chain(clip_by_global_norm(max_norm), adam(lr)) (reference: agent/mlp_ppo/ppo.py:517-520) over FLAT fp32 buffers
// (parameters, gradients and the two moments as one contiguous array each): one launch instead of the norm-dependent scale kernels,
// the multi-tensor scale and the multi-tensor Adam.  `grad_norm` is the device scalar ||g||_2 of the (already averaged) gradient.
// sum of squares of the flat gradient as ADAM_NORM_PARTS per-workgroup partials (float4 loads, fixed summation order: every rank of a data-
// parallel run gets the same bits from the same all-reduced gradient); k_adam_clip adds them up itself — the global norm costs no launch of
// its own beyond this one and no library reduction kernel
#define ADAM_NORM_PARTS 256
__global__ __launch_bounds__(256) void k_grad_sumsq(const float *__restrict__ g, long long n, float *__restrict__ partial) {
  __shared__ float lds[4];
  const long long n4 = n >> 2;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)ADAM_NORM_PARTS * 256) {
    const float4 v = reinterpret_cast<const float4 *>(g)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float t = g[(n4 << 2) + threadIdx.x]; s += t * t; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}
// NORM_PARTS: grad_norm points at ADAM_NORM_PARTS partial sums of squares (k_grad_sumsq) instead of the norm itself; norm_out (optional) receives the norm
template <bool NORM_PARTS>
__global__ __launch_bounds__(256) void k_adam_clip(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                                   const float *__restrict__ grad_norm, long long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2, float max_norm, float *__restrict__ norm_out) {
  float nrm;
  if (NORM_PARTS) {       // every workgroup adds the same 256 numbers in the same order
    __shared__ float lds[4];
    float s = wave_sum(grad_norm[threadIdx.x]);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    nrm = sqrtf((lds[0] + lds[1]) + (lds[2] + lds[3]));
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = nrm;
  } else nrm = grad_norm[0];
  const float scale = max_norm / fmaxf(max_norm, nrm);               // g if ||g|| < max_norm else g / ||g|| * max_norm
  const float step = lr / bc1, rs = 1.f / sqrtf(bc2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gi = g[i] * scale;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step * mi / (sqrtf(vi) * rs + eps);
  }
}

// ---- column sums of a row-major [rows][width] matrix (bias gradients dy.sum(0) of the dense layers): stage 1 = one block per (64
// row chunks) x (32 columns), 8 row slices per block, partial sums to [nchunk][width]; stage 2 = k_colsum over the chunks.  torch's
// generic reduction takes 10-25 us for the 20 480 x (76 .. 256) inputs of a minibatch step.
#define COLSUM_CHUNKS 64
__global__ __launch_bounds__(256) void k_colsum_rows(const float *__restrict__ src, float *__restrict__ partial, int rows, int width) {
  __shared__ float lds[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, col = blockIdx.x * 32 + cx;
  const int per = (rows + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS, r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < width) {
    int r = r0 + ry;
    for (; r + 56 < r1; r += 64) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] += src[(size_t)(r + 8 * u) * width + col];
    }
    for (; r < r1; r += 8) acc[0] += src[(size_t)r * width + col];
  }
  lds[ry][cx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (ry == 0 && col < width) { float s = 0.f; for (int j = 0; j < 8; j++) s += lds[j][cx]; partial[(size_t)blockIdx.y * width + col] = s; }
}


// Several independent column-sum problems (per-workgroup partial rows -> one row) in ONE launch: the (d gamma | d beta | d bias) partials of
// every LayerNorm block of a backward pass, reduced once behind it instead of by one launch per block inside its dependent chain.
#define COLSUM_GROUP_MAX 16
struct ColsumProblem { const float *partial; float *out; int nblk, width, blk_begin; };
struct ColsumGroup { ColsumProblem p[COLSUM_GROUP_MAX]; int n; };
__global__ __launch_bounds__(256) void k_colsum_grouped(const ColsumGroup G) {
  __shared__ float lds[8][33];
  int i = 0;
  for (int j = 1; j < G.n; j++) if ((int)blockIdx.x >= G.p[j].blk_begin) i = j;       // uniform
  const ColsumProblem &P = G.p[i];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, col = (blockIdx.x - P.blk_begin) * 32 + cx;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < P.width) {
    int k = ry;
    for (; k + 56 < P.nblk; k += 64) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] += P.partial[(size_t)(k + 8 * u) * P.width + col];
    }
    for (; k < P.nblk; k += 8) acc[0] += P.partial[(size_t)k * P.width + col];
  }
  lds[ry][cx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (ry == 0 && col < P.width) { float s = 0.f; for (int j = 0; j < 8; j++) s += lds[j][cx]; P.out[col] = s; }
}

// ---- K6: observation normaliser update (brax running_statistics.update, reference math track_mjx/agent/masked_running_statistics.py:161-214)
// ONE pass over the roll-out observations [rows][W]: per column S1 = sum(x - mean_old), S2 = sum((x - mean_old)^2).  The reference's
//   mean_update = S1 / count_new,  variance_update = sum((x - mean_old)(x - mean_new)) = S2 - mean_update * S1
// follow from the two sums (same numbers up to fp32 rounding), and across ranks S1 and S2 are what is summed: the reference's three
// psums (count, mean_update, variance_update) become ONE all-reduce of 2 W floats (the count increment is known on the host).
// Stage 1: block = 256 lanes = 64 float4 column groups x 4 row slices, one slab of rows per blockIdx.y; the reads of a row are
// contiguous (W floats), eight rows in flight per lane.  HBM-bound: rows * W * 4 bytes read once.
#define STATS_SLABS 512
__global__ __launch_bounds__(256) void k_stats_partial(const float *__restrict__ src, const float *__restrict__ mean, float *__restrict__ partial,
                                                       long long rows, int W) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  __shared__ f4 lds[2][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6, c4 = blockIdx.x * 64 + cx, W4 = W >> 2;
  const long long per = (rows + STATS_SLABS - 1) / STATS_SLABS, r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (c4 < W4) {
    const f4 m = ((const f4 *)mean)[c4];
    const f4 *p = (const f4 *)src + c4;
    long long r = r0 + ry;
    for (; r + 28 < r1; r += 32) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = p[(size_t)(r + 4 * u) * W4];
#pragma unroll
      for (int u = 0; u < 8; u++) { f4 d = v[u] - m; s1 += d; s2 += d * d; }
    }
    for (; r < r1; r += 4) { f4 d = p[(size_t)r * W4] - m; s1 += d; s2 += d * d; }
  }
  lds[0][ry][cx] = s1; lds[1][ry][cx] = s2;
  __syncthreads();
  if (ry == 0 && c4 < W4) {
    f4 a = (lds[0][0][cx] + lds[0][1][cx]) + (lds[0][2][cx] + lds[0][3][cx]), b = (lds[1][0][cx] + lds[1][1][cx]) + (lds[1][2][cx] + lds[1][3][cx]);
    ((f4 *)(partial + (size_t)blockIdx.y * 2 * W))[c4] = a;
    ((f4 *)(partial + (size_t)blockIdx.y * 2 * W + W))[c4] = b;
  }
}
// Stage 3 (after the optional cross-rank sum of sums[2][W]): count, mean, summed_variance and std in place, one lane per column
__global__ void k_stats_finalize(const float *__restrict__ sums, float n_added, float *count, float *mean, float *summed_variance, float *std,
                                 int W, float std_min, float std_max) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const float cnt = *count + n_added;
  if (c < W) {
    const float s1 = sums[c], s2 = sums[W + c];
    const float upd = s1 / cnt;
    const float sv = summed_variance[c] + (s2 - upd * s1);
    mean[c] += upd;
    summed_variance[c] = sv;
    std[c] = fminf(fmaxf(sqrtf(fmaxf(sv, 0.f) / cnt), std_min), std_max);
  }
  // every block read the old count above; the last lane of the grid to get here would race with slower blocks' reads, so the count is
  // written by a second tiny launch (k_stats_count) — stream order makes it safe
}
__global__ void k_stats_count(float *count, float n_added) { *count += n_added; }

// ---- Dense -> SiLU (brax value MLP: swish, no LayerNorm) element-wise halves: y = silu(z + bias) for layers the fused GEMM epilogue
// (tmjx_gemm_nt_silu) does not take, and dz = dy silu'(z + bias) — the operand of the layer's input- and weight-gradient GEMMs
// (the bias gradient = column sums of dz rides along in the weight-gradient kernel).  float4 per lane where the width allows.
__global__ __launch_bounds__(256) void k_silu_fwd_f32(const float *__restrict__ z, const float *__restrict__ bias, float *__restrict__ y, long long total, int N) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  if (!(N & 3)) {
    const float4 zv = *reinterpret_cast<const float4 *>(z + i), bv = *reinterpret_cast<const float4 *>(bias + (int)(i % N));
    float v[4] = {zv.x + bv.x, zv.y + bv.y, zv.z + bv.z, zv.w + bv.w};
    *reinterpret_cast<float4 *>(y + i) = make_float4(tm_silu(v[0]), tm_silu(v[1]), tm_silu(v[2]), tm_silu(v[3]));
  } else {
    for (long long j = i; j < i + 4 && j < total; j++) { const float v = z[j] + bias[(int)(j % N)]; y[j] = tm_silu(v); }
  }
}
__global__ __launch_bounds__(256) void k_silu_bwd_f32(const float *__restrict__ dy, const float *__restrict__ z, const float *__restrict__ bias, float *__restrict__ dz,
                                                      long long total, int N) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  if (!(N & 3)) {
    const float4 zv = *reinterpret_cast<const float4 *>(z + i), bv = *reinterpret_cast<const float4 *>(bias + (int)(i % N)), dv = *reinterpret_cast<const float4 *>(dy + i);
    const float v[4] = {zv.x + bv.x, zv.y + bv.y, zv.z + bv.z, zv.w + bv.w}, d[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const float sig = tm_sigmoid(v[k]); o[k] = d[k] * (sig * (1.f + v[k] * (1.f - sig))); }
    *reinterpret_cast<float4 *>(dz + i) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    for (long long j = i; j < i + 4 && j < total; j++) { const float v = z[j] + bias[(int)(j % N)], sig = tm_sigmoid(v); dz[j] = dy[j] * (sig * (1.f + v * (1.f - sig))); }
  }
}

// The 1-wide head of the value MLP (brax make_value_network: MLP(hidden..., 1)) behind a Dense -> SiLU layer, backward:
//   k_silu_bwd_rank1_f32: d loss / d z of that layer = (dy1[m] w1[k]) silu'(z[m][k] + bias[k]) — the head's input gradient is an outer product, formed here
//     instead of by an input-gradient GEMM with a contraction length of one + k_silu_bwd_f32 (same expressions, same bits);
//   k_head_dw / k_head_dw_reduce: the head's weight gradient dw[k] = sum_m dy1[m] x[m][k] and bias gradient sum_m dy1[m] — a matrix-vector product, which the
//     128 x 128 tiles of k_gemm_dw run at 1 / 128 of their width.  Row slabs of 64 rows per workgroup, a thread per column (coalesced), fixed-order reduction.
__global__ __launch_bounds__(256) void k_silu_bwd_rank1_f32(const float *__restrict__ dy1, const float *__restrict__ w1, const float *__restrict__ z, const float *__restrict__ bias,
                                                            float *__restrict__ dz, long long total, int N) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;      // N % 4 == 0 (host check): the four elements share a row
  if (i >= total) return;
  const long long m = i / N;
  const int k = (int)(i - m * N);
  const float d = dy1[m];
  const float4 zv = *reinterpret_cast<const float4 *>(z + i), bv = *reinterpret_cast<const float4 *>(bias + k), wv = *reinterpret_cast<const float4 *>(w1 + k);
  const float v[4] = {zv.x + bv.x, zv.y + bv.y, zv.z + bv.z, zv.w + bv.w}, w[4] = {wv.x, wv.y, wv.z, wv.w};
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { const float sig = tm_sigmoid(v[j]); o[j] = (d * w[j]) * (sig * (1.f + v[j] * (1.f - sig))); }
  *reinterpret_cast<float4 *>(dz + i) = make_float4(o[0], o[1], o[2], o[3]);
}
#define HEAD_DW_ROWS 64
__global__ __launch_bounds__(256) void k_head_dw(const float *__restrict__ dy1, const float *__restrict__ x, int ldx, float *__restrict__ partial, int M, int K) {
  const int r0 = blockIdx.x * HEAD_DW_ROWS, r1 = min(M, r0 + HEAD_DW_ROWS);
  float *out = partial + (size_t)blockIdx.x * (K + 1);
  for (int k = threadIdx.x; k < K; k += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int m = r0;
    for (; m + 4 <= r1; m += 4) {
      a0 = fmaf(dy1[m], x[(size_t)m * ldx + k], a0); a1 = fmaf(dy1[m + 1], x[(size_t)(m + 1) * ldx + k], a1);
      a2 = fmaf(dy1[m + 2], x[(size_t)(m + 2) * ldx + k], a2); a3 = fmaf(dy1[m + 3], x[(size_t)(m + 3) * ldx + k], a3);
    }
    for (; m < r1; m++) a0 = fmaf(dy1[m], x[(size_t)m * ldx + k], a0);
    out[k] = (a0 + a1) + (a2 + a3);
  }
  if (threadIdx.x == 0) { float sb = 0.f; for (int m = r0; m < r1; m++) sb += dy1[m]; out[K] = sb; }
}
// 32 columns per workgroup, the slabs dealt to 8 groups of threads (a thread per column and group: 40 dependent loads at 320 slabs instead of 320), the groups'
// sums added through LDS in a fixed order
__global__ __launch_bounds__(256) void k_head_dw_reduce(const float *__restrict__ partial, float *__restrict__ dw, float *__restrict__ db, int slabs, int K) {
  __shared__ float red[8][32];
  const int c = threadIdx.x & 31, sg = threadIdx.x >> 5, k = blockIdx.x * 32 + c;
  float a0 = 0.f, a1 = 0.f;
  if (k <= K) {
    int s = sg;
    for (; s + 8 < slabs; s += 16) { a0 += partial[(size_t)s * (K + 1) + k]; a1 += partial[(size_t)(s + 8) * (K + 1) + k]; }
    if (s < slabs) a0 += partial[(size_t)s * (K + 1) + k];
  }
  red[sg][c] = a0 + a1;
  __syncthreads();
  if (sg == 0 && k <= K) {
    const float v = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
    if (k < K) dw[k] = v; else if (db) db[0] = v;
  }
}
// y[m] = x[m][:] . w + b: the 1-wide head forward as a matrix-vector product — a wave per row (16 rows per workgroup of 4 waves), a float4 of the row per lane
// and 256 columns per pass, the lane sums added by wave shuffles.  K % 4 == 0, rows of x 16-byte aligned (host check).
__global__ __launch_bounds__(256) void k_head_fwd(const float *__restrict__ x, int ldx, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ y, int M, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float b = bias ? bias[0] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int m = (blockIdx.x * 4 + wave) * 4 + j;
    if (m >= M) break;
    float a = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + (size_t)m * ldx + k), wv = *reinterpret_cast<const float4 *>(w + k);
      a = fmaf(xv.x, wv.x, a); a = fmaf(xv.y, wv.y, a); a = fmaf(xv.z, wv.z, a); a = fmaf(xv.w, wv.w, a);
    }
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) y[m] = a + b;
  }
}

// ---- roll-out buffer stores of one env-group step in ONE launch (agent/ppo.py: collect; brax acting.actor_step builds the Transition the
// same way, track_mjx/agent/mlp_ppo/ppo.py:330-348): the env's observation [W][n] (env-minor, what the kernels and the LDS-free inference
// read) transposed into the row-major roll-out buffer [n][W] (up to two destinations: row t + 1 of this unroll, or the unroll's
// next_observation_last and row 0 of the next unroll), raw action / log-prob of the acting policy, reward, discount = 1 - done, truncation.
// Seven torch copy launches per group step before (about 700 runtime copy kernels per training step).  No LDS (the other env group's
// physics kernel owns it): one lane per env x 16 observation rows, coalesced on the env-minor side, 64 contiguous bytes per lane on the other.
struct RolloutStore {
  const float *obs; float *obs_dst0, *obs_dst1;
  const float *raw; float *raw_dst;
  const float *logp; float *logp_dst;
  const float *reward; float *reward_dst;
  const float *done; float *discount_dst;
  const float *trunc; float *trunc_dst;
  int n, W, A;
  float *obs_dst2;
};
__global__ __launch_bounds__(64) void k_rollout_store(const RolloutStore s) {
  TM_PRIO_ACTING();
  const int e = blockIdx.x * 64 + threadIdx.x, ny = gridDim.y - 1;
  if ((int)blockIdx.y < ny) {
    if (e >= s.n || !s.obs) return;
    const int k0 = blockIdx.y * 16;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = k0 + j < s.W ? s.obs[(size_t)(k0 + j) * s.n + e] : 0.f;
    float *d[3] = {s.obs_dst0, s.obs_dst1, s.obs_dst2};
#pragma unroll
    for (int q = 0; q < 3; q++) {
      if (!d[q]) continue;
      float *o = d[q] + (size_t)e * s.W + k0;
      if (!(s.W & 3) && k0 + 16 <= s.W) {
#pragma unroll
        for (int j = 0; j < 16; j += 4) *reinterpret_cast<float4 *>(o + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
      } else {
#pragma unroll
        for (int j = 0; j < 16; j++) if (k0 + j < s.W) o[j] = v[j];
      }
    }
    return;
  }
  if (e < s.n) {
    if (s.logp_dst) s.logp_dst[e] = s.logp[e];
    if (s.reward_dst) s.reward_dst[e] = s.reward[e];
    if (s.discount_dst) s.discount_dst[e] = 1.f - s.done[e];
    if (s.trunc_dst) s.trunc_dst[e] = s.trunc[e];
  }
  if (s.raw_dst) {
    const int total = s.n * s.A, nth = gridDim.x * 64;
    for (int i = e; i < total; i += nth) s.raw_dst[i] = s.raw[i];
  }
}
