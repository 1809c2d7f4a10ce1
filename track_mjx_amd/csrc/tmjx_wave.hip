// csrc/tmjx_wave.hip — third translation unit of libtmjx_hip.so: K2, the wave-per-env physics kernel (csrc/wave_physics.h), alone.
//
// Alone because it is compiled with a flag the other kernels do not want (track_mjx_amd/hip.py: -mllvm -disable-machine-licm).  The kernel body is
// one loop over the control step's substeps around ~30 k instructions; machine LICM hoists every loop-invariant scalar of that body — lane masks,
// LDS offsets, model addresses: each a one-instruction s_mov / s_add — in front of the loop, where they do not fit the 104 SGPRs and are spilled to
// VGPR lanes (v_writelane / v_readlane + their wait states) to be read back where they could have been re-made.  Without it: 168 -> 131 VGPRs and
// env.step of 4096 envs 2.74 -> 2.68 ms (tools/ab_k2.sh, round 4); the GEMM kernels of the main unit lose ~ 0.5 % under the same flag.
#include <hip/hip_runtime.h>

#include "../../include/tmjx.h"
#include "wave_physics.h"

// K2, wave-per-env: one 64-lane workgroup per env, all per-substep state in LDS (csrc/wave_physics.h).
// STATIC = true: the rodent's dims and LDS map are compile-time constants (wave_layout.h).
// __launch_bounds__(64, 3): three waves per SIMD = twelve envs per CU, at most 168 VGPRs (the kernel takes ~130) — as a bound the compiler keeps, not a
// number a later edit silently exceeds (tests/test_abi.py pins it, with zero spills, on the code object's metadata).
// Round 5 brought the env's LDS image down to 9 granules (wave_layout.h: 14 envs per CU would fit) and the kernel builds at 128 VGPRs without a
// spill under __launch_bounds__(64, 4) (-DTMW_WAVES_PER_SIMD=4) — and that build is SLOWER: scheduled for four waves per SIMD the same source runs
// 13 % longer per wave at equal residency (3072 envs at 12 per CU: 1.98 against 1.75 ms per step), which the fourth wave on two of the SIMDs does not
// win back — pipelined roll-out 164.4 against 155.3 ms, bench line 1.416 against 1.471 M env-steps/s (profiles/r05_k2_residency_ab.txt).
#ifndef TMW_WAVES_PER_SIMD
#define TMW_WAVES_PER_SIMD 3      // (-DTMW_WAVES_PER_SIMD=4: the 14-envs-per-CU build, for A/B runs)
#endif
template <bool STATIC>
__global__ __launch_bounds__(64, TMW_WAVES_PER_SIMD) void k_physics_wave(const DModel *__restrict__ mp, float *st, const float *action, int nsub,
                                                     int do_euler, float *ws_dump, int n, int e0, int rs, float *spill, int spill_stride) {
  extern __shared__ float tmw_lds[];
  WCtx c{(TmwModel *)mp, tmw_lds, st, n, (int)blockIdx.x + e0, (int)threadIdx.x, nullptr, 0ull, nullptr};
  c.rs = rs;
  c.mspill = spill ? spill + 64 + (size_t)(blockIdx.x + e0) * (size_t)spill_stride : nullptr;
  c.action = action;
#ifndef TMW_PROFILE
  c.dump = ws_dump;
#endif
#ifdef TMW_PROFILE
  if (ws_dump) { c.prof = (unsigned long long *)ws_dump + (size_t)(blockIdx.x + e0) * 40; c.tlast = __builtin_amdgcn_s_memtime(); }
#endif
  constexpr WLayout ks(TMW_RODENT_DIMS, 1);
  const WLayout kd = STATIC ? ks : WLayout(mp->nbody, mp->njnt, mp->nq, mp->nv, mp->nu, mp->ncon, mp->nlim, mp->nnz, mp->ngroup,
                                           mp->nround_body, mp->nround_dof);
  const WLayout &K = STATIC ? ks : kd;
  float time = tmw_load_state(c, K, action);
  for (int f = 0; f < nsub; f++) {
    // fresh, opaque copies of the lane id and the model pointer per substep: LICM otherwise hoists every lane-derived LDS /
    // global address of the substep body (cheap adds) out of this loop, and the register allocator then SPILLS them
    // (the lane id is RE-MADE from the hardware's lane count — one wave per workgroup: mbcnt of a full mask = threadIdx.x — instead of laundering
    // threadIdx.x, which kept the launch's own copy of it alive in a register across the whole substep body: with 128 registers for four waves per
    // SIMD that copy was the one value the allocator spilled to scratch)
#ifdef TMW_LANE_FROM_TID
    { int l = threadIdx.x; asm volatile("" : "+v"(l)); c.lane = l;
#else
    { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); c.lane = l;
#endif
      TmwModel *q = (TmwModel *)mp; asm volatile("" : "+s"(q)); c.mp = q; }
    tmw_forward(c, K, f == nsub - 1);
    if (do_euler) time = tmw_euler(c, K, time);
  }
#ifndef TMW_PROFILE
  if (ws_dump) tmw_dump(c, K, ws_dump);
#endif
  tmw_store_state(c, K, time);
  TMW_TICK(12);
}

extern "C" void tmjx_internal_launch_physics_wave(int rodent, int cnt, size_t lds, hipStream_t stream, const DModel *mp, float *st, const float *action, int nsub,
                                                  int do_euler, float *ws_dump, int n, int e0, int rs, float *spill, int spill_stride) {
  if (rodent) hipLaunchKernelGGL(k_physics_wave<true>, dim3(cnt), dim3(64), lds, stream, mp, st, action, nsub, do_euler, ws_dump, n, e0, rs, spill, spill_stride);
  else hipLaunchKernelGGL(k_physics_wave<false>, dim3(cnt), dim3(64), lds, stream, mp, st, action, nsub, do_euler, ws_dump, n, e0, rs, spill, spill_stride);
}
