// csrc/silu_math.h — sigmoid / SiLU as the learner's epilogues evaluate them (flax nn.silu in track_mjx/agent/mlp_ppo/intention_network.py:38,75 and
// brax's swish value MLP, ppo_networks.py:180-184): v_exp_f32 + v_rcp_f32 — two transcendental instructions, each good to 1 ulp, ~1e-6 relative on the
// sigmoid over |v| <= 20 — instead of expf + an IEEE division (~22 vector instructions per element).  Measured with in-kernel stamps (round 6): the
// epilogues of the GEMM / chain kernels were bound by exactly those instructions (a Dense -> SiLU epilogue of an 80 x 256 tile: 14 k cycles with
// the matrix pipe idle, 40 elements per thread), not by their stores.  ONE definition for every kernel that evaluates the activation — fused and
// unfused forms, forward and both backward passes, fp32 and bf16 families — so that they keep agreeing with each other to the bit.
#pragma once
__device__ __forceinline__ float tm_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
// (the product is kept out of floating-point contraction: fused into a following addition — a row sum — it would round differently from kernel to kernel,
// and the fused and unfused forms of a block would stop agreeing to the bit)
__device__ __forceinline__ float tm_silu(float v) {
#pragma clang fp contract(off)
  return v * tm_sigmoid(v);
}
