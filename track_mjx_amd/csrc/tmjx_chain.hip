// csrc/tmjx_chain.hip — fourth translation unit of libtmjx_hip.so: the whole-MLP-chain kernels (csrc/mlp_chain.h) and their C-ABI entry points
// (include/tmjx.h "whole-chain kernels").  Compiled next to the other units (track_mjx_amd/hip.py:build) and linked into the same library.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <string>

#include "../../include/tmjx.h"
#include "mlp_chain.h"

extern "C" int tmjx_internal_fail(int code, const char *msg);       // tmjx_hip.hip: records the calling thread's error message
extern "C" int tmjx_internal_gemm_mt(int M, int col_tiles);         // tmjx_hip.hip: the row tile (80 / 32 rows) the layer-by-layer kernels take
static int fail(int code, const std::string &msg) { return tmjx_internal_fail(code, msg.c_str()); }
static int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TMJX_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  return TMJX_OK;
}
static bool al16(const void *p) { return !((uintptr_t)p & 15); }
static bool rows16(const void *p, long long ld) { return al16(p) && !(ld & 3); }

template <int MT, int EPI, int FIN, bool LAT = false>
static int launch_chain_fwd(const ChainFwd &P, hipStream_t s) {
  constexpr size_t lds = sizeof(float) * (size_t)ChainLds<MT>::TOTAL;
  static bool attr_set = false;            // > 64 KiB of dynamic LDS needs the attribute once per kernel
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_chain_fwd<MT, EPI, FIN, LAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_chain_fwd): ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_chain_fwd<MT, EPI, FIN, LAT>), dim3((P.M + 16 * MT - 1) / (16 * MT)), dim3(CH_NT), lds, s, P);
  return check_launch("k_chain_fwd");
}
template <int EPI, int FIN, bool LAT = false>
static int chain_fwd_mt(const ChainFwd &P, hipStream_t s) {
  if (tmjx_internal_gemm_mt(P.M, 1) == 2) return launch_chain_fwd<2, EPI, FIN, LAT>(P, s);
  return launch_chain_fwd<5, EPI, FIN, LAT>(P, s);
}

static int chain_rows(int M) { const int bm = 16 * tmjx_internal_gemm_mt(M, 1); return (M + bm - 1) / bm * bm; }
static const char *chain_fwd_why(const tmjx_chain_fwd_t *c) {
  if (!c || !c->A) return "null argument";
  if (c->M >= 1 && c->rows_alloc < chain_rows(c->M)) return "rows_alloc: the y buffers must hold tmjx_chain_rows(M) rows (whole row tiles are stored)";
  if (c->epi != 1 && c->epi != 3) return "epi must be 1 (Dense -> SiLU -> LayerNorm blocks) or 3 (Dense -> SiLU layers)";
  if (c->n_hidden < 1 || c->n_hidden > TMJX_CHAIN_MAX_HIDDEN) return "1 .. 4 hidden layers";
  if (c->M < 1 || !rows16(c->A, c->lda)) return "M >= 1 and 16-byte aligned input rows";
  for (int l = 0; l < c->n_hidden; l++) {
    const tmjx_chain_layer_t &h = c->hidden[l];
    if (!h.W || !h.bias || !h.z || !h.y || (c->epi == 1 && (!h.gamma || !h.beta || !h.stats))) return "null layer argument";
    if (h.K < 1 || h.ldw < h.K || !rows16(h.W, h.ldw) || !al16(h.bias) || !al16(h.z) || !al16(h.y)) return "layer operands must be 16-byte aligned with ldw >= K";
    if (c->epi == 1 && (!al16(h.gamma) || !al16(h.beta))) return "gamma / beta must be 16-byte aligned";
    if (l > 0 && h.K != 256) return "every hidden layer is 256 wide (layers behind the first have K = 256)";
    if (l == 0 && c->lda < h.K) return "lda < K of the first layer";
  }
  if (c->Wf) {
    if (!c->outf || c->Nf < 1 || c->Nf > 128) return "the last layer has 1 .. 128 columns";
    if (c->Nf == 1) { if (!al16(c->Wf)) return "the head's weight row must be 16-byte aligned"; }
    else if (c->ldwf < 256 || !rows16(c->Wf, c->ldwf) || c->ldof < c->Nf) return "the last layer's weight rows must be 16-byte aligned, ldwf >= 256, ldof >= Nf";
  }
  if (c->lat_out) {
    if (c->epi != 1 || !c->Wf || c->Nf != 2 * c->lat_Z || !c->lat_eps || !c->prop) return "the latent tail follows a LayerNorm-block chain whose last layer is [mean | logvar] (Nf = 2 lat_Z)";
    if (c->lat_Z < 4 || (c->lat_Z & 3) || c->prop_w < 0 || (c->prop_w & 1) || c->lat_ld < c->lat_Z + c->prop_w || (c->lat_ld & 3) || (c->prop_ld & 1) || c->prop_ld < c->prop_w)
      return "latent tail: lat_Z % 4 == 0, prop_w and prop_ld even, lat_ld % 4 == 0 and >= lat_Z + prop_w";
    if (!al16(c->lat_eps) || !al16(c->lat_out) || ((uintptr_t)c->prop & 7)) return "latent tail: lat_eps / lat_out 16-byte aligned, prop 8-byte aligned";
  }
  return nullptr;
}

template <int MT, int EPI, bool HEAD, bool DX>
static int launch_chain_bwd(const ChainBwd &P, hipStream_t s) {
  constexpr size_t lds = sizeof(float) * (size_t)ChainLds<MT>::TOTAL;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_chain_bwd<MT, EPI, HEAD, DX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_chain_bwd): ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_chain_bwd<MT, EPI, HEAD, DX>), dim3((P.M + 16 * MT - 1) / (16 * MT)), dim3(CH_NT), lds, s, P);
  return check_launch("k_chain_bwd");
}
template <int EPI, bool HEAD, bool DX>
static int chain_bwd_mt(const ChainBwd &P, hipStream_t s) {
  if (tmjx_internal_gemm_mt(P.M, 1) == 2) return launch_chain_bwd<2, EPI, HEAD, DX>(P, s);
  return launch_chain_bwd<5, EPI, HEAD, DX>(P, s);
}

static const char *chain_bwd_why(const tmjx_chain_bwd_t *c) {
  if (!c || !c->G) return "null argument";
  if (c->M >= 1 && c->rows_alloc < chain_rows(c->M)) return "rows_alloc: the dz buffers must hold tmjx_chain_rows(M) rows (whole row tiles are stored)";
  if (c->epi != 2 && c->epi != 4) return "epi must be 2 (Dense -> SiLU -> LayerNorm blocks) or 4 (Dense -> SiLU layers)";
  if (c->n_stages < 1 || c->n_stages > TMJX_CHAIN_MAX_HIDDEN) return "1 .. 4 stages (one per hidden layer)";
  if (c->M < 1 || c->Kg < 1 || c->Kg > 128) return "M >= 1 and a last layer of 1 .. 128 columns";
  const bool head = c->Kg == 1;
  if (head && c->epi != 4) return "a 1-wide last layer follows Dense -> SiLU layers (epi 4) only";
  if (!head && (c->ldg < c->Kg || (c->Kg & 3) || !rows16(c->G, c->ldg))) return "the output gradient's rows must be 16-byte aligned with Kg % 4 == 0";
  for (int l = 0; l < c->n_stages; l++) {
    const tmjx_chain_bwd_stage_t &s = c->stage[l];
    if (!s.W || !s.z || !s.bias || !s.dz || (c->epi == 2 && (!s.gamma || !s.stats || !s.partial))) return "null stage argument";
    if (!al16(s.W) || !al16(s.z) || !al16(s.bias) || !al16(s.dz) || (c->epi == 2 && (!al16(s.gamma) || !al16(s.partial)))) return "stage operands must be 16-byte aligned";
    if (!(head && l == 0) && (s.ldw < 256 || (s.ldw & 3))) return "weight rows must be 16-byte aligned with ldw >= 256";
  }
  if (c->W0) {
    if (head) return "no trailing input gradient behind a 1-wide head's chain";
    if (!c->dx || c->dx_cols < 1 || c->dx_cols > 128 || c->lddx < c->dx_cols || !rows16(c->W0, c->ldw0) || c->ldw0 < ((c->dx_cols + 3) & ~3)) return "the trailing input gradient has 1 .. 128 columns, W0 rows 16-byte aligned";
  }
  return nullptr;
}

extern "C" {
int tmjx_chain_rows(int M) { return M < 1 ? 0 : chain_rows(M); }
int tmjx_chain_bwd_ok(const tmjx_chain_bwd_t *c) { return chain_bwd_why(c) == nullptr; }
int tmjx_chain_bwd(const tmjx_chain_bwd_t *c, void *stream) {
  if (const char *why = chain_bwd_why(c)) return fail(TMJX_EINVAL, std::string("tmjx_chain_bwd: ") + why);
  ChainBwd P{};
  P.G = c->G; P.ldg = c->ldg; P.Kg = c->Kg; P.M = c->M; P.ns = c->n_stages;
  for (int l = 0; l < c->n_stages; l++) {
    const tmjx_chain_bwd_stage_t &s = c->stage[l];
    P.s[l] = ChainBwdStage{s.W, s.ldw, s.z, s.bias, s.gamma, s.stats, s.dz, s.partial};
  }
  P.W0 = c->W0; P.ldw0 = c->ldw0; P.dx_cols = c->dx_cols; P.dx = c->dx; P.lddx = c->lddx;
  P.prof = (unsigned long long *)c->prof;
  hipStream_t s = (hipStream_t)stream;
  if (c->epi == 2) return c->W0 ? chain_bwd_mt<2, false, true>(P, s) : chain_bwd_mt<2, false, false>(P, s);
  if (c->Kg == 1) return chain_bwd_mt<4, true, false>(P, s);
  return c->W0 ? chain_bwd_mt<4, false, true>(P, s) : chain_bwd_mt<4, false, false>(P, s);
}
int tmjx_chain_fwd_ok(const tmjx_chain_fwd_t *c) { return chain_fwd_why(c) == nullptr; }
int tmjx_chain_fwd(const tmjx_chain_fwd_t *c, void *stream) {
  if (const char *why = chain_fwd_why(c)) return fail(TMJX_EINVAL, std::string("tmjx_chain_fwd: ") + why);
  ChainFwd P{};
  P.A = c->A; P.lda = c->lda; P.M = c->M; P.nh = c->n_hidden; P.eps = c->eps;
  for (int l = 0; l < c->n_hidden; l++) {
    const tmjx_chain_layer_t &h = c->hidden[l];
    P.h[l] = ChainHidden{h.W, h.bias, h.gamma, h.beta, h.z, h.y, h.stats, h.K, h.ldw};
  }
  P.Wf = c->Wf; P.bf = c->bf; P.outf = c->outf; P.Nf = c->Nf; P.ldwf = c->ldwf; P.ldof = c->ldof;
  P.prof = (unsigned long long *)c->prof;
  P.lat_eps = c->lat_eps; P.lat_out = c->lat_out; P.prop = c->prop; P.lat_Z = c->lat_Z; P.lat_ld = c->lat_ld; P.prop_w = c->prop_w; P.prop_ld = c->prop_ld;
  hipStream_t s = (hipStream_t)stream;
  const int fin = !c->Wf ? 0 : (c->Nf == 1 ? 2 : 1);
  if (c->epi == 1) {
    if (fin == 0) return chain_fwd_mt<1, 0>(P, s);
    if (fin == 1 && c->lat_out) return chain_fwd_mt<1, 1, true>(P, s);
    if (fin == 1) return chain_fwd_mt<1, 1>(P, s);
    return fail(TMJX_EINVAL, "tmjx_chain_fwd: a 1-wide last layer follows Dense -> SiLU layers (epi 3) only");
  }
  if (fin == 0) return chain_fwd_mt<3, 0>(P, s);
  if (fin == 2) return chain_fwd_mt<3, 2>(P, s);
  return chain_fwd_mt<3, 1>(P, s);
}
}  // extern "C"
