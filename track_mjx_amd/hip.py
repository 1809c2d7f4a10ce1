"""ctypes binding of libtmjx_hip.so (the C-ABI declared in include/tmjx.h).

The HIP library is the product: there is no CPU fallback.  Importing this module never fails
(so host-only tooling keeps working), but `lib()` raises if the shared object is missing or does
not load, and every entry point raises `TmjxError` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
SO_PATH = Path(os.environ.get("TMJX_SO", str(_PKG / "libtmjx_hip.so")))  # TMJX_SO: alternative build (profiling)
CSRC = _PKG / "csrc"


class TmjxError(RuntimeError):
    pass


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nq", "nv", "nu", "nbody", "ncon", "nefc", "obs_size", "ref_obs_size", "n_metrics", "window",
        "qpos", "qvel", "act", "qacc_warmstart", "time", "xpos", "xmat_torso", "qfrc_actuator",
        "prev_ctrl", "action_buffer", "done", "steps_f", "first_phys", "first_obs", "first_prev_ctrl", "state_rows",
        "i_clip_idx", "i_start_frame", "i_buffer_index", "i_nan_count", "istate_rows", "ws_rows")]


EXPORTS = ("tmjx_model_create", "tmjx_model_destroy", "tmjx_layout", "tmjx_clips_upload", "tmjx_reset", "tmjx_step",
           "tmjx_physics", "tmjx_physics_step", "tmjx_forward", "tmjx_reward_obs", "tmjx_gae", "tmjx_ppo_scratch_floats", "tmjx_ppo_loss",
           "tmjx_silu_ln_partial_floats", "tmjx_silu_ln_fwd", "tmjx_silu_ln_bwd", "tmjx_gather_normalize", "tmjx_latent_concat", "tmjx_latent_concat_bwd", "tmjx_sample_action", "tmjx_linear_nolds", "tmjx_adam_clip", "tmjx_colsum_scratch_floats", "tmjx_colsum",
           "tmjx_gemm_nt", "tmjx_gemm_nn", "tmjx_gemm_dw", "tmjx_gemm_dw_scratch_floats", "tmjx_set_wrappers", "tmjx_stats_scratch_floats", "tmjx_stats_sums", "tmjx_stats_apply",
           "tmjx_debug_rows", "tmjx_last_error", "tmjx_version")


class PpoCfg(C.Structure):
    """tmjx_ppo_cfg_t (include/tmjx.h)."""
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("A", C.c_int32), ("Z", C.c_int32), ("reward_scaling", C.c_float),
                ("discounting", C.c_float), ("gae_lambda", C.c_float), ("clip_eps", C.c_float), ("entropy_cost", C.c_float),
                ("kl_weight", C.c_float), ("normalize_advantage", C.c_int32)]

_lib = None


def build(verbose: bool = False, out: Path | None = None, defines: tuple = ()) -> Path:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU).  `out` / `defines`: alternative builds for
    tests (-DTMJX_LANE_IMPL: with the lane-per-env cross-check kernels) and profiling (-DTMW_PROFILE)."""
    # -fno-slp-vectorize: the SLP vectoriser packs the two dof slots of the row products into v_pk_* with more v_mov shuffles than
    # it saves (measured 1.8 % on the physics kernel); the explicitly packed FMAs of the chain kernels are not affected
    out = SO_PATH if out is None else Path(out)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC", "-Wno-unused-value",
           *[f"-D{d}" for d in defines], "-o", str(out), str(CSRC / "tmjx_hip.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise TmjxError("hipcc failed:\n" + res.stderr[-4000:])
    if verbose:
        print(" ".join(cmd))
    return out


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _lib = load(SO_PATH)
    return _lib


def load(path: Path):
    """dlopen a build of the library and declare its entry points (lib() = the product build; tests load the lane cross-check build)."""
    path = Path(path)
    if not path.exists():
        raise TmjxError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
    try:
        L = C.CDLL(str(path))
    except OSError as e:  # e.g. no ROCm runtime on this machine
        raise TmjxError(f"cannot load {path}: {e}") from e
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p
    L.tmjx_model_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.tmjx_model_destroy.argtypes = [vp]
    L.tmjx_model_destroy.restype = None
    L.tmjx_layout.argtypes = [vp, C.POINTER(Layout)]
    L.tmjx_clips_upload.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int]
    L.tmjx_reset.argtypes = [vp, fp, vp, vp, vp, fp, fp, fp, fp, C.c_int, vp]
    L.tmjx_step.argtypes = [vp, fp, vp, fp, fp, fp, fp, fp, fp, fp, C.c_int, vp]
    L.tmjx_physics.argtypes = [vp, fp, fp, C.c_int, fp, C.c_int, vp]
    L.tmjx_physics_step.argtypes = [vp, fp, fp, fp, C.c_int, vp]
    L.tmjx_forward.argtypes = [vp, fp, fp, C.c_int, vp]
    L.tmjx_reward_obs.argtypes = [vp, fp, vp, fp, fp, fp, fp, fp, fp, fp, C.c_int, vp]
    L.tmjx_gae.argtypes = [fp, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, C.c_int, C.c_int, vp]
    L.tmjx_ppo_scratch_floats.argtypes = [C.c_int, C.c_int]
    L.tmjx_ppo_loss.argtypes = [C.POINTER(PpoCfg)] + [fp] * 15 + [vp]
    L.tmjx_silu_ln_partial_floats.argtypes = [C.c_int, C.c_int]
    L.tmjx_silu_ln_fwd.argtypes = [fp] * 6 + [C.c_int, C.c_int, C.c_float, vp]
    L.tmjx_silu_ln_bwd.argtypes = [fp] * 8 + [C.c_int, C.c_int, vp]
    L.tmjx_gather_normalize.argtypes = [fp] * 5 + [C.c_int] * 4 + [vp]
    L.tmjx_latent_concat.argtypes = [fp] * 4 + [C.c_int] * 4 + [C.c_int64, C.c_int64, fp, fp, C.c_int, vp]
    L.tmjx_latent_concat_bwd.argtypes = [fp] * 4 + [C.c_int] * 3 + [vp]
    L.tmjx_sample_action.argtypes = [fp] * 5 + [C.c_int, C.c_int, vp]
    L.tmjx_linear_nolds.argtypes = [fp, C.c_int64, C.c_int64, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]
    L.tmjx_colsum_scratch_floats.argtypes = [C.c_int]
    L.tmjx_colsum.argtypes = [fp, fp, fp, C.c_int, C.c_int, vp]
    L.tmjx_adam_clip.argtypes = [fp] * 5 + [C.c_longlong] + [C.c_float] * 7 + [vp]
    L.tmjx_gemm_nt.argtypes = [fp, C.c_int, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.tmjx_gemm_nn.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.tmjx_gemm_dw_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    L.tmjx_gemm_dw_scratch_floats.restype = C.c_longlong
    L.tmjx_gemm_dw.argtypes = [fp, C.c_int, fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]
    L.tmjx_set_wrappers.argtypes = [vp, C.c_int, C.c_int]
    L.tmjx_stats_scratch_floats.argtypes = [C.c_int]
    L.tmjx_stats_sums.argtypes = [fp, fp, fp, fp, C.c_longlong, C.c_int, vp]
    L.tmjx_stats_apply.argtypes = [fp, C.c_float, fp, fp, fp, fp, C.c_int, C.c_float, C.c_float, vp]
    L.tmjx_debug_rows.argtypes = [vp, C.c_char_p, ip, ip]
    L.tmjx_last_error.restype = C.c_char_p
    L.tmjx_version.restype = C.c_char_p
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise TmjxError(f"{what} failed ({rc}): {lib().tmjx_last_error().decode()}")


def block_override() -> str | None:
    return os.environ.get("TMJX_BLOCK")
