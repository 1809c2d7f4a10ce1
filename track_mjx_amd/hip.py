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
SOURCES = (CSRC / "tmjx_hip.hip", CSRC / "tmjx_bf16.hip", CSRC / "tmjx_wave.hip", CSRC / "tmjx_chain.hip")
# per-source compiler flags: the physics kernel's unit is built without machine LICM (csrc/tmjx_wave.hip says why)
SOURCE_FLAGS = {"tmjx_wave.hip": ("-mllvm", "-disable-machine-licm")}


class TmjxError(RuntimeError):
    pass


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nq", "nv", "nu", "nbody", "ncon", "nefc", "obs_size", "ref_obs_size", "n_metrics", "window",
        "qpos", "qvel", "act", "qacc_warmstart", "time", "xpos", "xmat_torso", "qfrc_actuator",
        "prev_ctrl", "action_buffer", "done", "steps_f", "first_phys", "first_obs", "first_prev_ctrl", "state_rows",
        "i_clip_idx", "i_start_frame", "i_buffer_index", "i_nan_count", "istate_rows", "ws_rows")]


EXPORTS = ("tmjx_model_create", "tmjx_model_destroy", "tmjx_layout", "tmjx_clips_upload", "tmjx_reset", "tmjx_step",
           "tmjx_physics", "tmjx_physics_step", "tmjx_forward", "tmjx_reward_obs", "tmjx_reward_frame", "tmjx_gae", "tmjx_ppo_scratch_floats", "tmjx_ppo_loss", "tmjx_ppo_loss_phases",
           "tmjx_silu_ln_partial_floats", "tmjx_silu_ln_fwd", "tmjx_silu_ln_bwd", "tmjx_silu_ln_fwd_bf16", "tmjx_silu_ln_bwd_bf16", "tmjx_minibatch_begin_bf16", "tmjx_gather_normalize", "tmjx_latent_concat", "tmjx_latent_concat_bwd", "tmjx_latent_concat_bwd_add", "tmjx_sample_action", "tmjx_linear_nolds", "tmjx_linear_act", "tmjx_linear_act_ok", "tmjx_linear_nolds_norm", "tmjx_linear_nolds_bf16", "tmjx_adam_clip", "tmjx_adam_clip_norm", "tmjx_adam_norm_floats", "tmjx_colsum_scratch_floats", "tmjx_colsum",
           "tmjx_gather_minibatch", "tmjx_minibatch_begin", "tmjx_philox4x32_10", "tmjx_gemm_nt", "tmjx_gemm_nt_silu_ln", "tmjx_gemm_nt_silu_ln_ok", "tmjx_gemm_nn", "tmjx_colsum_grouped", "tmjx_gemm_nn_ln_bwd", "tmjx_gemm_nn_ln_bwd_ok", "tmjx_gemm_nn_ln_bwd_partial_floats", "tmjx_gemm_dw", "tmjx_gemm_dw_grouped", "tmjx_gemm_dw_grouped_wgs", "tmjx_gemm_dw_scratch_floats", "tmjx_set_wrappers", "tmjx_set_action_repeat", "tmjx_stats_scratch_floats", "tmjx_stats_sums", "tmjx_stats_apply",
           "tmjx_rollout_store", "tmjx_clips_share", "tmjx_gemm_nt_silu", "tmjx_gemm_nt_silu_ok", "tmjx_silu_fwd", "tmjx_silu_bwd", "tmjx_gemm_nn_silu_bwd_ok", "tmjx_gemm_nn_silu_bwd", "tmjx_silu_bwd_rank1", "tmjx_head_dw_scratch_floats", "tmjx_head_dw", "tmjx_head_fwd_ok", "tmjx_head_fwd", "tmjx_bf16_shadow", "tmjx_bgemm_nt", "tmjx_bgemm_dw", "tmjx_bgemm_dw_grouped", "tmjx_bgemm_dw_scratch_floats", "tmjx_bgemm_row_tile_ok", "tmjx_bf16_z_bytes", "tmjx_bgemm_partial_floats",
           "tmjx_bgemm_ln_fwd", "tmjx_bgemm_ln_bwd", "tmjx_bgemm_silu_fwd", "tmjx_bgemm_silu_bwd", "tmjx_bf_silu_bwd", "tmjx_bf_silu_bwd_rank1",
           "tmjx_chain_rows", "tmjx_chain_fwd_ok", "tmjx_chain_fwd", "tmjx_chain_bwd_ok", "tmjx_chain_bwd",
           "tmjx_debug_rows", "tmjx_last_error", "tmjx_version")


class Minibatch(C.Structure):
    """tmjx_minibatch_t (include/tmjx.h)."""
    _fields_ = ([(k, C.c_void_p) for k in ("obs", "next_last", "raw_action", "log_prob", "reward", "discount", "truncation", "perm", "mean", "std", "obs_n", "next_n",
                                           "raw_action_g", "scalars_g", "eps", "noise", "state")] + [("seed", C.c_uint64)] +
                [(k, C.c_int32) for k in ("T", "R", "B", "W", "A", "Z", "advance")])


class ColsumProblem(C.Structure):
    """tmjx_colsum_problem_t (include/tmjx.h)."""
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int32), ("width", C.c_int32)]


class DwProblem(C.Structure):
    """tmjx_dw_problem_t (include/tmjx.h)."""
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("scratch", C.c_void_p),
                ("ldy", C.c_int32), ("ldx", C.c_int32), ("lddw", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32)]


class BdwProblem(C.Structure):
    """tmjx_bdw_problem_t (include/tmjx.h)."""
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("scratch", C.c_void_p),
                ("y_is_f32", C.c_int32), ("x_is_f32", C.c_int32), ("ldy", C.c_int32), ("ldx", C.c_int32), ("lddw", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32)]


class RolloutStore(C.Structure):
    """tmjx_rollout_store_t (include/tmjx.h)."""
    _fields_ = [(k, C.c_void_p) for k in ("obs", "obs_dst0", "obs_dst1", "raw", "raw_dst", "logp", "logp_dst", "reward", "reward_dst", "done", "discount_dst",
                                          "trunc", "trunc_dst")] + [(k, C.c_int32) for k in ("n", "W", "A")] + [("obs_dst2", C.c_void_p)]


class Bf16Shadow(C.Structure):
    """tmjx_bf16_shadow_t (include/tmjx.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32), ("ld_src", C.c_int32),
                ("ld_dst", C.c_int32), ("ld_dst_t", C.c_int32)]


class ChainLayer(C.Structure):
    """tmjx_chain_layer_t (include/tmjx.h)."""
    _fields_ = [(k, C.c_void_p) for k in ("W", "bias", "gamma", "beta", "z", "y", "stats")] + [("K", C.c_int32), ("ldw", C.c_int32)]


class ChainFwd(C.Structure):
    """tmjx_chain_fwd_t (include/tmjx.h)."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int32), ("M", C.c_int32), ("n_hidden", C.c_int32), ("epi", C.c_int32), ("hidden", ChainLayer * 4),
                ("Wf", C.c_void_p), ("bf", C.c_void_p), ("outf", C.c_void_p), ("Nf", C.c_int32), ("ldwf", C.c_int32), ("ldof", C.c_int32), ("eps", C.c_float), ("rows_alloc", C.c_int32),
                ("lat_eps", C.c_void_p), ("lat_out", C.c_void_p), ("prop", C.c_void_p), ("lat_Z", C.c_int32), ("lat_ld", C.c_int32), ("prop_w", C.c_int32), ("prop_ld", C.c_int32),
                ("prof", C.c_void_p)]


class ChainBwdStage(C.Structure):
    """tmjx_chain_bwd_stage_t (include/tmjx.h)."""
    _fields_ = [("W", C.c_void_p), ("ldw", C.c_int32)] + [(k, C.c_void_p) for k in ("z", "bias", "gamma", "stats", "dz", "partial")]


class ChainBwd(C.Structure):
    """tmjx_chain_bwd_t (include/tmjx.h)."""
    _fields_ = [("G", C.c_void_p), ("ldg", C.c_int32), ("Kg", C.c_int32), ("M", C.c_int32), ("n_stages", C.c_int32), ("epi", C.c_int32), ("stage", ChainBwdStage * 4),
                ("W0", C.c_void_p), ("ldw0", C.c_int32), ("dx_cols", C.c_int32), ("dx", C.c_void_p), ("lddx", C.c_int32), ("rows_alloc", C.c_int32), ("prof", C.c_void_p)]


class PpoCfg(C.Structure):
    """tmjx_ppo_cfg_t (include/tmjx.h)."""
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("A", C.c_int32), ("Z", C.c_int32), ("reward_scaling", C.c_float),
                ("discounting", C.c_float), ("gae_lambda", C.c_float), ("clip_eps", C.c_float), ("entropy_cost", C.c_float),
                ("kl_weight", C.c_float), ("normalize_advantage", C.c_int32), ("accumulate", C.c_int32)]

_lib = None


def build(verbose: bool = False, out: Path | None = None, defines: tuple = ()) -> Path:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU).  `out` / `defines`: alternative builds for
    tests (-DTMJX_LANE_IMPL: with the lane-per-env cross-check kernels) and profiling (-DTMW_PROFILE)."""
    # -fno-slp-vectorize: the SLP vectoriser packs the two dof slots of the row products into v_pk_* with more v_mov shuffles than
    # it saves (measured 1.8 % on the physics kernel); the explicitly packed FMAs of the chain kernels are not affected
    out = SO_PATH if out is None else Path(out)
    # three translation units compiled side by side (each hipcc run is single-threaded per offload arch), then linked into ONE library.
    # Objects are cached next to the library, keyed by a hash of the flags and of the source with every header it includes (recursively):
    # editing one kernel family recompiles one unit
    import hashlib
    import re
    from concurrent.futures import ThreadPoolExecutor
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-Wno-unused-value", *[f"-D{d}" for d in defines],
             *os.environ.get("TMJX_EXTRA_HIPCC_FLAGS", "").split()]          # (tuning experiments: A/B builds with extra compiler flags)

    def closure(path: Path, seen: dict) -> dict:
        if path in seen or not path.exists():
            return seen
        text = path.read_text()
        seen[path] = text
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, re.M):
            closure((path.parent / inc).resolve(), seen)
        return seen

    cache = out.parent / ".build"
    cache.mkdir(exist_ok=True)
    objs = []
    def flags_of(src):
        return [*flags, *SOURCE_FLAGS.get(src.name, ())]

    for src in SOURCES:
        h = hashlib.sha256(" ".join(flags_of(src)).encode())
        for pth, text in sorted(closure(src.resolve(), {}).items()):
            h.update(str(pth.name).encode()); h.update(text.encode())
        objs.append(cache / f"{src.stem}.{h.hexdigest()[:16]}.o")

    def compile_one(src, obj):
        if obj.exists():
            return
        for stale in cache.glob(f"{src.stem}.*.o"):
            if len(list(cache.glob(f"{src.stem}.*.o"))) > 6:
                stale.unlink(missing_ok=True)
        tmp = obj.with_suffix(f".tmp{os.getpid()}.o")
        cmd = ["hipcc", *flags_of(src), "-c", "-o", str(tmp), str(src)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise TmjxError(f"hipcc failed on {src.name}:\n" + res.stderr[-4000:])
        tmp.replace(obj)
        if verbose:
            print(" ".join(cmd))
    with ThreadPoolExecutor(len(SOURCES)) as ex:
        for f in [ex.submit(compile_one, s, o) for s, o in zip(SOURCES, objs)]:
            f.result()
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *[str(o) for o in objs]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise TmjxError("hipcc link failed:\n" + res.stderr[-4000:])
    # the build id: a hash of WHAT was compiled (flags + every source with its include closure, the object cache keys), written next to the
    # library.  hipcc's objects are not bit-reproducible, so a hash of the .so would change with every recompilation of unchanged sources —
    # and the counters under profiles/ are tied to a build by this id (tools/buildid.py, bench.py)
    Path(str(out) + ".id").write_text(hashlib.sha256(" ".join(o.name for o in objs).encode()).hexdigest()[:16] + "\n")
    return out


def build_id(path: Path | None = None) -> str:
    """Id of a built library: the source-derived id `build` wrote next to it, else (a library from elsewhere) the hash of its bytes."""
    import hashlib
    path = Path(SO_PATH if path is None else path)
    side = Path(str(path) + ".id")
    try:
        if side.exists() and side.stat().st_mtime >= path.stat().st_mtime - 1:
            return side.read_text().strip()
        return hashlib.sha256(path.read_bytes()).hexdigest()[:16]
    except OSError:
        return "missing"


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
    sig: dict = {}      # entry point -> (argtypes, restype or None for the default int return code)
    sig.setdefault("tmjx_model_create", [None, None])[0] = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    sig.setdefault("tmjx_model_destroy", [None, None])[0] = [vp]
    sig.setdefault("tmjx_model_destroy", [None, None])[1] = None
    sig.setdefault("tmjx_layout", [None, None])[0] = [vp, C.POINTER(Layout)]
    sig.setdefault("tmjx_clips_upload", [None, None])[0] = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int]
    sig.setdefault("tmjx_reset", [None, None])[0] = [vp, fp, vp, vp, vp, fp, fp, fp, fp, C.c_int, vp]
    sig.setdefault("tmjx_step", [None, None])[0] = [vp, fp, vp, fp, fp, fp, fp, fp, fp, fp, C.c_int, vp]
    sig.setdefault("tmjx_physics", [None, None])[0] = [vp, fp, fp, C.c_int, fp, C.c_int, vp]
    sig.setdefault("tmjx_physics_step", [None, None])[0] = [vp, fp, fp, fp, C.c_int, vp]
    sig.setdefault("tmjx_forward", [None, None])[0] = [vp, fp, fp, C.c_int, vp]
    sig.setdefault("tmjx_reward_obs", [None, None])[0] = [vp, fp, vp, fp, fp, fp, fp, fp, fp, fp, C.c_int, vp]
    sig.setdefault("tmjx_reward_frame", [None, None])[0] = [vp, fp, vp, fp] + [fp] * 5 + [fp] * 5 + [C.c_int, vp]
    sig.setdefault("tmjx_gae", [None, None])[0] = [fp, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_ppo_scratch_floats", [None, None])[0] = [C.c_int, C.c_int]
    sig.setdefault("tmjx_ppo_loss", [None, None])[0] = [C.POINTER(PpoCfg)] + [fp] * 15 + [vp]
    sig.setdefault("tmjx_ppo_loss_phases", [None, None])[0] = [C.POINTER(PpoCfg)] + [fp] * 15 + [C.c_int, vp]
    sig.setdefault("tmjx_silu_ln_partial_floats", [None, None])[0] = [C.c_int, C.c_int]
    sig.setdefault("tmjx_silu_ln_fwd", [None, None])[0] = [fp] * 6 + [C.c_int, C.c_int, C.c_float, vp]
    sig.setdefault("tmjx_silu_ln_bwd", [None, None])[0] = [fp] * 8 + [C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_silu_ln_fwd_bf16", [None, None])[0] = [fp] * 5 + [C.c_int, fp, C.c_int, C.c_int, C.c_float, vp]
    sig.setdefault("tmjx_silu_ln_bwd_bf16", [None, None])[0] = [fp] * 6 + [C.c_int, fp, fp, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_gather_normalize", [None, None])[0] = [fp] * 5 + [C.c_int] * 4 + [vp]
    sig.setdefault("tmjx_gather_minibatch", [None, None])[0] = [fp] * 14 + [C.c_int] * 5 + [vp]
    sig.setdefault("tmjx_minibatch_begin", [None, None])[0] = [C.POINTER(Minibatch), vp]
    sig.setdefault("tmjx_minibatch_begin_bf16", [None, None])[0] = [C.POINTER(Minibatch), fp, C.c_int, vp]
    sig.setdefault("tmjx_philox4x32_10", [None, None])[0] = [fp, fp, vp]
    sig.setdefault("tmjx_latent_concat", [None, None])[0] = [fp] * 4 + [C.c_int] * 4 + [C.c_int64, C.c_int64, fp, fp, C.c_int, C.c_uint64, fp, vp]
    sig.setdefault("tmjx_latent_concat_bwd", [None, None])[0] = [fp] * 4 + [C.c_int] * 3 + [vp]
    sig.setdefault("tmjx_latent_concat_bwd_add", [None, None])[0] = [fp] * 5 + [C.c_int] * 3 + [vp]
    sig.setdefault("tmjx_sample_action", [None, None])[0] = [fp] * 5 + [C.c_int, C.c_int, C.c_uint64, fp, vp]
    sig.setdefault("tmjx_linear_nolds", [None, None])[0] = [fp, C.c_int64, C.c_int64, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_linear_act", [None, None])[0] = [fp, C.c_int64, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, vp]
    sig.setdefault("tmjx_linear_act_ok", [None, None])[0] = [fp, C.c_int64, fp, C.c_int, C.c_int]
    sig.setdefault("tmjx_linear_nolds_norm", [None, None])[0] = [fp, C.c_int64, C.c_int64, fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, vp]
    sig.setdefault("tmjx_linear_nolds_bf16", [None, None])[0] = [fp, C.c_int64, C.c_int64, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, vp]
    sig.setdefault("tmjx_colsum_scratch_floats", [None, None])[0] = [C.c_int]
    sig.setdefault("tmjx_colsum", [None, None])[0] = [fp, fp, fp, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_adam_clip", [None, None])[0] = [fp] * 5 + [C.c_longlong] + [C.c_float] * 7 + [vp]
    sig.setdefault("tmjx_adam_clip_norm", [None, None])[0] = [fp] * 6 + [C.c_longlong] + [C.c_float] * 7 + [vp]
    sig.setdefault("tmjx_gemm_nt", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_gemm_nt_silu_ln_ok", [None, None])[0] = [fp, C.c_int, fp, C.c_int, C.c_int]
    sig.setdefault("tmjx_gemm_nt_silu_ln", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, fp, fp, fp, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    sig.setdefault("tmjx_colsum_grouped", [None, None])[0] = [C.POINTER(ColsumProblem), C.c_int, vp]
    sig.setdefault("tmjx_gemm_nn_ln_bwd_ok", [None, None])[0] = [fp, C.c_int, fp, C.c_int, C.c_int]
    sig.setdefault("tmjx_gemm_nn_ln_bwd_partial_floats", [None, None])[0] = [C.c_int, C.c_int]
    sig.setdefault("tmjx_gemm_nn_ln_bwd_partial_floats", [None, None])[1] = C.c_longlong
    sig.setdefault("tmjx_gemm_nn_ln_bwd", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_gemm_nn", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_gemm_dw_scratch_floats", [None, None])[0] = [C.c_int, C.c_int, C.c_int]
    sig.setdefault("tmjx_gemm_dw_scratch_floats", [None, None])[1] = C.c_longlong
    sig.setdefault("tmjx_gemm_dw", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_gemm_dw_grouped", [None, None])[0] = [C.POINTER(DwProblem), C.c_int, vp]
    sig.setdefault("tmjx_gemm_dw_grouped_wgs", [None, None])[0] = [C.POINTER(DwProblem), C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_gemm_nt_silu_ok", [None, None])[0] = [fp, C.c_int, fp, C.c_int]
    sig.setdefault("tmjx_gemm_nt_silu", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_silu_fwd", [None, None])[0] = [fp, fp, fp, C.c_longlong, C.c_int, vp]
    sig.setdefault("tmjx_silu_bwd", [None, None])[0] = [fp, fp, fp, fp, C.c_longlong, C.c_int, vp]
    sig.setdefault("tmjx_gemm_nn_silu_bwd_ok", [None, None])[0] = [fp, C.c_int, fp, C.c_int]
    sig.setdefault("tmjx_gemm_nn_silu_bwd", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_silu_bwd_rank1", [None, None])[0] = [fp, fp, fp, fp, fp, C.c_longlong, C.c_int, vp]
    sig.setdefault("tmjx_head_dw_scratch_floats", [None, None])[0] = [C.c_int, C.c_int]
    sig.setdefault("tmjx_head_dw_scratch_floats", [None, None])[1] = C.c_longlong
    sig.setdefault("tmjx_head_dw", [None, None])[0] = [fp, fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_head_fwd_ok", [None, None])[0] = [fp, C.c_int, fp, C.c_int]
    sig.setdefault("tmjx_head_fwd", [None, None])[0] = [fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_rollout_store", [None, None])[0] = [C.POINTER(RolloutStore), vp]
    sig.setdefault("tmjx_clips_share", [None, None])[0] = [vp, vp]
    sig.setdefault("tmjx_bf16_shadow", [None, None])[0] = [C.POINTER(Bf16Shadow), C.c_int, vp]
    sig.setdefault("tmjx_bgemm_nt", [None, None])[0] = [fp, C.c_int, C.c_int, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_bgemm_row_tile_ok", [None, None])[0] = [C.c_int]
    sig.setdefault("tmjx_bgemm_partial_floats", [None, None])[0] = [C.c_int, C.c_int, C.c_int]
    sig.setdefault("tmjx_bgemm_partial_floats", [None, None])[1] = C.c_longlong
    sig.setdefault("tmjx_bgemm_ln_fwd", [None, None])[0] = [fp, C.c_int, C.c_int, fp, C.c_int, fp, fp, fp, fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    sig.setdefault("tmjx_bgemm_ln_bwd", [None, None])[0] = [fp, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, fp, fp, fp, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_bgemm_silu_fwd", [None, None])[0] = [fp, C.c_int, C.c_int, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_bgemm_silu_bwd", [None, None])[0] = [fp, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_bf_silu_bwd", [None, None])[0] = [fp, C.c_int, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_bf_silu_bwd_rank1", [None, None])[0] = [fp, fp, fp, C.c_int, fp, fp, C.c_int, fp, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_bgemm_dw_scratch_floats", [None, None])[0] = [C.c_int, C.c_int, C.c_int]
    sig.setdefault("tmjx_bgemm_dw_scratch_floats", [None, None])[1] = C.c_longlong
    sig.setdefault("tmjx_bgemm_dw", [None, None])[0] = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_bgemm_dw_grouped", [None, None])[0] = [C.POINTER(BdwProblem), C.c_int, C.c_int, vp]
    sig.setdefault("tmjx_set_wrappers", [None, None])[0] = [vp, C.c_int, C.c_int]
    sig.setdefault("tmjx_set_action_repeat", [None, None])[0] = [vp, C.c_int]
    sig.setdefault("tmjx_stats_scratch_floats", [None, None])[0] = [C.c_int]
    sig.setdefault("tmjx_stats_sums", [None, None])[0] = [fp, fp, fp, fp, C.c_longlong, C.c_int, vp]
    sig.setdefault("tmjx_stats_apply", [None, None])[0] = [fp, C.c_float, fp, fp, fp, fp, C.c_int, C.c_float, C.c_float, vp]
    sig.setdefault("tmjx_chain_rows", [None, None])[0] = [C.c_int]
    sig.setdefault("tmjx_chain_fwd_ok", [None, None])[0] = [C.POINTER(ChainFwd)]
    sig.setdefault("tmjx_chain_fwd", [None, None])[0] = [C.POINTER(ChainFwd), vp]
    sig.setdefault("tmjx_chain_bwd_ok", [None, None])[0] = [C.POINTER(ChainBwd)]
    sig.setdefault("tmjx_chain_bwd", [None, None])[0] = [C.POINTER(ChainBwd), vp]
    sig.setdefault("tmjx_debug_rows", [None, None])[0] = [vp, C.c_char_p, ip, ip]
    sig.setdefault("tmjx_last_error", [None, None])[1] = C.c_char_p
    sig.setdefault("tmjx_version", [None, None])[1] = C.c_char_p
    restype_set = {"tmjx_model_destroy"}
    for name, (argtypes, restype) in sig.items():
        fn = getattr(L, name, None)      # a library may export a subset (the oracle's ABI twin has no learner kernels): calling a missing one raises
        if fn is None:
            continue
        if argtypes is not None:
            fn.argtypes = argtypes
        if restype is not None or name in restype_set:
            fn.restype = restype
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise TmjxError(f"{what} failed ({rc}): {lib().tmjx_last_error().decode()}")


def block_override() -> str | None:
    return os.environ.get("TMJX_BLOCK")
