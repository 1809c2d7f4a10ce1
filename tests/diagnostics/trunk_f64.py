#!/usr/bin/env python3
"""What does the float64 trunk block of the physics kernel's L^T D L buy on the REAL kernel body?  (round-3 verdict, "next" item 1.)

CPU experiment through the host emulation of csrc/wave_physics.h (tests/hostemu): two builds of the same source — -DTMW_TRUNK_F32 (round 3's
float32 Schur accumulation and trunk elimination) and the default (v_mfma_f64 accumulation + float64 trunk rows) — stepped teacher-forced from
the float64 oracle's states, next to the float32 oracle (MJX's dense formulation in float32).  Printed per action scale: the relative error of
qacc_smooth = M^-1 qfrc_smooth (the stage where the gap opens: tests/diagnostics/stage_errors.py) and of one substep's qvel against the float64
oracle, median / 99th percentile / fraction within 1e-5, and the ratio to the float32 oracle's.
usage: python tests/diagnostics/trunk_f64.py [n_env] [substeps]"""
import ctypes as C
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "hostemu"))
import emu as _emu  # noqa: E402
from tests.common import default_blob, default_walker, make_oracle, rel_err  # noqa: E402
from tests.test_gpu_parity_strict import PHYS, _init_states  # noqa: E402
from track_mjx_amd import clips as _clips  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tmp = Path(tempfile.mkdtemp())


def build(tag, defs):
    so = tmp / f"libhostemu_{tag}.so"
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", *defs, "-o", str(so), str(ROOT / "tests" / "hostemu" / "hostemu.cpp")], check=True)
    return so


class EmuAt(_emu.Emu):
    def __init__(self, so, blob, n_env):
        orig = C.CDLL
        try:
            C.CDLL = lambda _p: orig(str(so))        # the wrapper loads tests/hostemu/libhostemu.so by path: hand it this build instead
            _emu.C.CDLL = C.CDLL
            super().__init__(blob, n_env)
        finally:
            C.CDLL = orig; _emu.C.CDLL = orig


w, cfg = default_walker()
cl = _clips.make_synthetic_clips(w.model, 4, seed=0)
blob = default_blob(w, cfg, auto_reset=False)
arms = {"trunk f32 (round 3)": EmuAt(build("f32", ["-DTMW_TRUNK_F32"]), blob, n), "trunk f64": EmuAt(build("f64", []), blob, n),
        "Schur f64, elimination f32": EmuAt(build("s64", ["-DTMW_TRUNK_ELIM_F32"]), blob, n)}
O32, O64 = make_oracle(blob, cl, "f32"), make_oracle(blob, cl, "f64")
for scale in (0.03, 0.3, 1.0):
    rng = np.random.default_rng(11)
    qpos, qvel = _init_states(cl, n, rng)
    d32 = [O32.new_data(qpos[e], qvel[e]) for e in range(n)]; d64 = [O64.new_data(qpos[e], qvel[e]) for e in range(n)]
    errs = {k: {"qacc_smooth": [], "qvel": []} for k in list(arms) + ["f32 oracle"]}
    for sub in range(nsub):
        a = np.clip(rng.normal(size=(n, 38)) * scale, -1, 1)
        st = {k: np.stack([O64.get(d, k) for d in d64], 1) for k in PHYS}
        for e in range(n):
            for k, v in st.items():
                O32.set(d32[e], k, v[:, e])
        f64d, f32d = [], []
        for e in range(n):
            for O, dd in ((O32, f32d), (O64, f64d)):
                d = O.new_data(st["qpos"][:, e], st["qvel"][:, e]); O.set(d, "act", st["act"][:, e]); O.set(d, "qacc_warmstart", st["qacc_warmstart"][:, e]); O.forward(d); dd.append(d)
        r64 = np.stack([O64.get(d, "qacc_smooth") for d in f64d], 1)
        errs["f32 oracle"]["qacc_smooth"].append(rel_err(np.stack([O32.get(d, "qacc_smooth") for d in f32d], 1), r64, axis=0))
        for name, E in arms.items():
            for k, v in st.items():
                E.rows(k)[:] = v
            E.physics_wave(None, 1, do_euler=False)
            errs[name]["qacc_smooth"].append(rel_err(E.rows("qacc_smooth"), r64, axis=0))
            for k, v in st.items():
                E.rows(k)[:] = v
            E.physics_wave(a.T.astype(np.float32).copy(), 1, True, dump=False)
        for e in range(n):
            O32.step(d32[e], a[e]); O64.step(d64[e], a[e])
        ref = np.stack([O64.get(d, "qvel") for d in d64], 1)
        ok = np.isfinite(ref).all(0)
        errs["f32 oracle"]["qvel"].append(rel_err(np.stack([O32.get(d, "qvel") for d in d32], 1)[:, ok], ref[:, ok], axis=0))
        for name, E in arms.items():
            errs[name]["qvel"].append(rel_err(E.rows("qvel")[:, ok], ref[:, ok], axis=0))
    print(f"action scale {scale}: {n} envs x {nsub} teacher-forced substeps, relative error against the float64 oracle")
    for stage in ("qacc_smooth", "qvel"):
        base = np.concatenate(errs["f32 oracle"][stage])
        for name in errs:
            v = np.concatenate(errs[name][stage])
            print(f"  {stage:12s} {name:22s} median {np.median(v):.2e} ({np.median(v) / np.median(base):4.2f} x f32 oracle)  p99 {np.quantile(v, .99):.2e}  worst {v.max():.2e}  within 1e-5: {np.mean(v <= 1e-5):.3f}")
