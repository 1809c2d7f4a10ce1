#!/usr/bin/env python3
"""Bit-for-bit A/B of the physics kernel BODY (csrc/wave_physics.h) between a git revision and the working tree, on the CPU.

Data-movement changes of the kernel (what lives in LDS, in registers, in the env's global record) must not change a single bit of the
results.  Both arms are the host emulation of the real kernel body (tests/hostemu: 64 emulated lanes, an LDS image per env), built once from
`git show <rev>:...` and once from the working tree, stepped FREE-RUNNING from the same states with the same actions — through contact, at the
three action scales of tests/test_gpu_parity_strict.py — and compared after every substep: every physics row (qpos, qvel, act, warm start,
time), the outputs of the last forward pass (xpos, torso xmat, qfrc_actuator) and the solver statistics.  Prints the worst absolute
difference (0 expected) and exits non-zero otherwise.
TMJX_EMU_POISON=nan in the environment fills what a wave finds when it starts (LDS image; in the working tree's build also registers and scratch)
with NaNs instead of zeros: a run under it must give the same bits as one without (no result may depend on words the kernel did not write).
usage: [TMJX_EMU_POISON=nan] python tests/diagnostics/kernel_ab.py [rev = HEAD] [n_env = 48] [control steps = 4]"""
import ctypes as C
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "hostemu"))
import emu as _emu  # noqa: E402
from tests.common import default_blob, default_walker  # noqa: E402
from track_mjx_amd import clips as _clips  # noqa: E402

rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
nctl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tmp = Path(tempfile.mkdtemp())


POISON_PATCH = ("    for (auto &v : lds) v = 0.f;\n", "    { const char *ps_ = getenv(\"TMJX_EMU_POISON\"); const float pz_ = ps_ ? (!strcmp(ps_, \"nan\") ? __builtin_nanf(\"\") : (float)atof(ps_)) : 0.f; for (auto &v : lds) v = pz_; }\n")


def build(tag, tree: Path, defs=()):
    so = tmp / f"libhostemu_{tag}.so"
    src = tree / "tests" / "hostemu" / "hostemu.cpp"
    text = src.read_text()
    if "TMJX_EMU_POISON" not in text and POISON_PATCH[0] in text and tree != ROOT:       # an older revision: give its LDS image the same poison switch
        src.write_text(text.replace(POISON_PATCH[0], POISON_PATCH[1]))
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", *defs, "-o", str(so), str(tree / "tests" / "hostemu" / "hostemu.cpp")], check=True)
    return so


def checkout(rev: str) -> Path:
    """The files the emulation is compiled from, as they were at `rev`."""
    tree = tmp / "rev"
    files = subprocess.run(["git", "-C", str(ROOT), "ls-tree", "-r", "--name-only", rev, "track_mjx_amd/csrc", "tests/hostemu", "tests/lane", "include"],
                           check=True, capture_output=True, text=True).stdout.split()
    for f in files:
        if not f.endswith((".h", ".cpp", ".hip")):
            continue
        dst = tree / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_bytes(subprocess.run(["git", "-C", str(ROOT), "show", f"{rev}:{f}"], check=True, capture_output=True).stdout)
    return tree


class EmuAt(_emu.Emu):
    def __init__(self, so, blob, n_env):
        orig = C.CDLL
        try:
            C.CDLL = lambda _p: orig(str(so))        # the wrapper loads tests/hostemu/libhostemu.so by path: hand it this build instead
            _emu.C.CDLL = C.CDLL
            super().__init__(blob, n_env)
        finally:
            C.CDLL = orig; _emu.C.CDLL = orig


def init_states(cl, n, rng, sink=0.001):
    qpos = np.zeros((n, 74)); qvel = rng.uniform(-1e-3, 1e-3, size=(n, 73))
    for e in range(n):
        c, f = e % cl.position.shape[0], (7 * e) % 44
        qpos[e] = np.concatenate([cl.position[c, f], cl.quaternion[c, f], cl.joints[c, f]]) + rng.uniform(-1e-3, 1e-3, 74)
        qpos[e, 2] -= sink * (e % 5)
    return qpos, qvel


w, cfg = default_walker()
cl = _clips.make_synthetic_clips(w.model, 4, seed=0)
blob = default_blob(w, cfg, auto_reset=False)
A = EmuAt(build("a", checkout(rev)), blob, n)
B = EmuAt(build("b", ROOT), blob, n)
print(f"A = {rev}: {A.lds_bytes()} bytes of LDS per env;  B = working tree: {B.lds_bytes()} bytes")
ROWS = ("qpos", "qvel", "act", "qacc_warmstart", "time", "xpos", "xmat_torso", "qfrc_actuator")
worst_all = 0.0
for scale in (0.03, 0.3, 1.0):
    rng = np.random.default_rng(11)
    qpos, qvel = init_states(cl, n, rng)
    for E in (A, B):
        E.st[:] = 0; E.ws[:] = 0
        E.rows("qpos")[:] = qpos.T; E.rows("qvel")[:] = qvel.T
        E.physics_wave(None, 1, do_euler=False)              # (forward only: the outputs of a first forward pass exist)
    worst, nan_envs, contact = 0.0, 0, 0
    for ctl in range(nctl):
        a = np.clip(rng.normal(size=(n, 38)) * scale, -1, 1).T.astype(np.float32).copy()
        for sub in range(10):
            for E in (A, B):
                E.physics_wave(a, 1, True, dump=True)
            for k in ROWS + ("solver_stats",):
                try:
                    x, y = A.rows(k), B.rows(k)
                except KeyError:
                    continue
                same = (x == y) | (np.isnan(x) & np.isnan(y))
                if not same.all():
                    d = np.abs(np.where(same, 0.0, x.astype(np.float64) - y.astype(np.float64)))
                    worst = max(worst, float(np.nanmax(d)) if np.isfinite(d).any() else float("inf"))
            try:
                contact += int((A.rows("con_dist") < 0).any(0).sum())
            except KeyError:
                pass
    nan_envs = int(np.isnan(A.rows("qpos")).any(0).sum())
    print(f"action scale {scale}: {n} envs x {nctl * 10} free-running substeps, env-substeps with a penetrating contact {contact}, envs ending in NaN {nan_envs}: worst diff {worst:g}")
    worst_all = max(worst_all, worst)
print("worst diff", worst_all)
sys.exit(0 if worst_all == 0.0 else 1)
