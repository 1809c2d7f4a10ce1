#!/usr/bin/env python3
"""Why is the HIP physics kernel's M^-1 b about 2.5 x less accurate than the float32 restatement of MJX's dense path on the typical env?
(tests/diagnostics/stage_errors.py: the gap opens at qacc_smooth = M^-1 qfrc_smooth and nowhere before.)

CPU experiment on the rodent's own inertia matrices (float64 oracle -> M, qfrc_smooth; cond(M) ~ 4e5), every operation rounded to float32:
  * sparse L^T D L, leaf -> root, + substitution  = MuJoCo's native mj_factorM / mj_solveM order, the order any fill-free tree-sparse
    factorisation has to use (and the one csrc/wave_physics.h uses);
  * dense Cholesky L L^T, root -> leaf             = what MJX's dense path (jax.scipy cho_factor) and oracle/tmjx_oracle.c do.
Result (32 poses): the leaf -> root order is 4.4 x less accurate on the median pose (2.9e-6 against 6.8e-7) — eliminating the light distal
links first subtracts their contribution from the composite inertias on their ancestors' diagonals (cancellation); one step of iterative
refinement with a float32 residual only recovers half of it (1.3e-6).  The gap is a property of the elimination order, not of the
kernel's reciprocal (v_rcp_f32 + one Newton step) or of its MFMA trunk block.   usage: python tests/diagnostics/ldl_vs_cholesky.py"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests.common import default_blob, default_walker, make_oracle
from tests.test_gpu_parity_strict import _init_states
from track_mjx_amd import clips as _clips
w, cfg = default_walker()
cl = _clips.make_synthetic_clips(w.model, 4, seed=0)
blob = default_blob(w, cfg, auto_reset=False)
O = make_oracle(blob, cl, "f64")
rng = np.random.default_rng(11)
qpos, qvel = _init_states(cl, 32, rng)
par = np.array(w.model["dof_parentid"])
f32 = np.float32
def ldl_sparse(M):   # mj_factorM-style L^T D L in float32, leaf -> root, exploiting nothing (dense loops, same order)
    n = M.shape[0]; A = M.astype(f32).copy()
    for k in range(n - 1, -1, -1):
        Akk = A[k, k]
        i = par[k]
        while i >= 0:
            t = f32(A[k, i] / Akk)
            j = i
            while j >= 0:
                A[i, j] = f32(A[i, j] - f32(A[k, j] * t)); j = par[j]
            A[k, i] = t
            i = par[i]
    return A
def ldl_solve(A, b):
    n = len(b); x = b.astype(f32).copy()
    for i in range(n - 1, -1, -1):      # x <- L^-T x
        j = par[i]
        while j >= 0:
            x[j] = f32(x[j] - f32(A[i, j] * x[i])); j = par[j]
    for i in range(n): x[i] = f32(x[i] / A[i, i])
    for i in range(n):                  # x <- L^-1 x
        j = par[i]
        while j >= 0:
            x[i] = f32(x[i] - f32(A[i, j] * x[j])); j = par[j]
    return x
def chol_solve32(M, b):
    A = M.astype(f32).copy(); n = len(b)
    Lc = np.zeros_like(A)
    for j in range(n):
        s = A[j, j] - np.dot(Lc[j, :j], Lc[j, :j]).astype(f32)
        Lc[j, j] = np.sqrt(f32(s))
        for i in range(j + 1, n):
            Lc[i, j] = f32((A[i, j] - np.dot(Lc[i, :j], Lc[j, :j]).astype(f32)) / Lc[j, j])
    y = np.zeros(n, f32)
    for i in range(n): y[i] = f32((f32(b[i]) - np.dot(Lc[i, :i], y[:i]).astype(f32)) / Lc[i, i])
    x = np.zeros(n, f32)
    for i in range(n - 1, -1, -1): x[i] = f32((y[i] - np.dot(Lc[i + 1:, i], x[i + 1:]).astype(f32)) / Lc[i, i])
    return x
e_ldl, e_ch, e_inv, conds = [], [], [], []
for e in range(32):
    d = O.new_data(qpos[e], qvel[e]); O.forward(d)
    M = O.get(d, "qM").reshape(73, 73); b = O.get(d, "qfrc_smooth")
    M = np.tril(M) + np.tril(M, -1).T
    x64 = np.linalg.solve(M, b)
    A = ldl_sparse(M); x1 = ldl_solve(A, b)
    x2 = chol_solve32(M, b)
    x3 = (np.linalg.inv(M.astype(f32)).astype(f32) @ b.astype(f32)).astype(f32)
    rel = lambda x: np.abs(x - x64).max() / np.abs(x64).max()
    e_ldl.append(rel(x1)); e_ch.append(rel(x2)); e_inv.append(rel(x3)); conds.append(np.linalg.cond(M))
print("cond median %.3g" % np.median(conds))
print("sparse LTDL + substitution (f32): median %.2e" % np.median(e_ldl))
print("dense Cholesky (f32):             median %.2e" % np.median(e_ch))
print("ratio %.2f" % (np.median(e_ldl) / np.median(e_ch)))
e_ref, e_ref_ch = [], []
for e in range(32):
    d = O.new_data(qpos[e], qvel[e]); O.forward(d)
    M = O.get(d, "qM").reshape(73, 73); b = O.get(d, "qfrc_smooth")
    M = np.tril(M) + np.tril(M, -1).T
    x64 = np.linalg.solve(M, b)
    A = ldl_sparse(M); x0 = ldl_solve(A, b)
    M32 = M.astype(f32)
    r = (b.astype(f32) - (M32 @ x0).astype(f32)).astype(f32)       # fp32 residual
    x1 = (x0 + ldl_solve(A, r)).astype(f32)
    rel = lambda x: np.abs(x - x64).max() / np.abs(x64).max()
    e_ref.append(rel(x1))
print("LTDL + one refinement step (fp32 residual): median %.2e" % np.median(e_ref))
