#!/usr/bin/env python3
"""Instrumented operation count of one physics substep / one control step of the CPU oracle.

Runs the flop-counting build (`make -C oracle count`: `real` counts its own + - * / and special functions) on the bench workload's
kind of state — synthetic clips, 64 envs, 0.3-scaled N(0,1) actions, 20 control steps so that paws are on the floor — and writes the
per-substep and per-control-step averages to profiles/oracle_flop_count.json.  The oracle is MJX's DENSE formulation (dense 73 x 73
inertia matrix and Cholesky, dense 187 x 73 constraint Jacobian), so this is the operation count of the algorithm as the reference
executes it; the HIP kernel's tree-sparse / matrix-free formulation executes fewer (its own count comes from the SQ_INSTS_VALU
counter, profiles/).  Test infrastructure: not imported by the product.
"""
import ctypes as C
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[0] = str(ROOT)          # (the script directory would shadow the `oracle` package)
from oracle.oracle import Oracle  # noqa: E402
from tests.common import default_blob, default_walker  # noqa: E402
from track_mjx_amd import clips as _clips  # noqa: E402


def main():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "count"], check=True, capture_output=True)
    w, cfg = default_walker()
    clip = _clips.make_synthetic_clips(w.model, 4, seed=0)
    O = Oracle(default_blob(w, cfg), "count")
    O.set_clips(clip.as_dict())
    get, reset = O.L.oracle_count_get, O.L.oracle_count_reset
    buf = (C.c_ulonglong * 5)()

    def read():
        get(buf)
        return np.array(list(buf), dtype=np.float64)
    n, steps = 64, 20
    rng = np.random.default_rng(0)
    envs = O.new_envs(n)
    for e in range(n):
        O.env_reset(envs, e, e % 4, e % 44, rng.uniform(-1e-3, 1e-3, 74), rng.uniform(-1e-3, 1e-3, 73))
    phys, post, niter, ls = np.zeros(5), np.zeros(5), [], []
    for s in range(steps):
        a = np.clip(rng.normal(size=(n, 38)) * 0.3, -1, 1)
        for e in range(n):
            reset(); O.env_step(envs, e, a[e]); full = read()
            niter.append(O.env_get(envs, e, "solver_niter")[0]); ls.append(O.env_get(envs, e, "ls_total")[0])
            reset(); O.env_post(envs, e, a[e]); k3 = read()        # K3 alone on the new state (a second bookkeeping step: same cost)
            phys += full - k3; post += k3
    per_sub = phys / (n * steps * 10)
    per_post = post / (n * steps)
    names = ("add_sub", "mul", "div", "special", "min_max_abs_floor")
    flops_sub = float(per_sub[:4].sum())
    out = {"what": "operations per env of the CPU oracle (MJX dense formulation), averaged over 64 envs x 20 control steps x 10 substeps, "
                   "0.3-scaled N(0,1) actions; flops = add/sub + mul + div + special functions (1 each)",
           "per_substep": dict(zip(names, per_sub.round(1).tolist())), "flops_per_substep": round(flops_sub, 1),
           "per_control_step_K3": dict(zip(names, per_post.round(1).tolist())), "flops_per_control_step_K3": round(float(per_post[:4].sum()), 1),
           "flops_per_env_step": round(10 * flops_sub + float(per_post[:4].sum()), 1),
           "mean_solver_niter_last_substep": float(np.mean(niter)), "mean_ls_total_last_substep": float(np.mean(ls))}
    (ROOT / "profiles" / "oracle_flop_count.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
