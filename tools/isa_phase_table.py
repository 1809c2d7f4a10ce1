#!/usr/bin/env python3
"""Per-phase instruction-class table of k_physics_wave<true> from its ISA: compile csrc/tmjx_wave.hip with -gline-tables-only --save-temps (and the flags of track_mjx_amd/hip.py: SOURCE_FLAGS), map every
instruction of the kernel to the source line of its innermost inlined location (.loc), and the line to the function of csrc/wave_physics.h that
contains it.  STATIC counts (unrolled code counts once per copy, loop bodies once).

usage: python tools/isa_phase_table.py <tmjx_wave-hip-amdgcn-amd-amdhsa-gfx950.s> [kernel mangled-name prefix]"""
import collections
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
asm = Path(sys.argv[1]).read_text().split("\n")
kern = sys.argv[2] if len(sys.argv) > 2 else "_Z14k_physics_waveILb1EE"

# function line ranges of wave_physics.h (TM_DEV / template definitions at column 0)
src = (ROOT / "track_mjx_amd" / "csrc" / "wave_physics.h").read_text().split("\n")
starts = []
for i, l in enumerate(src, 1):
    m = re.match(r"(?:TM_DEV|template\s*<[^>]*>\s*TM_DEV)?\s*(?:TM_DEV\s+)?[\w:<>\*&\s]+?\b(tmw_\w+)\s*\(", l) if l.startswith(("TM_DEV", "template")) else None
    if l.startswith("TM_DEV") and m:
        starts.append((i, m.group(1)))
    elif l.startswith("template") and i < len(src) and src[i].startswith("TM_DEV"):
        pass
starts.sort()


def func_of(line):
    name = "(other)"
    for s, n in starts:
        if s <= line:
            name = n
        else:
            break
    return name


PHASE = [("position", ("tmw_position",)), ("velocity + M", ("tmw_velocity_inertia", "tmw_chain_scan")),
         ("factor / Euler factor (chain kernels)", ("tmw_factor", "tmw_rank1_rows", "tmw_rows_load", "tmw_rows_factor", "tmw_chain_factor", "tmw_factor_chains", "tmw_schur_init", "tmw_schur_flush", "tmw_schur_apply", "tmw_pack4", "tmw_fnma2", "tmw_rcp")),
         ("invert L", ("tmw_invert_l", "tmw_rows_accum", "tmw_chain_trunk_products", "tmw_rows_invert", "tmw_chain_invert", "tmw_invert_chains")),
         ("Euler solve", ("tmw_subst_down", "tmw_rows_subst", "tmw_subst_chains", "tmw_euler")),
         ("M x / M^-1 x halves", ("tmw_rows_colacc", "tmw_colpart_chains", "tmw_row_dot", "tmw_col_dot", "tmw_row_runs2", "tmw_rowpart_chains", "tmw_solve", "tmw_mul_m", "tmw_solve_up", "tmw_solve_down", "tmw_opaque_s")),
         ("J v / J^T f", ("tmw_jmul_stage1", "tmw_jmul_stage2", "tmw_jmul", "tmw_jt_force", "tmw_mul_m_jmul")),
         ("constraint set-up", ("tmw_make_constraint",)),
         ("CG: costs, gradient, updates", ("tmw_dot", "tmw_eval_cost", "tmw_update_gradient", "tmw_solve_cg")),
         ("CG: line search", ("tmw_rowsum16", "tmw_ls_points16", "tmw_ls_points", "tmw_ls_core", "tmw_linesearch")),
         ("wave reductions / broadcasts", ("tmw_dpp_add", "tmw_sum", "tmw_sum_dpp", "tmw_sum_w", "tmw_readlane", "tmw_readlane_dpp", "tmw_table2", "tmw_prefix")),
         ("state load / store, driver", ("tmw_load_state", "tmw_store_state", "tmw_forward", "tmw_anc", "tmw_put_con_frame", "tmw_get_con_frame", "tmw_dump"))]
phase_of = {f: p for p, fs in PHASE for f in fs}

files, in_k, cur = {}, False, ("?", 0)
stats = collections.defaultdict(collections.Counter)
for l in asm:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
        continue
    if l.startswith(kern) and l.rstrip().endswith((":", ")")) or (l.startswith(kern) and ":" in l):
        in_k = True
        continue
    if not in_k:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if not l.startswith("\t") or l.strip().startswith((".", ";")):
        continue
    op = l.split()[0]
    if op == "s_endpgm":
        break
    f, line = cur
    ph = phase_of.get(func_of(line), "(other in wave_physics.h)") if f.endswith("wave_physics.h") else "outside wave_physics.h (tm_common.h helpers, kernel body)"
    if op.startswith("v_mfma"): cls = "mfma"
    elif op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32") or "permlane" in op or op.endswith("_dpp") or "dpp" in l: cls = "lane moves (readlane / permlane / dpp)"
    elif op.startswith("v_cndmask") or op.startswith("v_cmp"): cls = "select / compare"
    elif op.startswith(("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_pk_", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_max_f32", "v_min_f32", "v_div_", "v_mad_f32", "v_ldexp", "v_frexp", "v_rndne", "v_fract", "v_trunc", "v_floor", "v_med3_f32")): cls = "float arithmetic"
    elif op.startswith("v_mov") or op.startswith("v_accvgpr"): cls = "moves"
    elif op.startswith("v_"): cls = "integer / address"
    elif op == "s_nop": cls = "s_nop"
    elif op == "s_waitcnt": cls = "s_waitcnt"
    elif op.startswith("s_"): cls = "scalar"
    elif op.startswith("ds_"): cls = "LDS"
    else: cls = "global memory"
    stats[ph][cls] += 1

cols = ["float arithmetic", "mfma", "lane moves (readlane / permlane / dpp)", "select / compare", "integer / address", "moves", "scalar", "s_nop", "s_waitcnt", "LDS", "global memory"]
tot = collections.Counter()
print(f"{'phase':58s} " + " ".join(f"{c.split(' (')[0][:9]:>9s}" for c in cols) + f" {'total':>7s} {'arith %':>7s}")
for ph, _ in PHASE + [("(other in wave_physics.h)", ()), ("outside wave_physics.h (tm_common.h helpers, kernel body)", ())]:
    c = stats.get(ph)
    if not c:
        continue
    t = sum(c.values())
    valu = sum(c[k] for k in cols[:6])
    print(f"{ph:58s} " + " ".join(f"{c[k]:9d}" for k in cols) + f" {t:7d} {100.0 * (c['float arithmetic'] + c['mfma']) / max(valu, 1):7.1f}")
    tot.update(c)
t = sum(tot.values())
valu = sum(tot[k] for k in cols[:6])
print(f"{'all':58s} " + " ".join(f"{tot[k]:9d}" for k in cols) + f" {t:7d} {100.0 * (tot['float arithmetic'] + tot['mfma']) / max(valu, 1):7.1f}")
print("\n'arith %' = (float arithmetic + mfma) / all vector instructions of the phase.  Static instruction counts.")
