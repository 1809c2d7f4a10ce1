#!/usr/bin/env python3
"""Scan the device assembly of every translation unit for 64-bit scalar moves / logic ops whose 32-bit literal the COMPILER meant to be
sign-extended (its own listing prints them as 0xffffffffXXXXXXXX).  gfx950 zero-extends the literal of s_mov_b64 (tools/micro/literal64.hip),
so such an instruction silently clears lanes 32-63 of a mask.  hipcc (LLVM 22, ROCm 7.2) emits them for 64-bit constants whose upper 33 bits
are all ones; csrc/wave_physics.h builds those masks from two halves (tmw_lit64) — this script is the check that none slipped through.

usage: python tools/scan_sext_literals.py        (compiles each TU with -S, about 1.5 min per TU)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from track_mjx_amd import hip  # noqa: E402

bad = 0
with tempfile.TemporaryDirectory() as d:
    for src in hip.SOURCES:
        out = Path(d) / (Path(src).stem + ".s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-unused-value", "--cuda-device-only", "-S", "-o", str(out), str(src)],
                       check=True, capture_output=True)
        kern = "?"
        for ln in out.read_text().split("\n"):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                kern = m.group(1)
            if re.search(r"\bs_\w+_b64\b.*,\s*0xffffffff[0-9a-f]{8}\s*$", ln):
                dead = "tmw_rows_subst" if False else ""
                print(f"{Path(src).name}: {kern[:48]}: {ln.strip()} {dead}")
                bad += 1
print(f"{bad} sign-extended 64-bit literal(s)")
sys.exit(1 if bad else 0)
