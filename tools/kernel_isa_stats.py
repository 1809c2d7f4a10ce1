#!/usr/bin/env python3
"""Static instruction-class counts of one kernel of a built library (disassembly of its gfx950 code object).

usage: python tools/kernel_isa_stats.py <lib.so> [kernel name prefix = _Z14k_physics_waveILb1EE]
"""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

BUNDLER, OBJDUMP = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(so: Path, kern: str) -> list[str]:
    with tempfile.TemporaryDirectory() as d:
        fat = Path(d) / "fat.bin"
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(so), str(fat)], check=True)
        blob, magic = fat.read_bytes(), b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        for i in range(len(starts) - 1):
            part, co = Path(d) / f"fat{i}.bin", Path(d) / f"k{i}.co"
            part.write_bytes(blob[starts[i]:starts[i + 1]])
            subprocess.run([BUNDLER, "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(co)], check=True, capture_output=True, text=True).stdout
            m = re.search(rf"^[0-9a-f]+ <({re.escape(kern)}[^>]*)>:\n(.*?)(?=^\n[0-9a-f]+ <|\Z)", txt, re.S | re.M)
            if m:
                return [l.split("//")[0].strip() for l in m.group(2).split("\n") if l.startswith(("\t", " ")) and l.strip()]
    raise SystemExit(f"kernel {kern} not found in {so}")


def classify(op: str) -> str:
    if op.startswith(("flat_load", "global_load", "buffer_load")): return "vector memory load"
    if op.startswith(("flat_store", "global_store", "buffer_store", "global_atomic", "flat_atomic")): return "vector memory store"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "scalar memory load"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane") or "_dpp" in op or op.startswith("v_permlane"): return "lane moves (readlane / writelane / DPP)"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"): return "select / compare"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "other vector"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "other scalar"
    return "other"


if __name__ == "__main__":
    so = Path(sys.argv[1])
    kern = sys.argv[2] if len(sys.argv) > 2 else "_Z14k_physics_waveILb1EE"
    ins = disassemble(so, kern)
    cnt = collections.Counter(classify(l.split()[0]) for l in ins)
    print(f"{so}: {kern}: {len(ins)} instructions")
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"  {k:42s} {v:6d}")
