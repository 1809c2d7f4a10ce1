#!/usr/bin/env python3
"""Print VGPR / SGPR / LDS / scratch of every gfx950 kernel in a built library (from the code object's metadata notes).

usage: python tools/kernel_resources.py [libtmjx_hip.so] [name filter]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
BUNDLER, READELF = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_resources(so: Path) -> dict:
    with tempfile.TemporaryDirectory() as d:
        fat, co = Path(d) / "fat.bin", Path(d) / "k.co"
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(so), str(fat)], check=True)
        # one offload bundle per translation unit, back to back in the section
        blob, magic, notes = fat.read_bytes(), b"__CLANG_OFFLOAD_BUNDLE__", ""
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        for i in range(len(starts) - 1):
            part = Path(d) / f"fat{i}.bin"
            part.write_bytes(blob[starts[i]:starts[i + 1]])
            subprocess.run([BUNDLER, "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes += subprocess.run([READELF, "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    out = {}
    # one metadata map per kernel: fields in alphabetical order, .name before .private_segment_fixed_size .. .vgpr_count
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
        out[get("name")] = {k: get(k) for k in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
                                               "private_segment_fixed_size", "max_flat_workgroup_size")}
    return out


if __name__ == "__main__":
    so = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "track_mjx_amd" / "libtmjx_hip.so"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, r in sorted(kernel_resources(so).items()):
        if flt in name:
            print(f"{name[:70]:70s} vgpr {r['vgpr_count']:>4} agpr {r['agpr_count']:>3} sgpr {r['sgpr_count']:>4} spill v{r['vgpr_spill_count']}/s{r['sgpr_spill_count']} "
                  f"lds {r['group_segment_fixed_size']:>6} scratch {r['private_segment_fixed_size']:>5}")
