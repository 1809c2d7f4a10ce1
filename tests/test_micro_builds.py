"""The micro-benchmarks under tools/micro (what DESIGN.md's hardware figures were measured with) still parse as gfx950 device code: hipcc -fsyntax-only,
device side only — no GPU, no code generation, a few seconds in all."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SOURCES = sorted((ROOT / "tools" / "micro").glob("*.hip"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", SOURCES, ids=[s.name for s in SOURCES])
def test_micro_benchmark_source_parses_for_gfx950(src):
    assert len(SOURCES) >= 8
    res = subprocess.run(["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-fsyntax-only", "-std=c++17", str(src)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
