#!/usr/bin/env python3
"""Build id of the HIP library (the source-derived id written by hip.build(), see build_id) — the tie between a counter summary under profiles/ and the kernel it
was measured on.

  python tools/buildid.py                 print the id of track_mjx_amd/libtmjx_hip.so (or $TMJX_SO)
  python tools/buildid.py --stamp DIR     write DIR/so_build_id.txt (the collection scripts do this ON THE GPU BOX, next to the counters)

The summary scripts (pmc_summary.py, sq_summary.py, mfma_summary.py) call `checked_id(src)`: the id stamped next to the counters must be the id of
the library in this tree, else they refuse to write profiles/*.json (pass --force to record a foreign build, which is then labelled)."""
import hashlib
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def so_path() -> Path:
    return Path(os.environ.get("TMJX_SO", str(ROOT / "track_mjx_amd" / "libtmjx_hip.so")))


def build_id(path: Path | None = None) -> str:
    """The source-derived id hip.build() wrote next to the library (<lib>.id: hash of flags + sources + include closure — hipcc's objects are not
    bit-reproducible, a hash of the .so would change with every rebuild of unchanged sources); a library without one: the hash of its bytes."""
    path = Path(path or so_path())
    side = Path(str(path) + ".id")
    try:
        if side.exists() and side.stat().st_mtime >= path.stat().st_mtime - 1:
            return side.read_text().strip()
        return hashlib.sha256(path.read_bytes()).hexdigest()[:16]
    except OSError:
        return "missing"


def checked_id(src: Path, force: bool = False) -> dict:
    """{"so_build_id": ..., "so_build_id_matches_tree": bool}; raises SystemExit if the stamp is missing or foreign and not forced."""
    stamp = Path(src) / "so_build_id.txt"
    here = build_id()
    if not stamp.exists():
        if not force:
            raise SystemExit(f"{stamp} is missing: the counters are not tied to a build (re-collect with the current tools/*.sh, or --force)")
        return {"so_build_id": None, "so_build_id_matches_tree": False}
    there = stamp.read_text().strip()
    if there != here and not force:
        raise SystemExit(f"counters under {src} were collected on build {there}, the tree holds {here}: refusing to write profiles/ (--force to label and keep)")
    return {"so_build_id": there, "so_build_id_matches_tree": there == here}


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--stamp":
        d = Path(sys.argv[2]); d.mkdir(parents=True, exist_ok=True)
        (d / "so_build_id.txt").write_text(build_id() + "\n")
    print(build_id())
