"""Flat "model blob": the binary hand-off between the host (Python) and the C-ABI.

Layout (little endian):
    u32 magic 'TMJX' (0x584a4d54), u32 version, u32 n_entries, u32 pad
    per entry: char name[32]; i32 dtype (0 = int32, 1 = float64); i32 count;
               payload, zero padded to a multiple of 8 bytes.

Both the HIP library (track_mjx_amd/csrc/blob_reader.h) and the oracle
(oracle/tmjx_oracle.c) look entries up by name; float64 payloads are narrowed to
fp32 by the consumer (MJX's put_model does the same narrowing of MuJoCo's
float64 model: reference call site track_mjx/environment/task/single_clip_tracking.py:91).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

MAGIC = 0x584A4D54
VERSION = 1


def pack(entries: "OrderedDict[str, np.ndarray]") -> bytes:
    out = [struct.pack("<IIII", MAGIC, VERSION, len(entries), 0)]
    for name, arr in entries.items():
        arr = np.asarray(arr)
        if arr.dtype.kind in "iub":
            code, data = 0, np.ascontiguousarray(arr, dtype="<i4")
        else:
            code, data = 1, np.ascontiguousarray(arr, dtype="<f8")
        bname = name.encode()
        if len(bname) > 31:
            raise ValueError(f"blob entry name too long: {name}")
        payload = data.tobytes()
        pad = (-len(payload)) % 8
        out.append(bname.ljust(32, b"\0"))
        out.append(struct.pack("<ii", code, data.size))
        out.append(payload + b"\0" * pad)
    return b"".join(out)


def unpack(buf: bytes) -> "OrderedDict[str, np.ndarray]":
    magic, version, n, _ = struct.unpack_from("<IIII", buf, 0)
    if magic != MAGIC or version != VERSION:
        raise ValueError("not a TMJX model blob")
    off = 16
    entries: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for _ in range(n):
        name = buf[off:off + 32].split(b"\0", 1)[0].decode()
        code, count = struct.unpack_from("<ii", buf, off + 32)
        off += 40
        if code == 0:
            arr = np.frombuffer(buf, dtype="<i4", count=count, offset=off).copy()
            nbytes = 4 * count
        else:
            arr = np.frombuffer(buf, dtype="<f8", count=count, offset=off).copy()
            nbytes = 8 * count
        off += nbytes + ((-nbytes) % 8)
        entries[name] = arr
    return entries


def load(path) -> "OrderedDict[str, np.ndarray]":
    with open(path, "rb") as f:
        return unpack(f.read())


def save(path, entries) -> None:
    with open(path, "wb") as f:
        f.write(pack(entries))
