#!/usr/bin/env python3
"""One env group's serial phase between two physics launches, from a rocprofv3 kernel trace of bench.py (rocpd sqlite): every kernel of the group's
stream between the end of one k_physics_wave and the start of the next — start offset, duration, name.
usage: python tools/rollout_timeline.py x_results.db [index of the gap to print]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
by = collections.defaultdict(list)
for r in rows:
    by[r[3]].append(r)
# the queue with the most physics launches
q = max(by, key=lambda x: sum("k_physics_wave" in r[0] for r in by[x]))
seq = by[q]
phys = [i for i, r in enumerate(seq) if "k_physics_wave" in r[0]]
i0, i1 = phys[k], phys[k + 1]
t0 = seq[i0][2]
print(f"queue {q}: physics launch {k}: {(seq[i0][2] - seq[i0][1]) / 1e3:.1f} us; serial phase behind it:")
prev_end = t0
for n, s, e, _ in seq[i0 + 1:i1 + 1]:
    print(f"  +{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f} us  {n[:70]}")
    prev_end = e
print(f"serial phase {(seq[i1][1] - t0) / 1e3:.1f} us")
print("kernels of OTHER queues that start inside this window (graph branches, copies):")
for n, s, e, qq in rows:
    if qq != q and t0 <= s <= seq[i1][1] and "k_physics_wave" not in n:
        print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  queue {qq}  {n[:80]}")
gaps = [(seq[phys[j + 1]][1] - seq[phys[j]][2]) / 1e3 for j in range(len(phys) - 1)]
gaps = [g for g in gaps if g < 5000]
print(f"median serial phase over {len(gaps)} steps: {sorted(gaps)[len(gaps) // 2]:.1f} us; physics launch median {sorted((r[2] - r[1]) / 1e3 for r in seq if 'k_physics_wave' in r[0])[len(phys) // 2]:.1f} us")
