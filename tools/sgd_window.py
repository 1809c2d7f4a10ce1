#!/usr/bin/env python3
"""Per-kernel totals inside ONE SGD phase (update()) of a rocprofv3 kernel trace of bench.py (rocpd sqlite output): the window between the
last physics launch of a roll-out and the first of the next one.  usage: python tools/sgd_window.py gpurun_out/prof/x_results.db [top]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
phys = [r for r in rows if "k_physics_wave" in r[0]]
gaps = [(phys[i][2], phys[i + 1][1]) for i in range(len(phys) - 1) if phys[i + 1][1] - phys[i][2] > 30e6]
a, b = gaps[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
busy = []
for n, s, e, st in rows:
    if s >= a and e <= b:
        agg[n][0] += 1
        agg[n][1] += e - s
        busy.append((s, e))
busy.sort()
cov, cur_s, cur_e = 0, None, None
for s, e in busy:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += (cur_e - cur_s) if cur_e else 0
tot = sum(v[1] for v in agg.values())
print(f"window {(b - a) / 1e6:.2f} ms, kernel-time sum {tot / 1e6:.2f} ms, time with >= 1 kernel running {cov / 1e6:.2f} ms (idle {(b - a - cov) / 1e6:.2f} ms), {sum(v[0] for v in agg.values())} launches")
for n, (cnt, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print(f"{n[:110]:110s} {cnt:5d} {t / 1e6:8.2f} ms {t / cnt / 1e3:7.1f} us")
