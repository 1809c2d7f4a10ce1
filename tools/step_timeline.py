#!/usr/bin/env python3
"""Timeline of ONE minibatch SGD step (kernel, start offset, duration) from a rocprofv3 kernel trace of bench.py (rocpd sqlite).
usage: python tools/step_timeline.py x_results.db [step index inside the last SGD phase]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = c.execute("select name, start, end from kernels order by start").fetchall()
phys = [r for r in rows if "k_physics_wave" in r[0]]
gaps = [(phys[i][2], phys[i + 1][1]) for i in range(len(phys) - 1) if phys[i + 1][1] - phys[i][2] > 12e6]
a, b = gaps[-1]
win = [r for r in rows if r[1] >= a and r[2] <= b]
adam = [i for i, r in enumerate(win) if "k_adam_clip" in r[0]]
i0, i1 = adam[k] + 1, adam[k + 1] + 1
t0 = win[i0][1]
for n, s, e in win[i0:i1]:
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  {n[:100]}")
print("step wall us", (win[i1 - 1][2] - t0) / 1e3)
