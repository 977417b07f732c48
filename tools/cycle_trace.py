#!/usr/bin/env python3
"""Timeline of the SCF cycles from a rocprofv3 kernel trace of bench.py: per cycle (J/K launch to J/K launch) the busy time per
kernel family and the idle time on the device.  python3 tools/cycle_trace.py <kernel_trace.csv>"""
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
jk = [i for i, r in enumerate(rows) if "jk_tiles_kernel<true, true, true>" in r[2]]
# the RHF timed region: the longest run of J/K launches spaced by less than 2 ms
runs, cur = [], [jk[0]]
for a, b in zip(jk, jk[1:]):
    if rows[b][0] - rows[a][0] < 2_000_000: cur.append(b)
    else: runs.append(cur); cur = [b]
runs.append(cur)
run = max(runs, key=len)
if len(sys.argv) > 2 and sys.argv[2] == "rks":   # the B3LYP leg: cycles that contain the XC kernels
    run = jk
print("runs of J/K launches (length):", [len(r) for r in runs])
import statistics
d = [(rows[b][0] - rows[a][0]) / 1e3 for a, b in zip(run, run[1:])]
print("cycles in the longest run:", len(run) - 1, "J/K-to-J/K us: min %.0f median %.0f max %.0f" % (min(d), statistics.median(d), max(d)))
print("last 12:", [round(x) for x in d[-12:]])
fam = collections.defaultdict(float); idle = 0.0; tot = 0.0; n = 0
want = "xc_vmat" if (len(sys.argv) > 2 and sys.argv[2] == "rks") else "sp2_"
pairs = [(a, b) for a, b in zip(run, run[1:]) if sum("sp2_" in r[2] for r in rows[a:b]) >= 4 and any(want in r[2] for r in rows[a:b]) and rows[b][0] - rows[a][0] < 6_000_000]   # SCF cycles only (not the back-to-back roofline launches)
for a, b in pairs[-50:]:
    seg = rows[a:b]
    tot += rows[b][0] - rows[a][0]; n += 1
    busy_end = seg[0][0]
    for s, e, k in seg:
        name = k.split("(")[0].replace("void ", "")
        name = "rocblas gemm" if name.startswith("Cijk") else name[:40]
        fam[name] += e - s
        if s > busy_end: idle += s - busy_end
        busy_end = max(busy_end, e)
    if rows[b][0] > busy_end: idle += rows[b][0] - busy_end
print(f"cycle {tot / n / 1e3:.1f} us, device idle {idle / n / 1e3:.1f} us per cycle; busy per kernel family (us per cycle):")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]): print(f"  {k:42s} {v / n / 1e3:8.1f}")
a, b = pairs[-1]
cnt = collections.Counter(k.split("(")[0].replace("void ", "")[:40] for s, e, k in rows[a:b])
print("launches in one cycle:", sum(cnt.values()), dict(cnt))
