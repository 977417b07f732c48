#!/usr/bin/env python3
"""Sum one rocprofv3 --pmc counter per kernel for two runs ("new" and "old" directories) -> JSON.
   python3 tools/pmc_sum.py WRITE_SIZE <dir_new> <dir_old> <out.json>"""
import csv, collections, glob, json, os, sys
name, out = sys.argv[1], {}
for tag, d in (("new", sys.argv[2]), ("old", sys.argv[3])):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0, 0])
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k]
        a[0] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); a[1] += 1; a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = sorted(((v[0], k, v[1], v[2]) for k, v in agg.items()), reverse=True)
    stored = None
    try:
        stored = int(open(d.rstrip("/") + ".out").read().split("stored_bytes")[1].split()[0])
    except Exception:
        pass
    # WRITE_SIZE / FETCH_SIZE are reported in KiB-like units of the profiler's derived metric: keep the raw sums and the ratio
    eri = [r for r in rows if r[1].startswith("eri_")]
    out[tag] = {"stored_bytes": stored, "counter": name, "sum_all_eri_kernels": sum(r[0] for r in eri),
                "kernels": [dict(kernel=k, value=v, launches=n, total_ms=round(ns / 1e6, 3)) for v, k, n, ns in rows if v > 0][:24]}
json.dump(out, open(sys.argv[4], "w"), indent=1)
for tag in out:
    print(tag, out[tag]["stored_bytes"], out[tag]["sum_all_eri_kernels"])
