#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs (one row per dispatch and counter) per kernel -> JSON; used on the GPU box so that
only the summary travels back.   python3 tools/sq_summarize.py <dirA> <dirB> <out.json>"""
import csv, collections, json, sys, glob, os
out = {}
for d in sys.argv[1:-1]:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: {"launches": set(), "ns": 0, "counters": collections.defaultdict(float)})
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg[k]
        a["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); a["launches"].add(r["Dispatch_Id"])
            a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, a in agg.items():
        e = out.setdefault(k, {"counters": {}})
        e.setdefault("launches", len(a["launches"])); e.setdefault("total_ms", round(a["ns"] / 1e6, 3))
        e["counters"].update({c: v for c, v in a["counters"].items()})
rows = []
for k, e in out.items():
    c = e["counters"]
    if "SQ_WAVE_CYCLES" in c and e["total_ms"] > 0:
        flops = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0)) + 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)
        e["fp64_lane_ops_upper_TFLOPs"] = round(flops / (e["total_ms"] * 1e-3) / 1e12, 2)
        e["frac_of_78p6"] = round(e["fp64_lane_ops_upper_TFLOPs"] / 78.6, 3)
        if c.get("SQ_ACTIVE_INST_LDS"):
            e["lds_bank_conflict_over_lds_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_ACTIVE_INST_LDS"], 3)
    if c.get("SQ_BUSY_CYCLES") and "SQ_WAIT_ANY" in c:
        wc = c.get("SQ_WAIT_ANY", 0) + c.get("SQ_WAIT_INST_ANY", 0) + c.get("SQ_ACTIVE_INST_ANY", 0)
        if wc > 0:
            e["wave_cycle_shares"] = {"parked_waitcnt_or_barrier": round(c["SQ_WAIT_ANY"] / wc, 3), "issue_stall": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                                      "issuing": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3), "issuing_valu": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)}
    rows.append((e["total_ms"], k))
res = {"kernels": [dict(kernel=k, **out[k]) for _, k in sorted(rows, reverse=True) if out[k]["total_ms"] >= 1.0]}
json.dump(res, open(sys.argv[-1], "w"), indent=1)
for e in res["kernels"][:16]:
    print(e["kernel"][:48], e["launches"], e["total_ms"], e.get("frac_of_78p6"), e.get("lds_bank_conflict_over_lds_active"), e.get("wave_cycle_shares"))
