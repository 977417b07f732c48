#!/usr/bin/env python3
"""Post-processing of tools/collect_profiles_r02.sh output (gpurun_out/prof_r02) into the committed files under profiles/:
J/K rows of the PMC passes, the corrected per-launch traffic file bench.py reads, kernel stats and the bench JSON lines."""
import csv, collections, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof_r02")
OUT = os.path.join(ROOT, "profiles")
cases = {}
for b in ("cc-pVTZ", "cc-pVDZ"):
    txt = open(f"{P}/pmc_{b}_FETCH_SIZE.out").read()
    m = re.search(r"alg_bytes J\+K (\d+) J only (\d+)", txt)
    alg = {"J+K": int(m.group(1)), "J only": int(m.group(2))}
    stored = int(re.search(r"stored_bytes (\d+)", txt).group(1))
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = [r for r in csv.DictReader(open(f"{P}/pmc_{b}_{c}/pmc_counter_collection.csv")) if "jk_tiles" in r["Kernel_Name"]]
        with open(f"{OUT}/r02_pmc_{b}_{c}.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
        d = collections.defaultdict(list)
        for r in rows:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in d.items():
            var = "J+K" if ("<true, true" in k or "pipe" in k) else "J only"
            vals.setdefault(var, {})[c] = (sum(v) / len(v), k.replace("void ", "").replace("(JkArgs)", "").replace(", ", ","))
    for var, v in vals.items():
        f, k = v["FETCH_SIZE"]; w, _ = v["WRITE_SIZE"]
        cases[f"benzene/{b} {var}"] = {"kernel": k, "FETCH_SIZE_KB_per_launch": round(f, 2), "WRITE_SIZE_KB_per_launch": round(w, 2),
                                       "hbm_read_bytes": 2 * f * 1024, "hbm_write_bytes_atomics": w * 1024,
                                       "traffic_bytes": 2 * f * 1024 + w * 1024, "algorithmic_bytes": alg[var], "stored_bytes": stored}
note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, --output-format csv) on the J/K digestion "
        "kernel, 5 launches per variant, tools/jk_once.py via tools/collect_profiles_r02.sh; counter values are KB as reported; gfx950 "
        "correction (MI355X_MICROARCH.md HBM section / cdna_hip_programming.md section 7): FETCH_SIZE counts 1/2 of a wide coalesced "
        "stream -> hbm_read = 2*FETCH_SIZE*1024; WRITE_SIZE exact for atomics.  The cc-pVDZ tensor fits the 256 MiB Infinity Cache, "
        "whose hits FETCH_SIZE also counts.  Kernels of the final round-2 state (triangular rows in block-diagonal tiles, "
        "nontemporal stream + default-policy prefix of jk_cache_mb = 160 MiB).")
json.dump({"note": note, "cases": cases}, open(f"{OUT}/r02_pmc_jk_traffic.json", "w"), indent=1)
for k, v in cases.items():
    print(k, v["kernel"], "traffic/algorithmic", round(v["traffic_bytes"] / v["algorithmic_bytes"], 4), "read/stored", round(v["hbm_read_bytes"] / v["stored_bytes"], 4))
shutil.copy(f"{P}/stats/bench_kernel_stats.csv", f"{OUT}/r02_bench_kernel_stats.csv")
shutil.copy(f"{P}/r02_bench.json", f"{OUT}/r02_bench.json")
shutil.copy(f"{P}/r02_bench_under_rocprof.json", f"{OUT}/r02_bench_under_rocprof.json")
d = json.load(open(f"{OUT}/r02_bench.json"))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["traffic"], d["secondary"]["value"], d["cpu_baseline"]["value"])
d = json.load(open(f"{OUT}/r02_bench_under_rocprof.json"))
print("under rocprof", d["value"], d["roofline"]["ms_per_launch"])
for r in csv.DictReader(open(f"{OUT}/r02_bench_kernel_stats.csv")):
    if "jk_tiles" in r["Name"] or "sp2_fused" in r["Name"]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
