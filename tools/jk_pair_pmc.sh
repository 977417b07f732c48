#!/bin/bash
set -e
R=$PWD; O=$R/gpurun_out/pairpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o pmc -- python3 $R/tools/jk_pair_bench.py cc-pVTZ > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/f/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "jk_tiles" in r["Kernel_Name"]: d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in d.items(): print(k, len(v), "FETCH_SIZE x2 per launch: %.3f GB" % (2 * 1024 * sum(v) / len(v) / 1e9))
PY
rm -rf $O/f
