#!/bin/bash
set -e
R=$PWD; O=$R/gpurun_out/rks; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o b -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/err
python3 - <<PY
import csv, glob
f = glob.glob("$O/t/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print(r["Name"][:86], r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["MaxNs"])/1e3,1), round(float(r["TotalDurationNs"])/1e6,2))
PY
rm -rf $O/t
