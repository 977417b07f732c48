#!/bin/bash
set -e
R=$PWD; O=$R/gpurun_out/cyc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o b -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/err
python3 $R/tools/cycle_trace.py $(find $O/t -name "*kernel_trace.csv") | tee $O/r03_cycle_timeline.txt
python3 $R/tools/cycle_trace.py $(find $O/t -name "*kernel_trace.csv") rks | tee $O/r03_cycle_timeline_rks.txt
rm -rf $O/t
