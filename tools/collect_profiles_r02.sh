#!/bin/bash
# Round-2 profile collection on one MI355X (run through gpurun from the repo root).  Kernel trace / stats and the PMC passes are
# separate rocprofv3 runs (the guide's recipe: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), program directly after `--`.
set -e
R=$PWD; O=$R/gpurun_out/prof_r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== bench (plain)"; python3 $R/bench.py --steps 50 --warmup 5 > $O/r02_bench.json 2> $O/r02_bench.err
echo "== bench kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/r02_bench_under_rocprof.json 2> $O/stats.err
for b in cc-pVTZ cc-pVDZ; do for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $b $c"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${b}_$c -o pmc -- python3 $R/tools/jk_once.py $b > $O/pmc_${b}_$c.out 2> $O/pmc_${b}_$c.err
done; done
find $O -name "*.csv" | head -50
