#!/bin/bash
set -e
R=$PWD; O=$R/gpurun_out/eri_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ERI_OPTS="xf_stage=0" rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o pmc -- python3 $R/tools/eri_once.py ibuprofen def2-TZVP > $O/$c.out 2> $O/$c.err
done
echo ok
