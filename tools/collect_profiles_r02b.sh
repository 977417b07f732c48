#!/bin/bash
# Round-2 profile collection, part 2: BASELINE configs 3/4/5 run logs + ibuprofen kernel stats + per-row component figures.
set -e
R=$PWD; O=$R/gpurun_out/prof_r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== config3"; python3 $R/tools/run_config.py benzene cc-pVTZ B3LYP --grad > $O/r02_config3_benzene_b3lyp_ccpvtz.log 2>&1
echo "== config5 single point + gradient under rocprof"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ibu -o ibu -- python3 $R/tools/run_config.py ibuprofen def2-TZVP B3LYP --grad > $O/r02_config5_ibuprofen_b3lyp_def2tzvp_grad.log 2>&1
echo "== components"; python3 $R/tools/bench_components.py > $O/r02_components.jsonl 2> $O/components.err
echo "== config5 opt"; python3 $R/tools/run_config.py ibuprofen def2-TZVP B3LYP --opt=150 > $O/r02_config5_ibuprofen_b3lyp_def2tzvp_opt.log 2>&1
echo "== config4 (1 GPU direct)"; timeout -k 10 300 python3 $R/tools/run_config.py c60 "6-31G*" HF > $O/r02_config4_c60_rhf_631gs_direct_1gpu.log 2>&1
echo done
