#!/bin/bash
# Round-3 profile collection on one MI355X (gpurun from the repo root).  Kernel trace / stats and every PMC pass are separate
# rocprofv3 runs, program directly after `--`.  Summaries land in gpurun_out/prof_r03 and are copied into profiles/ by hand.
set -e
R=$PWD; O=$R/gpurun_out/prof_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== bench under kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-wall-clock --no-scale-leg > $O/r03_bench_under_rocprof.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
echo "== ERI evaluation: WRITE_SIZE, round-3 order vs round-2 order"
for v in new old; do
  if [ $v = old ]; then export ERI_OPTS="ao_order=0,ket_cluster=0,xcd_map=0"; else export ERI_OPTS=""; fi
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w_$v -o pmc -- python3 $R/tools/eri_once.py ibuprofen def2-TZVP > $O/w_$v.out 2> $O/w_$v.err
done
unset ERI_OPTS
python3 $R/tools/pmc_sum.py WRITE_SIZE $O/w_new $O/w_old $O/r03_pmc_eri_write.json
rm -rf $O/w_new $O/w_old
echo "== ERI + gradient: SQ counters"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/A -o pmc -- python3 $R/tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet > $O/A.out 2> $O/A.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/B -o pmc -- python3 $R/tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet > $O/B.out 2> $O/B.err
python3 $R/tools/sq_summarize.py $O/A $O/B $O/r03_pmc_eri_grad_ibuprofen.json
rm -rf $O/A $O/B $O/stats
echo "== ibuprofen kernel stats (ERI + gradient)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ibu -o ibu -- python3 $R/tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet > $O/ibu.out 2> $O/ibu.err
cp $(find $O/ibu -name "*kernel_stats.csv" | head -1) $O/r03_ibuprofen_eri_grad_kernel_stats.csv
rm -rf $O/ibu
echo ok
