#!/bin/bash
# MFMA / issue counters of the dense FP64-MFMA kernels of the SCF cycle (xc_vmat_kernel, sp2_plan_kernel) and of the J/K kernel,
# from two rocprofv3 --pmc passes over bench.py's RHF + B3LYP legs ("no MFMA-busy counter was collected").
set -e
R=$PWD; O=$R/gpurun_out/prof_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/MA -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wall-clock --no-scale-leg > $O/MA.out 2> $O/MA.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/MB -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wall-clock --no-scale-leg > $O/MB.out 2> $O/MB.err
python3 $R/tools/sq_summarize.py $O/MA $O/MB $O/r03_pmc_scf_cycle_kernels.json
rm -rf $O/MA $O/MB
echo ok
