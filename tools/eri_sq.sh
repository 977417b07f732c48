#!/bin/bash
# SQ counter passes over the ERI preparation + derivative-ERI gradient of ibuprofen/def2-TZVP (round-2 kernels)
set -e
R=$PWD; O=$R/gpurun_out/eri_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/A -o pmc -- python3 $R/tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet > $O/A.out 2> $O/A.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/B -o pmc -- python3 $R/tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet > $O/B.out 2> $O/B.err
python3 $R/tools/sq_summarize.py $O/A $O/B $O/r02_pmc_eri_grad_ibuprofen.json
rm -rf $O/A $O/B
echo ok
