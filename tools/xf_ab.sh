#!/bin/bash
# A/B on one box: previous build vs transform kernel with the tile-base table (xf_stage=0), staged M, nontemporal stores
for rep in 1 2; do
echo "prev build"; MI355SCF_LIB=computational-chemistry-ai_amd/csrc/libmi355scf_prev.so python tools/eri_bench.py ibuprofen def2-TZVP 2>&1 | grep -E "quartet evaluation" | tail -1
for o in "xf_stage=0" "xf_stage=0,xf_nt=1" "xf_stage=12"; do echo "$o"; ERI_OPTS="$o" python tools/eri_bench.py ibuprofen def2-TZVP 2>&1 | grep -E "quartet evaluation" | tail -1; done
done
