#!/bin/bash
# A/B of two builds on the same box: tools/probes/libmi355scf_old.so (copy of the previous build) against the current one,
# gradient of ibuprofen/def2-TZVP at the optimiser's screening threshold, alternating
for rep in 1 2 3; do
  for lib in tools/probes/libmi355scf_old.so ""; do
    echo -n "${lib:-current} "
    MI355SCF_LIB=$lib ERI_OPTS="grad_dtol=1e-10" python tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet 2>&1 | grep grad_eri_s | tr '\n' ' '; echo
  done
done
