#!/bin/bash
# one box, alternating: geometry options of the row gradient kernel (ibuprofen/def2-TZVP, optimiser's screening threshold)
for rep in 1 2; do
for o in "grad_rows=0" "grad_rows_min=20,grad_rows_g32=0" "grad_rows_min=20,grad_rows_g32=1" "grad_rows_min=33,grad_rows_g32=1" "grad_rows_min=33,grad_rows_g32=0" "grad_rows_min=65" "grad_rows_min=4,grad_rows_g32=1" "grad_rows_min=4,grad_rows_g16=1"; do
  echo -n "$o  "; ERI_OPTS="grad_dtol=1e-10,$o" python tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet 2>&1 | grep grad_eri_s | tail -1
done; done
