#!/bin/bash
for o in "grad_dtol=1e-13" "grad_dtol=1e-11" "grad_dtol=1e-10" "grad_dtol=1e-9" "grad_dtol=1e-8"; do echo "$o"; ERI_OPTS="$o" python tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet 2>&1 | grep grad_eri_s; done
