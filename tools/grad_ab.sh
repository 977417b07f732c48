#!/bin/bash
for o in "grad_nw=1" "grad_nw=2" "grad_nw=4" "grad_nw=1"; do echo "$o"; ERI_OPTS="$o" python tools/eri_bench.py ibuprofen def2-TZVP --grad --quiet 2>&1 | grep grad_eri_s; done
