# A/B of the thread-per-quartet ERI kernels on several tensors
for mol in "benzene cc-pVTZ" "benzene cc-pVDZ" "ibuprofen def2-TZVP"; do for v in 0 1; do echo "$mol eri_tpq=$v"; ERI_OPTS="eri_tpq=$v" MI355_DEBUG=1 python tools/eri_bench.py $mol 2>&1 | grep -E "quartet evaluation" | tail -1; done; done
