# sweep one ERI tunable on ibuprofen/def2-TZVP:  bash tools/eri_sweep.sh xf_mfma_min 0 150 300 600 1200
key=$1; shift
for v in "$@"; do echo "$key=$v"; ERI_OPTS="$key=$v" MI355_DEBUG=1 python tools/eri_bench.py ibuprofen def2-TZVP 2>&1 | grep "quartet evaluation" | tail -1; done
