for mb in 0 96 160 224; do echo "cache_mb=$mb"; JK_OPTS="jk_cache_mb=$mb" python tools/jk_bench.py cc-pVTZ 2>/dev/null; done
