#!/bin/bash
# A/B on one box: triangular rows in block-diagonal tiles (tri_tiles=1, default) vs the full-row layout of round 1
for rep in 1 2; do for t in 0 1; do echo "tri_tiles=$t"; JK_OPTS="tri_tiles=$t" python tools/jk_bench.py cc-pVDZ cc-pVTZ; done; done
