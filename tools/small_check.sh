#!/bin/bash
python -m pytest tests/test_gpu_scf.py -x -q 2>&1 | tail -2
python bench.py --basis cc-pVDZ --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cc-pVDZ', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])"
