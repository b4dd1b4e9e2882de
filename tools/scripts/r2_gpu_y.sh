#!/bin/bash
for rep in 1 2; do
for f in 0 1; do
KSW2AMD_PK_FIRST=$f timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 2 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pk_first=$f cfg2 resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
done
