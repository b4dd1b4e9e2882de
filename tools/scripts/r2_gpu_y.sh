#!/bin/bash
for rep in 1 2; do
for t in 0 1; do
export KSW2AMD_NO_TALL=$t; [ $t = 0 ] && unset KSW2AMD_NO_TALL
timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 2 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no_tall=$t cfg2 resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout 600 python bench.py --workload cfg2 --approx --steps 20 --warmup 2 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no_tall=$t cfg2 approx resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'])"
done
done
unset KSW2AMD_NO_TALL
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg2 or golden or ragged or fuzz or approx" 2>&1 | tail -2
timeout 200 python tools/scripts/fuzz_gpu.py 90 20260091 2>&1 | tail -1
