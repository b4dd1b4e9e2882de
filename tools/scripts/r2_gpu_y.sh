#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "sse" 2>&1 | tail -2
for rep in 1 2; do
timeout 600 python bench.py --workload cfg2 --sse-compat --pairs 16384 --steps 5 --warmup 1 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 sse e2e', d['value'], 'resident', d['value_hbm_resident'], d['roofline']['kernel_ms'])"
done
timeout 600 python bench.py --workload cfg3 --sse-compat --pairs 4096 --steps 3 --warmup 1 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 sse e2e', d['value'], 'resident', d['value_hbm_resident'], d['roofline']['kernel_ms'])"
env -u KSW2AMD_SSEC_HBM timeout 200 python tools/scripts/fuzz_gpu.py 60 20260072 2>&1 | tail -2
