#!/bin/bash
run() { timeout 900 python bench.py --workload $1 --steps $3 --warmup 1 --no-cpu --no-also $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'resident', d['value_hbm_resident'], 'ms/step', d['ms_per_step'], d['config']['host_pipeline'])"; }
KSW2AMD_TRACE=1 timeout 900 python bench.py --workload cfg4 --steps 1 --warmup 1 --no-cpu --no-also 2>&1 | grep "ksw2_amd\]" | tail -12
run cfg4 default 2
run cfg5 default 4
timeout 1200 python -m pytest tests -m gpu -x -q -k "cfg4 or mt_pair or 50k or split or memory or generation" 2>&1 | tail -3
