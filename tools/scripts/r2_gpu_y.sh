#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'resident', d['value_hbm_resident'], 'ms/step', d['ms_per_step'], d['config']['host_pipeline'])"; }
for rep in 1 2; do
KSW2AMD_ISSUE=0 run 10k pooled 10
KSW2AMD_ISSUE=1 run 10k "issued x2" 10
KSW2AMD_ISSUE=1 KSW2AMD_ISSUE_X2=0 run 10k "issued x1" 10
done
KSW2AMD_ISSUE=1 run 10k "issued x2 approx" 10 --approx
KSW2AMD_ISSUE=0 run 10k "pooled approx" 10 --approx
KSW2AMD_ISSUE=1 KSW2AMD_TRACE=1 timeout 600 python bench.py --workload 10k --steps 1 --warmup 1 --no-cpu --no-also 2>&1 | grep "ksw2_amd\]" | tail -8
timeout 600 python -m pytest tests -m gpu -x -q -k "10k" 2>&1 | tail -2
