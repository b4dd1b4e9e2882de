#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'resident', d['value_hbm_resident'], 'ms/step', d['ms_per_step'])"; }
KSW2AMD_TRACE=2 timeout 600 python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu --no-also 2>&1 | grep "plan_create\|serial plan" | tail -4
for rep in 1 2; do
run cfg5 parcopy 5
KSW2AMD_NO_PARCOPY=1 run cfg5 no-parcopy 5
run 10k-n1024 parcopy 20
run cfg4 parcopy 2
done
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg5 or ragged or ont or fuzz" 2>&1 | tail -3
