#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'resident', d['value_hbm_resident'], 'ms/step', d['ms_per_step'], d['config']['host_pipeline'])"; }
for rep in 1 2 3; do
KSW2AMD_RAMP=1 run 10k ramp 10
KSW2AMD_RAMP=0 run 10k noramp 10
done
KSW2AMD_RAMP=1 run 10k "ramp approx" 10 --approx
KSW2AMD_RAMP=0 run 10k "noramp approx" 10 --approx
