#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'ms/step', d['ms_per_step'], d['config']['host_pipeline'])"; }
for wl in exts extf; do
for tk in "6 0" "6 4" "6 6" "6 8" "8 8" "6 16" "12 16"; do set -- $tk
	if [ $2 = 0 ]; then KSW2AMD_THREADS=$1 run $wl "T=$1 default" 8; KSW2AMD_NO_UNITS=1 KSW2AMD_THREADS=$1 run $wl "T=$1 default no-units" 8
	else KSW2AMD_THREADS=$1 KSW2AMD_CHUNKS=$2 run $wl "T=$1 K=$2" 8; fi
done
done
