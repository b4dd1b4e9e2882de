#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'ms/step', d['ms_per_step'], d['config']['host_pipeline'])"; }
run exts default 8
KSW2AMD_THREADS=8 run exts T=8 8
KSW2AMD_THREADS=12 KSW2AMD_CHUNKS=16 run exts "T=12 K=16" 8
KSW2AMD_TRACE=1 KSW2AMD_THREADS=6 timeout 600 python bench.py --workload exts --steps 2 --warmup 1 --no-cpu --no-also 2>&1 | grep "ksw2_amd\]" | tail -8
timeout 900 python -m pytest tests -m gpu -x -q -k "splice or exts" 2>&1 | tail -3
