#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'resident', d['value_hbm_resident'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2 3; do run cfg2 pinned-res 60; done
run 10k pinned-res 8
run cfg3 pinned-res 30
run extf pinned-res 10
timeout 300 tools/coalesce-bench 1 2000 512 64 0 2>&1 | tail -1
timeout 300 tools/coalesce-bench 64 2000 512 64 0 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "thread or coalesc or reuse or golden or dropin or cli" 2>&1 | tail -2
