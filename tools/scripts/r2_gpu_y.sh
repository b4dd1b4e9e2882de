#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'resident', d['value_hbm_resident'], 'ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
run cfg5 pinned-pool 5
run cfg3 pinned-pool 30
run 10k-cigar pinned-pool 12
run exts pinned-pool 8
run cfg4 pinned-pool 2
done
timeout 900 python -m pytest tests -m gpu -x -q -k "cigar or golden or cfg3 or reuse or thread" 2>&1 | tail -3
