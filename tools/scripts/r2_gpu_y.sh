#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu --no-also $4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 steps=$3 e2e', d['value'], 'resident', d['value_hbm_resident'], 'ms/step', d['ms_per_step'], d['config']['host_pipeline'])"; }
for rep in 1 2; do
run 10k default 8
run 10k approx 8 --approx
run cfg2 default 40
run cfg3 default 30
run 10k-cigar default 12
run cfg5 default 5
run 10k-n1024 default 20
run cfg4 default 2
run exts default 5
run extf default 10
done
timeout 900 python -m pytest tests -m gpu -x -q -k "pipeline or pool or fuzz or ragged or cfg3 or golden" 2>&1 | tail -3
