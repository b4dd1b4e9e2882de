#!/bin/bash
mkdir -p gpurun_out/profiles
( timeout 900 python -m pytest tests -m gpu -x -q -k "packed_generation or mt_pair or cfg4 or 50k or fuzz" 2>&1 | tail -6 ) > gpurun_out/r2q_pytest.log
tail -3 gpurun_out/r2q_pytest.log
bash tools/scripts/profile_round.sh r2q cfg4 3 > gpurun_out/prof_r2q_cfg4.log 2>&1
tail -14 gpurun_out/prof_r2q_cfg4.log
for w in cfg4 cfg4-so; do timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'])"; done
