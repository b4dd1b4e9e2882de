#!/bin/bash
# GPU box: key atomics at workgroup scope (config 4 write traffic and rate), lane-form X-drop test, packed generation-serial parity
mkdir -p gpurun_out/profiles
( timeout 900 python -m pytest tests -m gpu -x -q -k "one_extension or packed_generation or mt_pair or cfg4 or 50k or solo_kernel" 2>&1 | tail -6 ) > gpurun_out/r2n_pytest.log
tail -3 gpurun_out/r2n_pytest.log
bash tools/scripts/profile_round.sh r2n cfg4 3 > gpurun_out/prof_r2n_cfg4.log 2>&1
tail -14 gpurun_out/prof_r2n_cfg4.log
timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'])"
