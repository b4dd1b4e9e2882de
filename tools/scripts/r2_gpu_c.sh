#!/bin/bash
# GPU box, round 2 third pass: staged traceback stores (parity, resident rates, PMC of config 3), chunk policy grid of the headline
mkdir -p gpurun_out/profiles
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r2c_pytest.log
for w in cfg3 10k-cigar cfg5 cfg2 10k; do
	timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu --resident-only 2>/dev/null | tail -1 > gpurun_out/r2c_res_$w.json
done
bash tools/scripts/profile_round.sh r2c cfg3 3 > gpurun_out/r2c_prof_cfg3.log 2>&1
for cfg in "64 4 1 1" "64 4 0 0" "128 4 1 0" "128 4 0 0" "256 4 1 0" "128 2 1 0" "128 8 1 0" "256 8 1 0"; do
	set -- $cfg
	echo "CHUNK_MB=$1 THREADS=$2 DBUF=$3 RAMP=$4" >> gpurun_out/r2c_grid.txt
	KSW2AMD_CHUNK_MB=$1 KSW2AMD_THREADS=$2 KSW2AMD_DBUF=$3 KSW2AMD_RAMP=$4 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_hbm_resident'], d['ms_per_step'], d['config']['host_pipeline'])" >> gpurun_out/r2c_grid.txt
done
tail -4 gpurun_out/r2c_pytest.log; cat gpurun_out/r2c_grid.txt; for w in cfg3 10k-cigar cfg5; do python -c "import json,sys; d=json.loads(open('gpurun_out/r2c_res_$w.json').read()); print('$w', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'], d['roofline']['fill_kernel_ms'])"; done; tail -12 gpurun_out/r2c_prof_cfg3.log
