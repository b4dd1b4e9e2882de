#!/bin/bash
# GPU box: packed generation-serial class (parity first, time-boxed), config 4 resident rate with and without it, default bench
mkdir -p gpurun_out/profiles
( timeout 600 python -m pytest tests -m gpu -x -q -k "packed_generation_serial or mt_pair or wide_band or very_long or 50k or row_state" 2>&1 | tail -15 ) > gpurun_out/r2f_pytest.log
tail -5 gpurun_out/r2f_pytest.log
if grep -q "passed" gpurun_out/r2f_pytest.log && ! grep -q "failed" gpurun_out/r2f_pytest.log; then
	timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 > gpurun_out/r2f_res_cfg4.json
	KSW2AMD_NO_PKMP=1 timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 > gpurun_out/r2f_res_cfg4_int32.json
	for f in gpurun_out/r2f_res_cfg4.json gpurun_out/r2f_res_cfg4_int32.json; do python -c "import json,sys; d=json.loads(open('$f').read()); print('$f', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'], d['roofline']['fill_kernel_ms'])"; done
fi
( timeout 1500 python bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err ); echo "bench rc=$?" >> gpurun_out/r2f_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2f_bench.json').read().strip().splitlines()[-1])
a=d.pop('also',[])
print(d['value'], d['value_hbm_resident'], d['ms_per_step'], d['config']['host_pipeline'])
for x in a: print(x['workload'][:40], x.get('value'), x.get('value_hbm_resident'), x.get('error'))
PY
