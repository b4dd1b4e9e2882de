#!/bin/bash
# GPU box: full parity suite (4-bit packed traceback codes, packed generation-serial class), config 4 resident rates, default bench
mkdir -p gpurun_out/profiles
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r2j_pytest.log
tail -4 gpurun_out/r2j_pytest.log
timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu --resident-only 2> gpurun_out/r2j_cfg4.err | tail -1 > gpurun_out/r2j_res_cfg4.json
KSW2AMD_NO_PKMP=1 timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 > gpurun_out/r2j_res_cfg4_int32.json
timeout 600 python bench.py --workload 10k-cigar --steps 5 --warmup 2 --no-cpu --resident-only 2>/dev/null | tail -1 > gpurun_out/r2j_res_10k-cigar.json
for f in cfg4 cfg4_int32 10k-cigar; do python -c "import json,sys; d=json.loads(open('gpurun_out/r2j_res_$f.json').read()); print('$f', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'], d['roofline']['fill_kernel_ms'])" 2>&1 | tail -1; done
tail -3 gpurun_out/r2j_cfg4.err
( timeout 1500 python bench.py > gpurun_out/r2j_bench.json 2> gpurun_out/r2j_bench.err ); echo "bench rc=$?" >> gpurun_out/r2j_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2j_bench.json').read().strip().splitlines()[-1])
a=d.pop('also',[])
print(d['value'], d['value_hbm_resident'], d['ms_per_step'], d['config']['host_pipeline'])
for x in a: print(x['workload'][:40], x.get('value'), x.get('value_hbm_resident'), x.get('error'))
PY
