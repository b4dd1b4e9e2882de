#!/bin/bash
# GPU box: full parity suite, fuzz soak, default bench -- the state that goes into the round's profiles
mkdir -p gpurun_out/profiles
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r2r_pytest.log
tail -3 gpurun_out/r2r_pytest.log
( env -u KSW2AMD_SIMDS timeout 300 python tools/scripts/fuzz_gpu.py 120 20260081 2>&1 | tail -3 ) > gpurun_out/r2r_fuzz.txt
( env -u KSW2AMD_SIMDS timeout 400 python tools/scripts/fuzz_gpu.py 180 20260082 long 2>&1 | tail -3 ) >> gpurun_out/r2r_fuzz.txt
cat gpurun_out/r2r_fuzz.txt
( timeout 1500 python bench.py > gpurun_out/r2r_bench.json 2> gpurun_out/r2r_bench.err ); echo "bench rc=$?" >> gpurun_out/r2r_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2r_bench.json').read().strip().splitlines()[-1])
a=d.pop('also',[])
print(d['value'], d['value_hbm_resident'], d['ms_per_step'], d['config']['host_pipeline'])
for x in a: print(x['workload'][:40], x.get('value'), x.get('value_hbm_resident'), x.get('error'))
PY
