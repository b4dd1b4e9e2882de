#!/bin/bash
# GPU box: default bench with the path-aware pipeline policy; fuzz soak (short + long modes) of the round's new kernels
mkdir -p gpurun_out/profiles
( timeout 1500 python bench.py > gpurun_out/r2l_bench.json 2> gpurun_out/r2l_bench.err ); echo "bench rc=$?" >> gpurun_out/r2l_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2l_bench.json').read().strip().splitlines()[-1])
a=d.pop('also',[])
print(d['value'], d['value_hbm_resident'], d['ms_per_step'], d['config']['host_pipeline'])
for x in a: print(x['workload'][:40], x.get('value'), x.get('value_hbm_resident'), x.get('error'))
PY
( env -u KSW2AMD_SIMDS timeout 400 python tools/scripts/fuzz_gpu.py 150 20260011 2>&1 | tail -4 ) > gpurun_out/r2l_fuzz_short.txt
( env -u KSW2AMD_SIMDS timeout 500 python tools/scripts/fuzz_gpu.py 200 20260012 long 2>&1 | tail -4 ) > gpurun_out/r2l_fuzz_long.txt
cat gpurun_out/r2l_fuzz_short.txt gpurun_out/r2l_fuzz_long.txt
