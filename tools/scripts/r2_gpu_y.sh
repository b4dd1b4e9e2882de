#!/bin/bash
mkdir -p gpurun_out/profiles
bash tools/scripts/profile_round.sh r2z exts 3 > gpurun_out/prof_r2z_exts.log 2>&1
tail -16 gpurun_out/prof_r2z_exts.log
timeout 900 python bench.py --workload exts --steps 10 --warmup 2 --cpu-seconds 8 --no-also 2> gpurun_out/bench_r2z_exts.err | tail -1 > gpurun_out/profiles/r2z_bench_exts.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/profiles/r2z_bench_exts.json').read())
print(d['value'], d['value_hbm_resident'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('gpu_over_cpu_1thread'))
PY
python tools/scripts/exts_classes.py 2>&1 | tail -20 > gpurun_out/profiles/r2z_exts_classes.txt; cat gpurun_out/profiles/r2z_exts_classes.txt
