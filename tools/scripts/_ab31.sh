mkdir -p gpurun_out
run() { lab="$1"; shift
  env "$@" timeout 300 python bench.py --workload ${WL:-10k-n1024} --steps 5 --warmup 2 --no-cpu --resident-only ${EXTRA} 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lab', '${WL:-10k-n1024}', r.get('kernel_gcups'), r.get('kernel_ms'), r.get('kernels'))
except Exception as e: print('$lab', 'ERR', e)
"
}
run n1024 A=1
for w in cfg5 cfg3; do WL=$w run $w A=1; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload 10k-n1024 --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_n1024.json; python -c "
import json; d=json.load(open('gpurun_out/bench_n1024.json')); print(d['value'], d.get('value_flat_arena'), d.get('parity_sample'), d['roofline']['frac'], d['cpu_baseline'])"
