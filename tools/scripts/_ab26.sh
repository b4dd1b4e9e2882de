mkdir -p gpurun_out
run() { lab="$1"; shift
  env "$@" timeout 300 python bench.py --workload ${WL:-10k-n1024} --steps 5 --warmup 2 --no-cpu --resident-only ${EXTRA} 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lab', '${WL:-10k-n1024}', r.get('kernel_gcups'), r.get('kernel_ms'), r.get('kernels'))
except Exception as e: print('$lab', 'ERR', e)
"
}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solo" 2>&1 | tail -3
run default A=1
run solo-all KSW2AMD_SOLO=all
WL=cfg5 run cfg5-default A=1
WL=10k EXTRA="--pairs 2048" run p2048-default A=1
WL=10k EXTRA="--pairs 2048" run p2048-solo KSW2AMD_SOLO=all
WL=10k EXTRA="--pairs 4096" run p4096-default A=1
WL=10k EXTRA="--pairs 4096" run p4096-solo KSW2AMD_SOLO=all
