mkdir -p gpurun_out
run() { lab="$1"; shift
  env "$@" timeout 300 python bench.py --workload ${WL:-10k-n1024} --steps 5 --warmup 2 --no-cpu --resident-only ${EXTRA} 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lab', '${WL:-10k-n1024}', r.get('kernel_gcups'), r.get('kernel_ms'), r.get('kernels'))
except Exception as e: print('$lab', 'ERR', e)
"
}
run n1024-solo KSW2AMD_SOLO=all
run n1024-pk A=1
for w in 10k cfg2 cfg3 cfg5 10k-cigar cfg4; do WL=$w run $w A=1; done
WL=cfg2 run cfg2-defer KSW2AMD_DEFER=1
WL=cfg2 run cfg2-168 KSW2AMD_PK_FIRST=1
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
