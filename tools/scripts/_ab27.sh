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
run solo-all-ch4 KSW2AMD_SOLO=all
WL=cfg5 run cfg5-ch4 A=1
WL=10k EXTRA="--pairs 2048" run p2048-solo-ch4 KSW2AMD_SOLO=all
cp ksw2_amd/libksw2_amd.so /tmp/lib_keep.so
cp build_ab/lib_soloch8.so ksw2_amd/libksw2_amd.so
run solo-all-ch8 KSW2AMD_SOLO=all
WL=cfg5 run cfg5-ch8 A=1
WL=10k EXTRA="--pairs 2048" run p2048-solo-ch8 KSW2AMD_SOLO=all
cp /tmp/lib_keep.so ksw2_amd/libksw2_amd.so
