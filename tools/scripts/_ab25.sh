mkdir -p gpurun_out
run() { # label env... 
  lab="$1"; shift
  env "$@" timeout 300 python bench.py --workload 10k-n1024 --steps 5 --warmup 2 --no-cpu --resident-only 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lab', r.get('kernel_gcups'), r.get('kernel_ms'), r.get('kernels'), d.get('parity_sample'))
except Exception as e: print('$lab', 'ERR', e)
"
}
run default A=1
run solo8-all KSW2AMD_SOLO=all
cp ksw2_amd/libksw2_amd.so /tmp/lib_keep.so
cp build_ab/lib_solo16.so ksw2_amd/libksw2_amd.so
run solo16-all KSW2AMD_SOLO=all
run solo16-default A=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "solo" 2>&1 | tail -3
cp /tmp/lib_keep.so ksw2_amd/libksw2_amd.so
