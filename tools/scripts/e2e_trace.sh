# end-to-end numbers of the configurations (GPU box):  bash tools/scripts/e2e_trace.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $W $*"; env "$@" timeout 300 python bench.py --workload $W --steps $S --warmup 5 --no-cpu --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('value_flat_arena'), d.get('value_hbm_resident'), d['ms_per_step'], d.get('parity_sample'))"; }
S=80
for W in cfg2; do run A=1; run KSW2AMD_SHARED_UP_STREAMS=2; run KSW2AMD_SHARED_UP_STREAMS=3; run HSA_ENABLE_SDMA=0; run A=1; run KSW2AMD_SHARED_UP_STREAMS=2; done
S=40
for W in cfg3; do run A=1; run KSW2AMD_SHARED_UP_STREAMS=2; done
S=12
for W in 10k 10k-cigar cfg5; do run A=1; run KSW2AMD_SHARED_UP_STREAMS=2; done
