# End-to-end A/B of the bench workloads under environment switches, all in ONE gpurun call (boxes of the pool differ by 10-20 %):
#   bash tools/scripts/e2e_trace.sh "cfg2 cfg3" "A=1" "KSW2AMD_SHARED_UP_MIN_MB=16" ...      (A=1 = no switch)
# prints value | value_flat_arena | value_hbm_resident | ms_per_step | parity_sample per run (profiles/r4_e2e_ab.txt).
cd $GRAFT_REPO_ROOT
WORKLOADS=${1:-"cfg2 cfg3"}; shift
[ $# -eq 0 ] && set -- "A=1"
run() { echo "== $W $*"; env $* timeout 300 python bench.py --workload $W --steps ${S:-40} --warmup 5 --no-cpu --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('value_flat_arena'), d.get('value_hbm_resident'), d['ms_per_step'], d.get('parity_sample'))"; }
for W in $WORKLOADS; do for e in "$@"; do run $e; done; done
