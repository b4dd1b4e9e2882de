#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of one bench.py workload, then the
# summaries into profiles/ (copied back through gpurun_out/profiles/).
#   usage: tools/scripts/profile_round.sh <tag> <workload> [steps]
# PMC passes never carry another trace domain than --kernel-trace (pool rule), and the program follows `--` directly.
set -u
TAG=$1; WL=$2; STEPS=${3:-5}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/gpurun_out/profiles"
BENCH="python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu --resident-only"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
P=0
for CNT in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
	P=$((P+1))
	rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/pmc$P" -- $BENCH > "$OUT/pmc$P.log" 2>&1
done
cd "$ROOT"
python3 tools/scripts/pmc_summary.py "$TAG" "$WL" "$OUT" "$ROOT/gpurun_out/profiles" "$STEPS"
