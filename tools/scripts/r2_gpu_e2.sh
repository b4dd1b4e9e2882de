#!/bin/bash
# PMC of the X-drop lane form (one extension per lane) on a big batch
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r2z_extflane; rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/profiles
BENCH="python3 $ROOT/bench.py --workload extf --pairs 262144 --steps 3 --warmup 1 --no-cpu --resident-only"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
P=0
for CNT in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
	P=$((P+1))
	rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/pmc$P" -- $BENCH > "$OUT/pmc$P.log" 2>&1
done
cd "$ROOT"
python3 tools/scripts/pmc_summary.py r2z extf-lane "$OUT" "$ROOT/gpurun_out/profiles" 3
tail -30 gpurun_out/profiles/r2z_extf-lane_pmc.json
