#!/bin/bash
# PMC of the SSE-compatible kernel on config 2's shape (16 384 pairs)
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r2z_ssec; rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/profiles
BENCH="python3 $ROOT/bench.py --workload cfg2 --sse-compat --pairs 16384 --steps 3 --warmup 1 --no-cpu --resident-only"
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
python3 tools/scripts/pmc_summary.py r2z cfg2-ssec "$OUT" "$ROOT/gpurun_out/profiles" 3
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/profiles/r2z_cfg2-ssec_pmc.json'))
dom=d['derived']['dominant_kernel']; c=d['per_launch'][dom]; w=c['SQ_WAVES']
print(dom, d['derived'])
for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_WAVE_CYCLES'): print(k, round(c[k]/w))
PY
