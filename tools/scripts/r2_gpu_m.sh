#!/bin/bash
# GPU box: extf one-extension-per-lane class (parity through every golden case + fuzz slice, rates by batch size), solo kernel with
# staged traceback stores, write traffic of 10k-cigar / cfg5 / cfg4
mkdir -p gpurun_out/profiles
( KSW2AMD_EXTF_LANE=1 timeout 900 python -m pytest tests -m gpu -x -q -k "linear or extf or solo or cfg5 or cli" 2>&1 | tail -6 ) > gpurun_out/r2m_pytest.log
tail -3 gpurun_out/r2m_pytest.log
( timeout 900 python -m pytest tests -m gpu -x -q -k "linear or extf or solo or cfg5 or fuzz" 2>&1 | tail -6 ) >> gpurun_out/r2m_pytest.log
tail -3 gpurun_out/r2m_pytest.log
timeout 900 python tools/scripts/extf_lane_probe.py > gpurun_out/profiles/r2_extf_lane.txt 2>&1
cat gpurun_out/profiles/r2_extf_lane.txt
export TMPDIR=/tmp
ROOT=$(pwd)
for w in 10k-cigar cfg5; do
	OUT=$ROOT/gpurun_out/prof_r2m_$w; rm -rf $OUT; mkdir -p $OUT
	( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu --resident-only > $OUT/log 2>&1 )
	python3 - <<PY
import csv,glob
tot={}
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:60]; d=tot.setdefault(k,{}); d[r["Dispatch_Id"]]=d.get(r["Dispatch_Id"],0)+float(r["Counter_Value"])
for k,d in tot.items(): print("$w", k, "launches", len(d), "WRITE GB per launch", sum(d.values())/len(d)*1024/1e9)
PY
done
