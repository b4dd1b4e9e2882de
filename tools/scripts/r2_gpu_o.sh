#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
for w in cfg4-so; do
	OUT=$ROOT/gpurun_out/prof_r2o_$w; rm -rf $OUT; mkdir -p $OUT
	( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu --resident-only > $OUT/log 2>&1 )
	tail -2 $OUT/log
	python3 - <<PY
import csv,glob
tot={}
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:60]; d=tot.setdefault(k,{}); d[r["Dispatch_Id"]]=d.get(r["Dispatch_Id"],0)+float(r["Counter_Value"])
for k,d in tot.items(): print("$w", k, "launches", len(d), "WRITE GB per launch", sum(d.values())/len(d)*1024/1e9)
PY
done
