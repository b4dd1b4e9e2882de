#!/bin/bash
# end-to-end headline with the code planes in LDS forced off / on / by rule
for rep in 1 2; do
for v in 0 1 rule; do
	if [ $v = rule ]; then unset KSW2AMD_LDSCODES; else export KSW2AMD_LDSCODES=$v; fi
	timeout 600 python bench.py --workload 10k --steps 12 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rep$rep ldscodes=$v 10k e2e', d['value'], 'resident', d['value_hbm_resident'])"
done
done 2>&1 | tee gpurun_out/r2u_ab.txt
