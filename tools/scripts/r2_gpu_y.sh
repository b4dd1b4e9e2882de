#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "splice or exts" 2>&1 | tail -3
for rep in 1 2; do
timeout 600 python bench.py --workload exts --steps 8 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('exts e2e', d['value'], 'resident', d['value_hbm_resident'], 'kernel_ms', d['roofline']['kernel_ms'], d['roofline']['fill_kernel_ms'])"
done
python tools/scripts/exts_classes.py 2>&1 | tail -12
