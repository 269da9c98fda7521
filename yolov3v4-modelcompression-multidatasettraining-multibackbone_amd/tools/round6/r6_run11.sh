#!/bin/bash
# hpp statistics rows staged in LDS (YH_HPP_STATS_STAGE=1, default) against direct four-lane stores (=0): training step A/B + tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6l; mkdir -p $O
rm -f $O/stage_ab.txt
for round in 1 2; do
  for st in 1 0; do
    echo "== YH_HPP_STATS_STAGE=$st" >> $O/stage_ab.txt
    YH_HPP_STATS_STAGE=$st timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), r['by_role_ms'].get('conv'), d['config']['loss'])" >> $O/stage_ab.txt 2>&1
  done
done
( timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t.txt 2>&1
cat $O/stage_ab.txt $O/t.txt
