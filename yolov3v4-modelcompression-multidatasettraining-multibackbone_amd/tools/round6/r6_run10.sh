#!/bin/bash
# weight-gradient reduce launches on the plan's reduce stream (YOLO_HIP_ASYNC_REDUCE=1, the default) against one stream (=0)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6k; mkdir -p $O
rm -f $O/async_ab.txt
for round in 1 2 3; do
  for ar in 1 0; do
    echo "== YOLO_HIP_ASYNC_REDUCE=$ar" >> $O/async_ab.txt
    YOLO_HIP_ASYNC_REDUCE=$ar timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), r['by_role_ms'].get('wgrad'), d['config']['loss'])" >> $O/async_ab.txt 2>&1
  done
done
( timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t_train.txt 2>&1
cat $O/async_ab.txt $O/t_train.txt
