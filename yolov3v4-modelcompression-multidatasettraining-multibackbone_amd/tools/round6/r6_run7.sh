#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6h; mkdir -p $O
rm -f $O/lane_ab.txt
for round in 1 2; do
  for lane in 0 2 3; do
    echo "== YOLO_HIP_WGRAD_LANE=$lane" >> $O/lane_ab.txt
    YOLO_HIP_WGRAD_LANE=$lane timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), r['by_role_ms'].get('wgrad'), r['by_role_ms'].get('dgrad'))" >> $O/lane_ab.txt 2>&1
  done
done
cat $O/lane_ab.txt
