#!/bin/bash
# LDS-plane max-pool backward (gather, no atomics) against the scatter form (YH_POOL_BWD_SCATTER=1): YOLOv4-608 b32 training step; int8 first layer
# with the four-value Mish decision: YOLOv4-640 int8 detect; max-pool / determinism / int8 tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6r; mkdir -p $O
rm -f $O/pool_ab.txt
for sc in 0 1 0 1; do
  echo "== YH_POOL_BWD_SCATTER=$sc" >> $O/pool_ab.txt
  YH_POOL_BWD_SCATTER=$sc timeout 600 python bench.py --mode train --cfg $PKG/cfg/yolov4/yolov4.cfg --size 608 --batch 32 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), 'dpool', r['by_role_ms'].get('dpool'), d['config']['loss'])" >> $O/pool_ab.txt 2>&1
done
cat $O/pool_ab.txt
timeout 600 python bench.py --mode detect --precision int8 --cfg $PKG/cfg/yolov4/yolov4.cfg --size 640 --batch 32 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])"
( timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py tests/test_gpu_network.py tests/test_ptq_calibration.py tests/test_ptq.py -m gpu -q -s 2>&1 | grep "^E  \|passed\|failed\|FAILED\|three runs" | cut -c1-300 | head -30 ) > $O/t.txt 2>&1
cat $O/t.txt
