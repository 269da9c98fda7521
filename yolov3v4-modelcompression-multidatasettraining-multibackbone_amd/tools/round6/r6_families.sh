#!/bin/bash
# other model families on the same paths (training step; detect legs), as profiles/r02_model_families.txt
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6m; mkdir -p $O
rm -f $O/families.txt
run() {  # cfg size batch
  echo "== $1 $2 b$3 train" >> $O/families.txt
  timeout 600 python bench.py --mode train --cfg $PKG/cfg/$1 --size $2 --batch $3 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['metric'], d['value'], d['ms_per_step'], r.get('top_classes'), r.get('batchnorm_passes_ms'))" >> $O/families.txt 2>&1
}
run yolov4/yolov4.cfg 608 32
run yolov3tiny/yolov3-tiny.cfg 416 64
run yolov4tiny/yolov4-tiny.cfg 416 64
run yolov3-mobilenet/yolov3-mobilenet-coco.cfg 416 64
cat $O/families.txt
