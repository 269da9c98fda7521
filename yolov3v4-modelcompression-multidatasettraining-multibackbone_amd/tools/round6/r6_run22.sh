#!/bin/bash
# picker rules of the ring-tile sweep (ups layers -> 64 x 128; <= 2 K steps -> 128 x 128): detect legs of YOLOv3-608 / YOLOv4-640 fp16 + network / kernel tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6w; mkdir -p $O
rm -f $O/d.txt
for i in 1 2; do
timeout 300 python bench.py --mode detect --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])" >> $O/d.txt 2>&1
timeout 300 python bench.py --mode detect --cfg $PKG/cfg/yolov4/yolov4.cfg --size 640 --batch 32 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])" >> $O/d.txt 2>&1
done
cat $O/d.txt
timeout 300 python $PKG/tools/profile_layers.py --batch 32 --size 640 --cfg $PKG/cfg/yolov4/yolov4.cfg 2>&1 | tail -3
timeout 300 python $PKG/tools/profile_layers.py --batch 64 --size 608 2>&1 | tail -3
( timeout 1500 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py tests/test_gpu_train.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t.txt 2>&1
cat $O/t.txt
