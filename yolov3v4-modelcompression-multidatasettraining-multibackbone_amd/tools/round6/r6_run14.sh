#!/bin/bash
# Squeeze-excite kernels of round 6 (row products on sixteen lanes, image/channel sums on 1024 threads, walking apply pass) and the
# two-level depthwise partial-row sum: YOLOv3-Mobilenetv3 training layer table, detect speed, tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6o; mkdir -p $O
CFG=$PKG/cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg
timeout 600 python $PKG/tools/profile_train.py --cfg $CFG --size 416 --batch 64 > $O/mobilenet_train.txt 2>&1
sed -n '/^wgrad\|^dgrad\|^conv /,$p' $O/mobilenet_train.txt | head -30
grep -E "fwd  se|bwd  dse|dwwgrad" $O/mobilenet_train.txt
timeout 600 python bench.py --mode detect --cfg $CFG --size 416 --batch 64 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
( timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py tests/test_gpu_network.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -30 ) > $O/t.txt 2>&1
cat $O/t.txt
