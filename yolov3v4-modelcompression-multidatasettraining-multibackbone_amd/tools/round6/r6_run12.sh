#!/bin/bash
# Walking depthwise forward / data-gradient kernels (YH_DW_WALK=1, default) against the round-1 kernels (=0): YOLOv3-Mobilenet training
# layer table both ways + the kernel, network and training tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6m; mkdir -p $O
CFG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg
for w in 1 0; do
  YH_DW_WALK=$w timeout 600 python yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools/profile_train.py --cfg $CFG --size 416 --batch 64 > $O/mobilenet_train_walk$w.txt 2>&1
  echo "== YH_DW_WALK=$w"; grep -E "^(dw|dwdgrad|dwwgrad|se|dse|total)" $O/mobilenet_train_walk$w.txt
done
( timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py tests/test_gpu_network.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -30 ) > $O/t.txt 2>&1
cat $O/t.txt
