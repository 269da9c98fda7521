#!/bin/bash
# int8 Mish epilogue with the tie test as a v_fma / v_min3 chain: YOLOv4-640 int8 / fp16 detect legs, int8 layer table, int8 + PTQ tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6q; mkdir -p $O
for prec in int8 fp16 int8; do
  timeout 600 python bench.py --mode detect --precision $prec --cfg $PKG/cfg/yolov4/yolov4.cfg --size 640 --batch 32 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])" >> $O/v4_detect.txt 2>&1
done
cat $O/v4_detect.txt
timeout 300 python $PKG/tools/profile_layers.py --batch 32 --size 640 --precision int8 --cfg $PKG/cfg/yolov4/yolov4.cfg > $O/layers_v4_int8.txt 2>&1
tail -16 $O/layers_v4_int8.txt
( timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py tests/test_ptq_calibration.py tests/test_ptq_large.py tests/test_ptq.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -30 ) > $O/t.txt 2>&1
cat $O/t.txt
