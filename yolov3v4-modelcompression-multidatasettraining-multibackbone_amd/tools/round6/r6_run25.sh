#!/bin/bash
# int8 picker rules of the ring-tile sweep: int8 detect legs + layer tables of YOLOv3-608 b64 / YOLOv4-640 b32, int8 / PTQ / network tests; then the training tile sweep
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6y; mkdir -p $O
rm -f $O/d.txt
for i in 1 2; do
timeout 300 python bench.py --mode detect --precision int8 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])" >> $O/d.txt 2>&1
timeout 300 python bench.py --mode detect --precision int8 --cfg $PKG/cfg/yolov4/yolov4.cfg --size 640 --batch 32 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'])" >> $O/d.txt 2>&1
done
cat $O/d.txt
timeout 300 python $PKG/tools/profile_layers.py --batch 32 --size 640 --precision int8 --cfg $PKG/cfg/yolov4/yolov4.cfg > $O/layers_v4_int8.txt 2>&1; tail -1 $O/layers_v4_int8.txt
timeout 300 python $PKG/tools/profile_layers.py --batch 64 --size 608 --precision int8 > $O/layers_v3_int8.txt 2>&1; tail -1 $O/layers_v3_int8.txt
( timeout 1500 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py tests/test_ptq_calibration.py tests/test_ptq.py tests/test_ptq_large.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t.txt 2>&1
cat $O/t.txt
bash $PKG/tools/round6/r6_run24.sh
