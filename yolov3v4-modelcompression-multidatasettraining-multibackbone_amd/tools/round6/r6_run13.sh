#!/bin/bash
# rocprofv3 kernel stats of the YOLOv3-Mobilenetv3 training step (416, batch 64): which of the small kernels carry the 22 ms
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6n; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_mb
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_mb -- python $R/bench.py --mode train --cfg $PKG/cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg --size 416 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/bench_prof.log 2>&1
cd $R
python $PKG/tools/rocprof_summary.py stats $(find /tmp/prof_mb -name "*.db" | head -1) > $O/rocprof_stats_mobilenet_train.txt 2>&1
head -50 $O/rocprof_stats_mobilenet_train.txt
tail -1 $O/bench_prof.log | cut -c1-400
