#!/bin/bash
# YOLOv4-608 batch 32 fp16 training: per-layer table of the final library
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6s; mkdir -p $O
timeout 600 python $PKG/tools/profile_train.py --cfg $PKG/cfg/yolov4/yolov4.cfg --size 608 --batch 32 > $O/v4_train_layers.txt 2>&1
tail -25 $O/v4_train_layers.txt
