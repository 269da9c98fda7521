#!/bin/bash
# per-layer training tables of YOLOv4-608 b32 and YOLOv3-Mobilenetv3-416 b64 with the new picker rules (compare with gpurun_out/r6z/*_tile_0.txt)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6ab; mkdir -p $O
timeout 300 python $PKG/tools/profile_train.py --cfg $PKG/cfg/yolov4/yolov4.cfg --size 608 --batch 32 > $O/v4.txt 2>&1
timeout 300 python $PKG/tools/profile_train.py --cfg $PKG/cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg --size 416 --batch 64 > $O/mb.txt 2>&1
timeout 300 python $PKG/tools/profile_train.py --batch 64 --size 608 > $O/v3.txt 2>&1
tail -3 $O/v4.txt
