#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
PKG=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6i; mkdir -p $O
rm -f $O/pwl_px.txt
for px in 262144 190000 90000; do
  for prec in int8 fp16; do
    echo "== YH_PWL_MIN_PIXELS=$px yolov4-640 b32 $prec" >> $O/pwl_px.txt
    YH_PWL_MIN_PIXELS=$px timeout 300 python $T/profile_layers.py --batch 32 --size 640 --precision $prec --cfg $PKG/cfg/yolov4/yolov4.cfg 2>&1 | tail -16 | grep "igemm\|total" >> $O/pwl_px.txt
  done
  echo "== YH_PWL_MIN_PIXELS=$px yolov3-608 b64 fp16 / int8" >> $O/pwl_px.txt
  YH_PWL_MIN_PIXELS=$px timeout 300 python $T/profile_layers.py --batch 64 --size 608 2>&1 | tail -3 | grep total >> $O/pwl_px.txt
  YH_PWL_MIN_PIXELS=$px timeout 300 python $T/profile_layers.py --batch 64 --size 608 --precision int8 2>&1 | tail -3 | grep total >> $O/pwl_px.txt
done
cat $O/pwl_px.txt
