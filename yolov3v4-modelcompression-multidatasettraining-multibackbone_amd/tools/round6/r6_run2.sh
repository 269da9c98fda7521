#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
O=gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/bn_probe5.txt
for sk in 0 256 4096 65536 1056768; do
  echo "== YH_PROBE_SKEW=$sk" >> $O/bn_probe5.txt
  ( YH_PROBE_SKEW=$sk timeout 300 python $T/probe/run_bn_probe.py 2>&1 | tail -9 | cut -c1-130 ) >> $O/bn_probe5.txt 2>&1
done
cat $O/bn_probe5.txt
