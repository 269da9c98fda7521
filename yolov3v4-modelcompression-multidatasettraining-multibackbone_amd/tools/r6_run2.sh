#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
O=gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/bn_probe4.txt $O/bn_step_ab4.txt
for rt in 1024 2048 4096; do
  echo "== YH_BN_RTARGET=$rt" >> $O/bn_probe4.txt
  ( YH_BN_RTARGET=$rt timeout 300 python $T/probe/run_bn_probe.py 2>&1 | tail -9 | cut -c1-150 ) >> $O/bn_probe4.txt 2>&1
done
for round in 1 2; do
  for rt in 1024 2048 4096; do
    echo "== YH_BN_RTARGET=$rt" >> $O/bn_step_ab4.txt
    YH_BN_RTARGET=$rt timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), {k: v for k, v in r['by_role_ms'].items() if 'bn' in k})" >> $O/bn_step_ab4.txt 2>&1
  done
done
cat $O/bn_probe4.txt | cut -c1-100; cat $O/bn_step_ab4.txt
