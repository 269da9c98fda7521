#!/bin/bash
# round 6, GPU call 1: the new reference anchors (YOLOv4 training-step golden, ten-step SGD trajectory, mAP protocol on the device-calibrated
# YOLOv4-640 int8 state), run-to-run determinism, the cost of the deterministic reductions (A/B), the Infinity Cache probe.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "well_conditioned or sgd_trajectory" 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | grep "calm v\|608 b2\|sgd traj\|yolov3 320\|largest contrib\|passed\|failed\|^E  \|FAILED" | cut -c1-900 ) > $O/tests_train.txt 2>&1
( true || timeout 900 python -m pytest tests/test_ptq_calibration.py -m gpu -q -s -k "map_protocol" 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | tail -12 ) > $O/tests_ptq.txt 2>&1
for round in ; do
  for det in 1 0; do
    echo "== YH_DETERMINISTIC=$det" >> $O/det_ab.txt
    YH_DETERMINISTIC=$det timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
k=r.get('kernels') or {}
print(d['value'], d['ms_per_step'], {n: v['ms'] for n, v in list(k.items())[:0]}, r.get('by_class', {}).get('conv_wgrad'), {n: round(v['ms'],3) for n, v in k.items() if 'bn' in n or 'wgrad' in n or 'stem' in n})" >> $O/det_ab.txt 2>&1
  done
done
cat $O/tests_train.txt | tail -25; cat $O/tests_ptq.txt | tail -6; 
