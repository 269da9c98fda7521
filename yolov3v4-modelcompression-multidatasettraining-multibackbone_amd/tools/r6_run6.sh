#!/bin/bash
# round 6, GPU call 6: backward sums carried by the completing data gradient (kernel test, train tests, A/B in the step)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
O=gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "carries_the_completed" 2>&1 | grep "^E  \|passed\|failed\|FAILED" | head -20 ) > $O/t_kernel.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | head -20 ) > $O/t_kernels.txt 2>&1
rm -f $O/dbn_ab.txt
for round in 1 2; do
  for fuse in 1 0; do
    echo "== YOLO_HIP_FUSE_DBN=$fuse" >> $O/dbn_ab.txt
    YOLO_HIP_FUSE_DBN=$fuse timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), {k: v for k, v in r['by_role_ms'].items() if 'bn' in k or k == 'dgrad'}, d['config']['loss'])" >> $O/dbn_ab.txt 2>&1
  done
done
( timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t_train.txt 2>&1
cat $O/t_kernel.txt $O/t_kernels.txt $O/dbn_ab.txt $O/t_train.txt
