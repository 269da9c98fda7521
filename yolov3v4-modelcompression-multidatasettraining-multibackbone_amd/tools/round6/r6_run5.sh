#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | head -20 ) > $O/tests_kernels.txt 2>&1
( timeout 900 python -m pytest tests/test_ptq_large.py tests/test_ptq_calibration.py tests/test_ptq.py tests/test_gpu_network.py tests/test_gpu_train.py -m gpu -q 2>&1 | grep "passed\|failed\|FAILED\|^E  " | cut -c1-300 ) > $O/tests_rest.txt 2>&1
rm -f $O/fin_ab.txt
for round in 1 2; do
  for rows in 4096 0; do
    echo "== YH_BN_FIN_ROWS=$rows" >> $O/fin_ab.txt
    YH_BN_FIN_ROWS=$rows timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), {k: v for k, v in r['by_role_ms'].items() if 'bn' in k})" >> $O/fin_ab.txt 2>&1
  done
done
cat $O/tests_kernels.txt $O/tests_rest.txt $O/fin_ab.txt
