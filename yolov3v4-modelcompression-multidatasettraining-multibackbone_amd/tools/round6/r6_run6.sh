#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t_kernels.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "backward_sums_in_the_data or carries_the_completed" 2>&1 | grep "^E  \|passed\|failed\|FAILED\|yolov3 608 b3" | cut -c1-400 | head -20 ) > $O/t_fused.txt 2>&1
( timeout 600 python bench.py 2>&1 | tail -1 ) > $O/bench.json 2>&1
cat $O/t_kernels.txt $O/t_fused.txt; cut -c1-1800 $O/bench.json
