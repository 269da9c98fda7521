#!/bin/bash
# round 6, GPU call 3: whole GPU tier with the printed measurements, smoke, default bench line
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | grep "passed\|failed\|FAILED\|^E  \|int8 vs\|raw heads\|mAP\|608 b\|calm v\|sgd traj\|yolov3 320\|largest contrib\|drift\|pruned mobilenet\|calibration on\|cosine searches\|int8 engine vs" | cut -c1-1000 ) > $O/tests.txt 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | tail -4 ) > $O/smoke.txt 2>&1
( timeout 900 python bench.py 2>&1 | tail -1 ) > $O/bench.json 2>&1
tail -12 $O/tests.txt | cut -c1-300; cat $O/smoke.txt; cut -c1-1500 $O/bench.json
