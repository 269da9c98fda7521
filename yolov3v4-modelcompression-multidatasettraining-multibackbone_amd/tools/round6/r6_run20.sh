#!/bin/bash
# 1x1 weight gradients at two workgroups per CU (YH_WGRAD_TARGET default -512 for 1x1): training bench, A/B against -768 on the same box, training tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6u; mkdir -p $O
rm -f $O/ab.txt
for round in 1 2; do
  for t in default -768; do
    if [ "$t" = "default" ]; then unset YH_WGRAD_TARGET; else export YH_WGRAD_TARGET=$t; fi
    echo "== YH_WGRAD_TARGET=$t" >> $O/ab.txt
    timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), 'wgrad', r['by_role_ms'].get('wgrad'), d['config']['loss'])" >> $O/ab.txt 2>&1
  done
done
unset YH_WGRAD_TARGET
cat $O/ab.txt
( timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t.txt 2>&1
cat $O/t.txt
