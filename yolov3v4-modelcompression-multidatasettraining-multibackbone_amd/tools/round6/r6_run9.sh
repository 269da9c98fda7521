#!/bin/bash
# BatchNorm passes walking their tensor back to front (YH_BN_REVERSE bit 0 forward, 1 reduce, 2 apply): A/B in the training step
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r6j; mkdir -p $O
rm -f $O/rev_ab.txt
for round in 1 2; do
  for rev in 0 1 4 6 7; do
    echo "== YH_BN_REVERSE=$rev" >> $O/rev_ab.txt
    YH_BN_REVERSE=$rev timeout 300 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), {k: v for k, v in r['by_role_ms'].items() if 'bn' in k})" >> $O/rev_ab.txt 2>&1
  done
done
cat $O/rev_ab.txt
