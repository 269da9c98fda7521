#!/bin/bash
# Pixel-split count of the im2col weight-gradient kernel (YH_WGRAD_TARGET: workgroups per launch; negative = rounded down) on the 1x1 / stride-2
# layers of YOLOv3-608 b64: per-layer weight-gradient rows of tools/profile_train.py for four targets
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6t; mkdir -p $O
for t in ${TARGETS:-default -256 -512 -1024 -1536}; do
  if [ "$t" = "default" ]; then unset YH_WGRAD_TARGET; else export YH_WGRAD_TARGET=$t; fi
  timeout 300 python $PKG/tools/profile_train.py --batch 64 --size 608 > $O/train_$t.txt 2>&1
  echo "== target $t: $(grep -E '^wgrad ' $O/train_$t.txt)"
done
python - <<'PY'
import re, collections
rows = collections.OrderedDict()
import os
ts = os.environ.get("TARGETS", "default -256 -512 -1024 -1536").split()
for t in ts:
    for l in open('gpurun_out/r6t/train_%s.txt' % t):
        m = re.match(r'bwd\s+(wgrad\d+)\s+(\S+ \S+ k\d s\d)\s+([\d.]+)', l)
        if m:
            rows.setdefault((m.group(1), m.group(2)), {})[t] = float(m.group(3))
print('%-10s %-28s ' % ('layer', 'shape') + ' '.join('%8s' % t for t in ts))
for (n, s), v in rows.items():
    if 'k1' in s or 's2' in s:
        print('%-10s %-28s ' % (n, s) + ' '.join('%8.4f' % v.get(t, float('nan')) for t in ts))
PY
