#!/bin/bash
# Ring-tile sweep of the TRAINING step's forward and data-gradient convolutions (YOLO_HIP_TILE forces one tile on every conv of engine/train.py): YOLOv3-608 b64 fp16
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6x; mkdir -p $O
TILES="0 21 24 25 26 27 31 35"
for t in $TILES; do
  if [ "$t" = "0" ]; then unset YOLO_HIP_TILE; else export YOLO_HIP_TILE=$t; fi
  timeout 300 python $PKG/tools/profile_train.py --batch 64 --size 608 > $O/train_$t.txt 2>&1
done
python - <<'PY'
import re, collections
tiles = "0 21 24 25 26 27 31 35".split()
rows = collections.OrderedDict()
for t in tiles:
    try:
        for l in open('gpurun_out/r6x/train_%s.txt' % t):
            m = re.match(r'(fwd|bwd)\s+((?:conv|dgrad)\d+)\s+(\S+ \S+ k\d s\d)\s+([\d.]+)', l)
            if m:
                rows.setdefault((m.group(1), m.group(2), m.group(3)), {})[t] = float(m.group(4))
    except OSError:
        pass
print('layers where a forced ring tile beats the default by > 4 %; all tiles shown')
print('%-4s %-10s %-28s ' % ('', 'layer', 'shape') + ' '.join('%8s' % t for t in tiles))
gain = 0.0
for (d, n, s), v in rows.items():
    if '0' not in v: continue
    best = min(v.values())
    if best < 0.96 * v['0']:
        gain += v['0'] - best
        print('%-4s %-10s %-28s ' % (d, n, s) + ' '.join('%8.4f' % v.get(t, float('nan')) for t in tiles))
print('sum of the gains %.3f ms' % gain)
PY
