#!/bin/bash
# The two training sweeps of r6_run19 / r6_run24 (weight-gradient split target, ring tile of the forward / data-gradient convs) on the other families:
# YOLOv4-608 b32 and YOLOv3-Mobilenetv3-416 b64
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6z; mkdir -p $O
for net in v4 mb; do
  if [ $net = v4 ]; then A="--cfg $PKG/cfg/yolov4/yolov4.cfg --size 608 --batch 32"; else A="--cfg $PKG/cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg --size 416 --batch 64"; fi
  unset YOLO_HIP_TILE YH_WGRAD_TARGET
  for t in 0 21 24 25 26 27; do
    if [ "$t" = "0" ]; then unset YOLO_HIP_TILE; else export YOLO_HIP_TILE=$t; fi
    timeout 300 python $PKG/tools/profile_train.py $A > $O/${net}_tile_$t.txt 2>&1
  done
  unset YOLO_HIP_TILE
  for t in -384 -512 -768 -1024; do
    export YH_WGRAD_TARGET=$t
    timeout 300 python $PKG/tools/profile_train.py $A > $O/${net}_target_$t.txt 2>&1
  done
  unset YH_WGRAD_TARGET
done
python - <<'PY'
import re, collections
def table(net, kind, keys, pat):
    rows = collections.OrderedDict()
    for t in keys:
        try:
            for l in open('gpurun_out/r6z/%s_%s_%s.txt' % (net, kind, t)):
                m = re.match(pat, l)
                if m:
                    rows.setdefault((m.group(1), m.group(2), m.group(3)), {})[t] = float(m.group(4))
        except OSError:
            pass
    return rows
for net in ('v4', 'mb'):
    tiles = "0 21 24 25 26 27".split()
    rows = table(net, 'tile', tiles, r'(fwd|bwd)\s+((?:conv|dgrad)\d+)\s+(\S+ \S+ k\d s\d)\s+([\d.]+)')
    print('==', net, 'forward / data-gradient tiles: layers where a forced ring tile beats the picker (column 0) by > 6 %')
    print('%-4s %-10s %-28s ' % ('', 'layer', 'shape') + ' '.join('%8s' % t for t in tiles))
    gain = 0.0
    for (d, n, s), v in rows.items():
        if '0' not in v: continue
        best = min(v.values())
        if best < 0.94 * v['0']:
            gain += v['0'] - best
            print('%-4s %-10s %-28s ' % (d, n, s) + ' '.join('%8.4f' % v.get(t, float('nan')) for t in tiles))
    print('sum of the gains %.3f ms' % gain)
    targets = "0 -384 -512 -768 -1024".split()
    rows = table(net, 'target', targets[1:], r'(bwd)\s+(wgrad\d+)\s+(\S+ \S+ k\d s\d)\s+([\d.]+)')
    for k, v in table(net, 'tile', ['0'], r'(bwd)\s+(wgrad\d+)\s+(\S+ \S+ k\d s\d)\s+([\d.]+)').items():
        rows.setdefault(k, {}).update(v)
    print('==', net, 'weight-gradient split target: layers where a forced target beats the shipped rule (column 0) by > 6 %')
    print('%-4s %-10s %-28s ' % ('', 'layer', 'shape') + ' '.join('%8s' % t for t in targets))
    gain = 0.0
    for (d, n, s), v in rows.items():
        if '0' not in v: continue
        best = min(v.values())
        if best < 0.94 * v['0']:
            gain += v['0'] - best
            print('%-4s %-10s %-28s ' % (d, n, s) + ' '.join('%8.4f' % v.get(t, float('nan')) for t in targets))
    print('sum of the gains %.3f ms' % gain)
    for t in targets[1:]:
        try:
            print('   target', t, [l.strip() for l in open('gpurun_out/r6z/%s_target_%s.txt' % (net, t)) if l.startswith('wgrad ')])
        except OSError:
            pass
PY
