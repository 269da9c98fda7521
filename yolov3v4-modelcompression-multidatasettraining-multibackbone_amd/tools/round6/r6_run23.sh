#!/bin/bash
# Ring-tile sweep of the forward convolutions (tools/profile_layers.py --tile = YOLO_HIP_TILE): does any layer of YOLOv3-608 b64 / YOLOv4-640 b32 fp16 have a
# faster generic ring tile than the picker's choice?  (The 3x3 / s1 layers on the halo ping-pong kernel keep it: tiles < 40 replace every layer.)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6v; mkdir -p $O
TILES="0 21 24 25 26 27"
for net in v3 v4; do
  if [ $net = v3 ]; then A="--batch 64 --size 608"; else A="--batch 32 --size 640 --cfg $PKG/cfg/yolov4/yolov4.cfg"; fi
  for t in $TILES; do
    timeout 200 python $PKG/tools/profile_layers.py $A --precision int8 --tile $t > $O/${net}_int8_$t.txt 2>&1
  done
done
python - <<'PY'
import re, collections
tiles = "0 21 24 25 26 27".split()
for net in ('v3', 'v4'):
    rows = collections.OrderedDict()
    for t in tiles:
        try:
            for l in open('gpurun_out/r6v/%s_int8_%s.txt' % (net, t)):
                m = re.match(r'(conv\d+)\s+(\S+)\s+(\S+ \S+ k\d s\d(?: \+res| ups)?)\s+([\d.]+)', l)
                if m:
                    rows.setdefault((m.group(1), m.group(3)), {})[t] = (float(m.group(4)), m.group(2))
        except OSError:
            pass
    print('==', net, 'int8 (layers where a forced ring tile beats the default by > 4 %)')
    gain = 0.0
    for (n, s), v in rows.items():
        if '0' not in v: continue
        d = v['0'][0]
        best = min((x[0], t, x[1]) for t, x in v.items())
        if best[0] < 0.96 * d:
            gain += d - best[0]
            print('%-8s %-34s default %-18s %.4f   tile %s %-18s %.4f' % (n, s, v['0'][1], d, best[1], best[2], best[0]))
    print('   sum of the gains %.3f ms' % gain)
PY
