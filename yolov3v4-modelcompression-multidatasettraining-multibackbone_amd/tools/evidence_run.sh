#!/bin/bash
# Evidence run on a GPU box (what produced profiles/r04_* .. r06_*): full GPU test tier, smoke, the default bench line, rocprofv3 stats of the same
# command, HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass, aggregated PER DISPATCH) and SQ / TCC counters of the training step,
# per-layer tables.
# Usage:  gpurun --timeout 3000 -- bash yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools/evidence_run.sh TAG [notests]
# Everything lands under gpurun_out/TAG_*; copy what is to be kept into profiles/ (hbm_traffic.json: gpurun_out/TAG_hbm_traffic.json).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
T=$PKG/tools
TAG=${1:-evidence}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
( timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | grep "passed\|failed\|FAILED\|Error\|int8 vs\|raw heads\|mAP\|608 b\|calm v\|sgd traj\|three runs\|largest contrib\|drift\|pruned mobilenet\|calibration on\|cosine searches\|int8 engine vs" | grep -v "print(" | cut -c1-1200 | tail -120 ) > gpurun_out/${TAG}_tests.log 2>&1
fi
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | tail -4 ) > gpurun_out/${TAG}_smoke.log 2>&1
( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/${TAG}_bench_default.json 2>&1
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_train_$c -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_det_$c -- python $R/bench.py --mode detect --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_i8_$c -- python $R/bench.py --mode detect --precision int8 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_sq -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc_tcc -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python $T/rocprof_summary.py stats $(find /tmp/prof_default -name "*.db" | head -1) > gpurun_out/${TAG}_rocprof_stats.txt 2>&1
python $T/rocprof_summary.py traffic gpurun_out/${TAG}_traffic_train.json $(find /tmp/prof_train_FETCH_SIZE /tmp/prof_train_WRITE_SIZE -name "*.db") > gpurun_out/${TAG}_traffic.log 2>&1
python $T/rocprof_summary.py traffic gpurun_out/${TAG}_traffic_detect.json $(find /tmp/prof_det_FETCH_SIZE /tmp/prof_det_WRITE_SIZE -name "*.db") >> gpurun_out/${TAG}_traffic.log 2>&1
python $T/rocprof_summary.py traffic gpurun_out/${TAG}_traffic_int8.json $(find /tmp/prof_i8_FETCH_SIZE /tmp/prof_i8_WRITE_SIZE -name "*.db") >> gpurun_out/${TAG}_traffic.log 2>&1
python - <<PY
import json
out = {'_note': 'HBM bytes per dispatch per kernel from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (tools/evidence_run.sh); '
               'per-dispatch means; fetch_kib_raw = counter as read, hbm_bytes_per_dispatch = 2 x fetch + write (see tools/rocprof_summary.py traffic and '
               'profiles/r04_traffic_calibration.txt: exact for streaming reads, an upper bound for kernels that fetch 64-byte row pieces)'}
for sec, f in (('train_batch64', 'train'), ('batch64', 'detect'), ('batch64_int8', 'int8')):
    try:
        out[sec] = json.load(open('gpurun_out/${TAG}_traffic_%s.json' % f))
    except Exception as e:
        out[sec] = {'_error': str(e)}
json.dump(out, open('gpurun_out/${TAG}_hbm_traffic.json', 'w'), indent=1, sort_keys=True)
PY
python $T/rocprof_summary.py pmc $(find /tmp/pmc_sq /tmp/pmc_tcc -name "*.db") > gpurun_out/${TAG}_pmc_train.txt 2>&1
timeout 300 python $T/profile_train.py --batch 64 --size 608 > gpurun_out/${TAG}_train_layers.txt 2>&1
timeout 300 python $T/profile_layers.py --batch 64 --size 608 > gpurun_out/${TAG}_layers_fp16.txt 2>&1
timeout 300 python $T/profile_layers.py --batch 64 --size 608 --precision int8 > gpurun_out/${TAG}_layers_int8.txt 2>&1
timeout 300 python $T/profile_layers.py --batch 32 --size 640 --precision int8 --cfg $PKG/cfg/yolov4/yolov4.cfg > gpurun_out/${TAG}_layers_v4_int8.txt 2>&1
timeout 300 python $T/profile_layers.py --batch 32 --size 640 --cfg $PKG/cfg/yolov4/yolov4.cfg > gpurun_out/${TAG}_layers_v4_fp16.txt 2>&1
timeout 600 python $T/pruned_finetune.py --bench > gpurun_out/${TAG}_pruned.txt 2>&1
# the other training families: rider legs of the bench line (YOLOv3-Mobilenetv3-416 b64 = BASELINE configs[4], YOLOv4-608 b32) + their layer tables
( timeout 900 python bench.py --families --no-v4 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/${TAG}_bench_families.json 2>&1
timeout 300 python $T/profile_train.py --cfg $PKG/cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg --size 416 --batch 64 > gpurun_out/${TAG}_train_layers_mobilenet.txt 2>&1
timeout 300 python $T/profile_train.py --cfg $PKG/cfg/yolov4/yolov4.cfg --size 608 --batch 32 > gpurun_out/${TAG}_train_layers_v4.txt 2>&1
python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_bench_families.json').read().strip().splitlines()[-1])
    for k in ('train_mobilenet_416', 'train_v4_608'):
        t = d.get(k) or {}
        print(k, t.get('value'), 'images/s', t.get('ms_per_step'), 'ms/step', (t.get('roofline') or {}).get('gpu_ms_per_step'), 'ms of kernels', d.get(k + '_error', ''))
except Exception as e:
    print('families bench:', e)
PY
tail -14 gpurun_out/${TAG}_tests.log 2>/dev/null; cat gpurun_out/${TAG}_smoke.log; cut -c1-300 gpurun_out/${TAG}_bench_default.json; head -14 gpurun_out/${TAG}_rocprof_stats.txt; cat gpurun_out/${TAG}_traffic.log; tail -5 gpurun_out/${TAG}_pruned.txt
