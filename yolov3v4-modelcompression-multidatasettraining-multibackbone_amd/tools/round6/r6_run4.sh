#!/bin/bash
# round 6, GPU call 4: the cheaper int8 Mish epilogue - exhaustive self-test, int8 tests, detect legs
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -8 ) > $O/tests_kernels.txt 2>&1
( timeout 900 python -m pytest tests/test_ptq_large.py tests/test_ptq_calibration.py tests/test_ptq.py tests/test_gpu_network.py -m gpu -q -s 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | grep "passed\|failed\|FAILED\|^E  \|int8 vs\|raw heads\|mAP\|drift\|int8 engine vs" | cut -c1-400 ) > $O/tests_int8.txt 2>&1
( timeout 600 python bench.py --mode detect --precision int8 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_detect.json 2>&1
python - <<PY > $O/bench_detect_summary.txt
import json
d = json.loads(open('$O/bench_detect.json').read())
def walk(o, path=''):
    if isinstance(o, dict):
        if 'value' in o and 'metric' in o: print(path, o['metric'], o['value'], o.get('ms_per_step'))
        for k, v in o.items(): walk(v, path + '/' + k)
walk(d)
PY
timeout 300 python $T/profile_layers.py --batch 32 --size 640 --precision int8 --cfg yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/cfg/yolov4/yolov4.cfg > $O/layers_v4_int8.txt 2>&1
tail -4 $O/tests_kernels.txt; cat $O/tests_int8.txt | tail -15; cat $O/bench_detect_summary.txt; tail -16 $O/layers_v4_int8.txt
