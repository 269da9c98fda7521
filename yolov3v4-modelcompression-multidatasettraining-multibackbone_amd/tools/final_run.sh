#!/bin/bash
# reduced end-of-round evidence: full GPU tier, smoke, default bench line, rocprofv3 stats of the same command, layer tables
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
mkdir -p gpurun_out/r5f
export TMPDIR=/tmp
( timeout 1100 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | tail -6 ) > gpurun_out/r5f/tests.txt 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Model Summary\|amdgpu.ids" | tail -4 ) > gpurun_out/r5f/smoke.txt 2>&1
( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r5f/bench.json 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_default -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r5f/bench_prof.log 2>&1
cd $R
python $T/rocprof_summary.py stats $(find /tmp/prof_default -name "*.db" | head -1) > gpurun_out/r5f/rocprof_stats.txt 2>&1
timeout 300 python $T/profile_train.py --batch 64 --size 608 > gpurun_out/r5f/train_layers.txt 2>&1
timeout 300 python $T/profile_layers.py --batch 64 --size 608 > gpurun_out/r5f/layers_fp16.txt 2>&1
timeout 300 python $T/profile_layers.py --batch 64 --size 608 --precision int8 > gpurun_out/r5f/layers_int8.txt 2>&1
tail -3 gpurun_out/r5f/tests.txt; cat gpurun_out/r5f/smoke.txt; cut -c1-200 gpurun_out/r5f/bench.json; head -8 gpurun_out/r5f/rocprof_stats.txt; tail -3 gpurun_out/r5f/train_layers.txt; tail -2 gpurun_out/r5f/layers_fp16.txt; tail -2 gpurun_out/r5f/layers_int8.txt
