#!/bin/bash
# rocprofv3 kernel stats of the training step (bench.py --mode train), summary into gpurun_out/$1
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
T=$R/yolov3v4-modelcompression-multidatasettraining-multibackbone_amd/tools
O=gpurun_out/${1:-r6p}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_train
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/bench_prof.log 2>&1
cd $R
python $T/rocprof_summary.py stats $(find /tmp/prof_train -name "*.db" | head -1) > $O/rocprof_stats_train.txt 2>&1
head -60 $O/rocprof_stats_train.txt
