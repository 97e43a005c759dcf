#!/bin/bash
# per-kernel times of the bracket path (uniform01 and sorted, bracket variant only)
mkdir -p gpurun_out/r02v
cd /tmp && export TMPDIR=/tmp
for c in uniform01 sorted; do
SELECT_AB_CASES=$c SELECT_AB_VARIANTS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o sel --output-format csv -- python $GRAFT_REPO_ROOT/tools/select_ab.py > $GRAFT_REPO_ROOT/gpurun_out/r02v/run_$c.log 2>&1
f=$(find /tmp/prof_$c -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r02v/kernel_stats_$c.csv
done
cd $GRAFT_REPO_ROOT
for c in uniform01 sorted; do echo == $c; cut -c1-90,150-400 gpurun_out/r02v/kernel_stats_$c.csv | head -14; done
