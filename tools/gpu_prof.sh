#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_prof.json 2> gpurun_out/b_prof.err; cat gpurun_out/b_prof.json
