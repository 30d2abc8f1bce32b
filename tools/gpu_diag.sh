#!/bin/bash
# Diagnostic GPU call: event overhead, kernel trace (gaps), FETCH_SIZE pass.
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd $GRAFT_REPO_ROOT
date
timeout 300 python bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline > $O/b_noroof.json 2> $O/b_noroof.err
date
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b_trace.json 2> $O/b_trace.err
date
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O/b_pmc.json 2> $O/b_pmc.err
date
ls -la $O $O/trace/* $O/pmc_fetch/* | head -40
cat $O/b_noroof.json $O/b_trace.json $O/b_pmc.json
