#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python tools/gemm_group_pmc.py run > $O/run.out 2> $O/run.err; tail -3 $O/run.err
F=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python tools/gemm_group_pmc.py parse $F > $O/group_pmc.txt 2>&1; cat $O/group_pmc.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
