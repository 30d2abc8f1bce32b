#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c5
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" -s 2>&1 | tail -25 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
timeout 300 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt
