#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention and not other_head and not map_" -x 2>&1 | tail -5 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
timeout 300 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt | cut -c1-400
