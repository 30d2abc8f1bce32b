#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k
rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_contrastive_gpu.py -q -m gpu -k "sgemm or contrastive or siglip_loss" -x 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
timeout 200 python tools/sgemm_bench.py > $O/sgemm_bench.txt 2>&1; cat $O/sgemm_bench.txt
