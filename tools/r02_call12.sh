#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c12
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_nccl_gpu.py tests/test_dp_two_ranks_gpu.py -q -x 2>&1 | tail -30 > $O/pytest.txt; cat $O/pytest.txt
