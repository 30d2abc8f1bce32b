#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_dp_two_ranks_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/pytest.txt
cat gpurun_out/pytest.txt
