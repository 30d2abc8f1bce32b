#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c8
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_adafactor_gpu.py tests/test_contrastive_gpu.py tests/test_dp_two_ranks_gpu.py -q -x 2>&1 | tail -30 > $O/pytest_new.txt; cat $O/pytest_new.txt
