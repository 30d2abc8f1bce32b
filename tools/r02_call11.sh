#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c11
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_naflex_gpu.py tests/test_checkpoint_gpu.py tests/test_evaluators_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -40 > $O/pytest.txt; cat $O/pytest.txt
