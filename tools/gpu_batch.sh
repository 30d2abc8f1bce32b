#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest "tests/test_siglip_step_gpu.py::test_l16_336_siglip_step_small_batch" tests/test_train_step_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/pytest_tail.txt
cat gpurun_out/pytest_tail.txt
