#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest "tests/test_siglip_step_gpu.py::test_l16_336_siglip_step_small_batch" -x -q -m gpu 2>&1 | tail -30 > gpurun_out/pytest.txt
cat gpurun_out/pytest.txt
