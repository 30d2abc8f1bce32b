#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_step_gpu.py "tests/test_kernels_gpu.py::test_softmax_xent" "tests/test_kernels_gpu.py::test_sigmoid_xent" "tests/test_kernels_gpu.py::test_tanh_and_mixup" "tests/test_kernels_gpu.py::test_sqnorm_and_adam_vs_oracle" -x -q -m gpu 2>&1 | tail -30 > gpurun_out/pytest.txt
cat gpurun_out/pytest.txt
