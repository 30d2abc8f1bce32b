#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py tests/test_siglip_step_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/pytest.txt
cat gpurun_out/pytest.txt
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; tail -3 gpurun_out/b.err; cat gpurun_out/b.json
