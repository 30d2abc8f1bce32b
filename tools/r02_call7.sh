#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c7
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm256_gpu.py -q -x 2>&1 | tail -4 > $O/pytest_k.txt; cat $O/pytest_k.txt
timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n512.json 2> $O/bench_n512.err; tail -2 $O/bench_n512.err; cat $O/bench_n512.json
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
