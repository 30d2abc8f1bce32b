#!/bin/bash
# round 2, call 2: rolling-epilogue GEMM probe (A/B + bit-exactness), GEMM tests, short bench
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2
rm -rf $O; mkdir -p $O
timeout 300 tools/probes/gemm_roll_probe.out > $O/roll_probe.txt 2>&1; cat $O/roll_probe.txt
timeout 600 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -15 > $O/pytest_gemm.txt; tail -5 $O/pytest_gemm.txt
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
