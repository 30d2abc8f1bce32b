#!/bin/bash
# GPU call 1 (round 3): GEMM epilogue probe + 32x32x16 main-loop probe, vendor yardstick, the GEMM tests.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; rm -rf $O; mkdir -p $O
timeout 300 tools/probes/gemm_r3_probe.out > $O/gemm_r3_probe.txt 2>&1; tail -20 $O/gemm_r3_probe.txt
timeout 600 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -x -q -k "gemm or gelu or epilogue or nt_ or tn_ or roll or reserved" 2>&1 | tail -6 > $O/pytest_gemm.txt; cat $O/pytest_gemm.txt
timeout 420 python tools/gemm_yardstick.py > $O/gemm_yardstick.txt 2> $O/gemm_yardstick.err; cat $O/gemm_yardstick.txt | cut -c1-200 | head -40; tail -3 $O/gemm_yardstick.err
timeout 300 python -m pytest tests/test_siglip_step_gpu.py -x -q -k "tiny_two_towers or b16_siglip_step_small or microbatched" 2>&1 | tail -6 > $O/pytest_step.txt; cat $O/pytest_step.txt
