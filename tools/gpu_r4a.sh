#!/bin/bash
# round-4 development run A: new parity tests + the one-launch attention backward (tests, A/B timing, short bench)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention and not other_head and not map_" -x 2>&1 | tail -25 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
timeout 300 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt
timeout 600 python -m pytest tests/test_gemm256_gpu.py -q -m gpu -k "multi_tile" 2>&1 | tail -15 > $O/pytest_gemm.txt; cat $O/pytest_gemm.txt
timeout 600 python -m pytest tests/test_siglip_step_gpu.py -q -m gpu -k "n64 or tiny_two or small_batch" 2>&1 | tail -15 > $O/pytest_e2e.txt; cat $O/pytest_e2e.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-stream > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cat $O/bench.json
