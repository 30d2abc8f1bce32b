#!/bin/bash
# GPU call 3 (round 3): BERT tower tests, LayerNorm re-emission, micro-batch / light-context step tests, bench N=1.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_bert_gpu.py tests/test_kernels_gpu.py -x -q -k "bert or layernorm" 2>&1 | tail -12 > $O/pytest_bert.txt; cat $O/pytest_bert.txt
timeout 900 python -m pytest tests/test_siglip_step_gpu.py -x -q -k "microbatched or n32_through or bench_mode or tiny_two" 2>&1 | tail -8 > $O/pytest_step.txt; cat $O/pytest_step.txt
timeout 400 python bench.py --no-cpu-baseline --no-bf16-stream > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json | cut -c1-900
ls $O
