#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c10
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_checkpoint_gpu.py tests/test_evaluators_gpu.py -q -x 2>&1 | tail -12 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python tools/bench_configs.py c2 c5 c4 > $O/bench_configs.jsonl 2> $O/bench_configs.err; tail -3 $O/bench_configs.err; cat $O/bench_configs.jsonl
