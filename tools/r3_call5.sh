#!/bin/bash
# GPU call 5 (round 3): the full GPU suite (parity report -> profiles/r03_parity_report.jsonl, together with the
# full-depth L/16 row) and the other BASELINE configs with their roofline objects.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/pytest.txt; cat $O/pytest.txt
BV_RUN_SLOW=1 timeout 900 python -m pytest tests/test_siglip_step_gpu.py -q -k full_depth 2>&1 | tail -3 > $O/pytest_slow.txt; cat $O/pytest_slow.txt
cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl
timeout 900 python tools/bench_configs.py c2 c4 c5 c5b --steps 5 > $O/bench_configs.jsonl 2> $O/bench_configs.err; tail -3 $O/bench_configs.err; cut -c1-400 $O/bench_configs.jsonl
