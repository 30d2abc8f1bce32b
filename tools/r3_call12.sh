#!/bin/bash
# GPU call 12 (round 3): full GPU suite (timed), default bench line (with bf16 object + CPU baseline), other configs.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c12; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -20 > $O/pytest.txt; cat $O/pytest.txt
BV_RUN_SLOW=1 timeout 900 python -m pytest tests/test_siglip_step_gpu.py -q -k full_depth 2>&1 | tail -3 > $O/pytest_slow.txt; cat $O/pytest_slow.txt
cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl
cp $O/parity_report.jsonl profiles/r03_parity_report.jsonl
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
timeout 900 python tools/bench_configs.py c2 c4 c5 c5b --steps 5 > $O/bench_configs.jsonl 2> $O/bench_configs.err; tail -3 $O/bench_configs.err; cut -c1-200 $O/bench_configs.jsonl
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
