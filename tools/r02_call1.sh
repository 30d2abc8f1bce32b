#!/bin/bash
# round 2, call 1: state check (GPU tests incl. the new parity cases), dbuf probe, bench + kernel stats
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c1
rm -rf $O; mkdir -p $O
timeout 120 tools/probes/gemm_dbuf_probe.out > $O/dbuf_probe.txt 2>&1; tail -20 $O/dbuf_probe.txt
timeout 900 python -m pytest tests -q -m gpu -rA 2>&1 | tail -150 > $O/pytest.txt; tail -5 $O/pytest.txt
timeout 300 python bench.py --steps 6 --warmup 2 > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*kernel_trace.csv" -delete
ls -la $O
