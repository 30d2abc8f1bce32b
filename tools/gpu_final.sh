#!/bin/bash
# Round-end measurement set: full GPU test suite, default bench line, rocprofv3 kernel stats of
# the same command, and the two --pmc passes (FETCH_SIZE / WRITE_SIZE, counters only).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
ls $O/*/*
