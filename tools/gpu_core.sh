#!/bin/bash
# The part of tools/gpu_final.sh that the bench line's roofline depends on: default bench line, rocprofv3
# kernel stats of the same command, the two --pmc passes (counters only) and their summary.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/core
rm -rf $O; mkdir -p $O
timeout 300 python bench.py > $O/bench_line.json 2> $O/bench.err; cat $O/bench_line.json | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/stats.err
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W --microbatch=2048 --n_gpus=1 > $O/pmc_traffic.json 2> $O/pmc_summary.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls $O
