#!/bin/bash
# Round-end measurement set: full GPU test suite, default bench line, the N=8 rank shape and the RCCL
# call path on one GPU, rocprofv3 kernel stats of the bench command, and the two --pmc passes
# (FETCH_SIZE / WRITE_SIZE, counters only - never combined with tracing domains).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -rf $O; mkdir -p $O
if [ "$1" != "notests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
fi
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
timeout 300 python bench.py --residual-stream float32 --no-cpu-baseline > $O/bench_line_f32stream.json 2> $O/bench_f32.err; cat $O/bench_line_f32stream.json
timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n512.json 2> $O/bench_n512.err; cat $O/bench_n512.json
for gb in 2048 1024; do timeout 300 python bench.py --global-batch $gb --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_n$gb.json 2> $O/bench_n$gb.err; cat $O/bench_n$gb.json; done
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 BV_DP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_n512_rccl.json 2> $O/bench_n512_rccl.err; tail -2 $O/bench_n512_rccl.err; cat $O/bench_n512_rccl.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_profiled.json 2> $O/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W --microbatch=2048 --n_gpus=1 > $O/pmc_traffic.json 2> $O/pmc_summary.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls -la $O
