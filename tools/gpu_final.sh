#!/bin/bash
# Round-end measurement set (round 3): [full GPU test suite,] default bench line (fp32 residual stream = the reference's
# arithmetic, with the bf16-stream object and the CPU baseline), the rank shapes of N = 2 / 4 / 8 on one GPU, the N = 8
# shape with the one-rank RCCL collectives in the loop, rocprofv3 kernel stats of the bench command and of the n = 512
# shape, and the two --pmc passes (FETCH_SIZE / WRITE_SIZE, counters only - never combined with tracing domains).
#   bash tools/gpu_final.sh [notests] [slow] [yard]     (slow: the full-depth L/16@336 case; yard: the vendor yardsticks - GEMM, fused epilogues, attention, LayerNorm / Adam, whole step)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.jsonl
if [[ " $* " != *" notests "* ]]; then
  timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
fi
if [[ " $* " == *" slow "* ]]; then
  BV_RUN_SLOW=1 timeout 900 python -m pytest tests/test_siglip_step_gpu.py -q -k full_depth 2>&1 | tail -4 > $O/pytest_slow.txt; cat $O/pytest_slow.txt
fi
cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream > $O/bench_n512.json 2> $O/bench_n512.err; cat $O/bench_n512.json
timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream --no-roofline --residual-stream bfloat16 > $O/bench_n512_bf16stream.json 2> $O/bench_n512_bf16.err
for gb in 2048 1024; do timeout 300 python bench.py --global-batch $gb --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-bf16-stream > $O/bench_n$gb.json 2> $O/bench_n$gb.err; cat $O/bench_n$gb.json | cut -c1-300; done
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 BV_DP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-stream > $O/bench_n512_rccl.json 2> $O/bench_n512_rccl.err; tail -2 $O/bench_n512_rccl.err; cat $O/bench_n512_rccl.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-stream > $O/bench_line_profiled.json 2> $O/stats.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats512 -- python bench.py --global-batch 512 --steps 4 --warmup 2 --no-cpu-baseline --no-bf16-stream --no-roofline > $O/bench_n512_profiled.json 2> $O/stats512.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-bf16-stream > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-bf16-stream > $O/pmc_write.json 2> $O/pmc_write.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats512 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_n512.csv
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W --microbatch=2048 --n_gpus=1 > $O/pmc_traffic.json 2> $O/pmc_summary.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
timeout 900 python tools/bench_configs.py c2 c4 c5 c5b --steps 5 > $O/bench_configs.jsonl 2> $O/bench_configs.err; cut -c1-200 $O/bench_configs.jsonl
if [[ " $* " == *" yard "* ]]; then
  timeout 500 python tools/gemm_yardstick.py > $O/gemm_yardstick.txt 2> $O/gemm_yardstick.err; tail -12 $O/gemm_yardstick.txt | cut -c1-160
  timeout 200 python tools/gemm_epilogue_yardstick.py > $O/gemm_epilogue_yardstick.txt 2> /dev/null
  timeout 200 python tools/attn_yardstick.py > $O/attn_yardstick.txt 2> /dev/null
  timeout 200 python tools/hbm_yardstick.py > $O/hbm_yardstick.txt 2> /dev/null
  timeout 400 python tools/step_yardstick.py --batch 512 1024 256 > $O/step_yardstick.jsonl 2> /dev/null; cut -c1-200 $O/step_yardstick.jsonl
fi
ls -la $O
