#!/bin/bash
# Round-end measurement set (rounds 4-5), in parts so that a GPU call stays short:
#   bash tools/gpu_final.sh prof     rocprofv3 kernel stats of the bench command (N = 1) and of the n = 512 rank shape, the two
#                                    --pmc passes (FETCH_SIZE / WRITE_SIZE, counters only - never combined with tracing domains)
#   bash tools/gpu_final.sh bench    the default bench line (headline + bf16 stream + configs + CPU baseline), the rank shapes
#                                    with the one-rank RCCL collectives in the loop
#   bash tools/gpu_final.sh tests    the whole -m gpu suite (what the driver runs at round end)
#   bash tools/gpu_final.sh yard     the vendor yardsticks (GEMM, fused epilogues, attention, LayerNorm / Adam, whole step)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
if [[ " $* " == *" tests "* ]]; then
  rm -f gpurun_out/parity_report.jsonl
  BV_PARITY_FLOOR=1 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
  cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
fi
if [[ " $* " == *" bench "* ]]; then
  timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-600 $O/bench_line.json
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 BV_DP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-stream > $O/bench_n512_rccl.json 2> $O/bench_n512_rccl.err; tail -2 $O/bench_n512_rccl.err; cut -c1-300 $O/bench_n512_rccl.json
  timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream > $O/bench_n512.json 2> $O/bench_n512.err; cut -c1-300 $O/bench_n512.json
fi
if [[ " $* " == *" prof "* ]]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc > $O/bench_line_profiled.json 2> $O/stats.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats512 -- python bench.py --global-batch 512 --steps 4 --warmup 2 --no-cpu-baseline --no-bf16-stream --no-roofline > $O/bench_n512_profiled.json 2> $O/stats512.err
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc > $O/pmc_fetch.json 2> $O/pmc_fetch.err
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc > $O/pmc_write.json 2> $O/pmc_write.err
  find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
  find $O/stats512 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_n512.csv
  F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py $F $W --microbatch=2048 --n_gpus=1 > $O/pmc_traffic.json 2> $O/pmc_summary.err
  # MFMA busy / wave cycles / LDS conflicts / waits per kernel (one pass, counters only)
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc > /dev/null 2> $O/pmc_sq.err
  python tools/pmc_sq_summary.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq.json 2>> $O/pmc_summary.err
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
  head -30 $O/kernel_stats.csv | cut -c1-150
fi
if [[ " $* " == *" yard "* ]]; then
  timeout 500 python tools/gemm_yardstick.py > $O/gemm_yardstick.txt 2> $O/gemm_yardstick.err; tail -12 $O/gemm_yardstick.txt | cut -c1-160
  timeout 200 python tools/gemm_epilogue_yardstick.py > $O/gemm_epilogue_yardstick.txt 2> /dev/null
  timeout 200 python tools/attn_yardstick.py > $O/attn_yardstick.txt 2> /dev/null
  timeout 200 python tools/hbm_yardstick.py > $O/hbm_yardstick.txt 2> /dev/null
  timeout 400 python tools/step_yardstick.py --batch 512 1024 256 > $O/step_yardstick.jsonl 2> /dev/null; cut -c1-200 $O/step_yardstick.jsonl
fi
ls -la $O
