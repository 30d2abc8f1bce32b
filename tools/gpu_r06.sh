#!/bin/bash
# Round-6 measurement set, in parts so that a GPU call stays short (bash tools/gpu_r06.sh <part> ...):
#   tests      the whole -m gpu suite (what the driver runs at round end)
#   bench      the default bench line (headline + bf16 stream + configs + CPU baseline)
#   prof       rocprofv3 kernel stats + exclusive family busy time of the bench command, FETCH / WRITE / SQ --pmc passes
#   prof1      kernel stats of the bench command with both towers on ONE stream (exclusive per-kernel durations)
#   prof_cfg   kernel stats + FETCH / WRITE passes of BASELINE configs[3] (c4) and configs[4] (c5b)
#   prof_rank  FETCH / WRITE passes at the rank shapes of N = 2 / 4 / 8 (2048 / 1024 / 512 pairs on one GPU)
#   rccl3      the rank shape with the towers on two streams AND the one-rank RCCL collectives on their side stream
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
LEAN="--no-roofline --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc"
pmc_pair() {   # name, per_gpu_batch, workload label, command...
  local name=$1 pgb=$2 label=$3; shift 3
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${name}_fetch -- "$@" > /dev/null 2> $O/${name}_fetch.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${name}_write -- "$@" > /dev/null 2> $O/${name}_write.err
  local F=$(find $O/${name}_fetch -name "*counter_collection.csv" | head -1) W=$(find $O/${name}_write -name "*counter_collection.csv" | head -1)
  python tools/pmc_summary.py $F $W --microbatch=2048 --n_gpus=1 --per_gpu_batch=$pgb "--workload=$label" > $O/pmc_traffic_${name}.json 2>> $O/pmc_summary.err
  rm -rf $O/${name}_fetch $O/${name}_write
}
if [[ " $* " == *" tests "* ]]; then
  rm -f gpurun_out/parity_report.jsonl
  timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/pytest.txt; cat $O/pytest.txt
  cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
fi
if [[ " $* " == *" bench "* ]]; then
  timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-700 $O/bench_line.json
fi
if [[ " $* " == *" prof "* ]]; then
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc > $O/bench_line_profiled.json 2> $O/stats.err
  find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
  python tools/trace_family_busy.py $(find $O/stats -name "*kernel_trace.csv" | head -1) > $O/family_busy.json 2> $O/family_busy.err
  rm -rf $O/stats
  pmc_pair headline 4096 "headline (bench.py, 4096 pairs in micro-batches of 2048)" python bench.py --steps 1 --warmup 0 $LEAN
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python bench.py --steps 1 --warmup 0 $LEAN > /dev/null 2> $O/pmc_sq.err
  python tools/pmc_sq_summary.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq.json 2>> $O/pmc_summary.err
  rm -rf $O/pmc_sq
  head -24 $O/bench_kernel_stats.csv | cut -c1-150; cut -c1-600 $O/family_busy.json
fi
if [[ " $* " == *" prof1 "* ]]; then
  # per-kernel durations are exclusive only when the towers share one stream: the per-kernel roofline table is built from this trace
  BV_TOWER_STREAMS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc > $O/bench_line_profiled_one_stream.json 2> $O/stats1.err
  find $O/stats1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_one_stream.csv
  rm -rf $O/stats1
  head -16 $O/bench_kernel_stats_one_stream.csv | cut -c1-150
fi
if [[ " $* " == *" prof_cfg "* ]]; then
  for w in c4 c5b; do
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python tools/bench_configs.py $w --steps 3 > $O/${w}_line_profiled.json 2> $O/stats_$w.err
    find $O/stats_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${w}_kernel_stats.csv
    python tools/trace_family_busy.py $(find $O/stats_$w -name "*kernel_trace.csv" | head -1) 5 > $O/${w}_family_busy.json 2>> $O/family_busy.err
    rm -rf $O/stats_$w
    pmc_pair $w 0 "tools/bench_configs.py $w --steps 1 (BASELINE config)" python tools/bench_configs.py $w --steps 1
    head -16 $O/${w}_kernel_stats.csv | cut -c1-150
  done
fi
if [[ " $* " == *" prof_rank "* ]]; then
  for n in 512 1024 2048; do
    pmc_pair rank$n $n "bench.py --global-batch $n: the pairs one rank owns at N = $((4096 / n)), one GPU, no RCCL" python bench.py --global-batch $n --steps 1 --warmup 0 $LEAN
  done
fi
if [[ " $* " == *" rccl3 "* ]]; then
  for ts in 2 1 2 1; do
    BV_TOWER_STREAMS=$ts RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 BV_DP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream --no-live-pmc > $O/bench_n512_rccl_ts$ts.json 2> $O/bench_n512_rccl_ts$ts.err; tail -1 $O/bench_n512_rccl_ts$ts.err; python -c "import json,sys; d=json.load(open('$O/bench_n512_rccl_ts$ts.json')); print('tower_streams', $ts, 'rccl in loop:', round(d['ms_per_step'],2), 'ms', d['roofline']['frac'], d['config']['final_loss'], d.get('rccl'))"
    BV_TOWER_STREAMS=$ts timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream --no-live-pmc > $O/bench_n512_ts$ts.json 2> $O/bench_n512_ts$ts.err; python -c "import json,sys; d=json.load(open('$O/bench_n512_ts$ts.json')); print('tower_streams', $ts, 'no rccl:', round(d['ms_per_step'],2), 'ms', d['roofline']['frac'], d['roofline'].get('stream_overlap_factor'), d['config']['final_loss'])"
  done
fi
ls -la $O | head -40
