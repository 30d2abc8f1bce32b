# kernel-trace of the headline step on ONE stream and on two, gap analysis of the last step (tools/trace_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 1 2; do
  rm -rf /tmp/gp$s
  BV_TOWER_STREAMS=$s timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp$s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs --no-live-pmc --no-roofline > /dev/null 2> /tmp/gp$s.err
  F=$(find /tmp/gp$s -name "*kernel_trace.csv" | head -1)
  echo "=== tower_streams = $s"
  python $R/tools/trace_gaps.py $F | head -75
done
