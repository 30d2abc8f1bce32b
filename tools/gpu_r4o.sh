#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
for g in 4 0 4 0; do
  timeout 300 python tools/gemm_group_ab.py --step $g --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs > $O/step_g$g.json 2> $O/step_g$g.err
  echo "g=$g $(python -c "import json,sys; d=json.load(open('$O/step_g$g.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
