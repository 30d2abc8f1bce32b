#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4r; mkdir -p $O
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs > $O/headline.json 2> $O/headline.err
python -c "import json; d=json.load(open('$O/headline.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'))"
timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream --no-configs --no-roofline > $O/n512.json 2> $O/n512.err
python -c "import json; d=json.load(open('$O/n512.json')); print(d['value'], d['ms_per_step'])"
