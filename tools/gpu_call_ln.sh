#!/bin/bash
# LN probe + the tests of the kernels touched + short bench lines.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ln
rm -rf $O; mkdir -p $O
timeout 200 tools/probes/ln_probe.out > $O/ln_probe.txt 2>&1; cat $O/ln_probe.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "layernorm or transpose or weight_images" 2>&1 | tail -3
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-bf16-stream > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-330 $O/bench_line.json
timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream --no-roofline > $O/bench_n512.json 2> $O/bench_n512.err; cut -c1-330 $O/bench_n512.json
