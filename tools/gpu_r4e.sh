#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e
rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=0 -DA5_STAMPS -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/attn5_probe_0.out 2> /dev/null &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=0 -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/attn5_probe_plain.out 2> /dev/null &
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention and not other_head and not map_" -x -s 2>&1 | grep -v "^$" | tail -12 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
wait
for a in 0 plain; do timeout 120 /tmp/attn5_probe_$a.out 2048 196 >> $O/attn5_probe.txt 2>&1; done
timeout 60 /tmp/attn5_probe_plain.out 2048 64 >> $O/attn5_probe.txt 2>&1
timeout 60 /tmp/attn5_probe_plain.out 512 196 >> $O/attn5_probe.txt 2>&1
timeout 60 /tmp/attn5_probe_plain.out 512 64 >> $O/attn5_probe.txt 2>&1
cat $O/attn5_probe.txt
timeout 300 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt
timeout 900 python -m pytest tests/test_siglip_step_gpu.py -q -m gpu -k "tiny or small_batch or lit_frozen or n64 or microbatched" 2>&1 | tail -8 > $O/pytest_e2e.txt; cat $O/pytest_e2e.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-stream > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-400 $O/bench.json
