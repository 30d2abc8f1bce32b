#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i
rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=0 -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/attn5_probe_plain.out 2> /dev/null &
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention and not other_head and not map_" -x 2>&1 | tail -5 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
wait
timeout 120 /tmp/attn5_probe_plain.out 2048 196 >> $O/attn5_probe.txt 2>&1
cat $O/attn5_probe.txt
