#!/bin/bash
# round-4 development run B: phase stamps + ablations of the one-launch attention backward
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b
rm -rf $O; mkdir -p $O
for a in 0 1 2 4 16 32 7 55; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=$a -DA5_STAMPS -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/attn5_probe_$a.out 2> /dev/null &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=0 -DA5_UNROLL_1A=13 -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/attn5_probe_u13.out 2> /dev/null &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=0 -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/attn5_probe_plain.out 2> /dev/null &
wait
for a in 0 1 2 4 16 32 7 55 plain u13; do timeout 120 /tmp/attn5_probe_$a.out 2048 196 >> $O/attn5_probe.txt 2>&1; done
timeout 60 /tmp/attn5_probe_0.out 2048 64 >> $O/attn5_probe.txt 2>&1
timeout 60 /tmp/attn5_probe_0.out 512 196 >> $O/attn5_probe.txt 2>&1
cat $O/attn5_probe.txt
