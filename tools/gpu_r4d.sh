#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d
rm -rf $O; mkdir -p $O
i=0
while read -r v; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=0 $v -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/a5v_$i.out 2> /dev/null &
  echo "$i: $v" >> $O/variants.txt
  i=$((i+1))
done <<'VARS'
-DA5_UNROLL_1A=1
-DA5_UNROLL_1A=7
-DA5_UNROLL_1A=7 -DA5_UNROLL_1B=6 -DA5_SB_1B=0
-DA5_UNROLL_1A=7 -DA5_UNROLL_P2=6
-DA5_UNROLL_1A=7 -DA5_UNROLL_1B=6 -DA5_SB_1B=0 -DA5_UNROLL_P2=6
-DA5_UNROLL_1A=7 -DA5_UNROLL_1B=3 -DA5_SB_1B=0 -DA5_UNROLL_P2=6
-DA5_UNROLL_1A=4 -DA5_UNROLL_P2=3
VARS
wait
cat $O/variants.txt
for j in $(seq 0 $((i-1))); do echo "variant $j" >> $O/sweep.txt; timeout 60 /tmp/a5v_$j.out 2048 196 | grep -v "^  " >> $O/sweep.txt 2>&1; timeout 60 /tmp/a5v_$j.out 2048 64 | grep -v "^  " >> $O/sweep.txt 2>&1; done
cat $O/sweep.txt
