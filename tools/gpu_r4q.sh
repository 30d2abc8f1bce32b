#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
for r in 1 2; do for f in tools/probes/attn3_sb_4_2_0.out tools/probes/attn3_sb_0_0_0.out tools/probes/attn3_sb_7_3_0.out tools/probes/attn3_sb_2_1_0.out tools/probes/attn3_sb_4_0_0.out; do echo "== $f" >> $O/sb.txt; timeout 60 $f 2048 >> $O/sb.txt 2>&1; done; done
timeout 60 tools/probes/attn3_sb_4_2_3.out 2048 >> $O/stamps.txt 2>&1
cat $O/sb.txt; cat $O/stamps.txt
