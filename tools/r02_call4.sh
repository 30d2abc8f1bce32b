#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4
rm -rf $O; mkdir -p $O
timeout 300 tools/probes/gemm_roll_probe.out > $O/roll_probe.txt 2>&1; cut -c1-250 $O/roll_probe.txt | grep -v "^check"
