#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
timeout 400 python tools/gemm_group_ab.py > $O/group_ab.txt 2> $O/group_ab.err; tail -3 $O/group_ab.err; cat $O/group_ab.txt
