#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c9
rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_report.jsonl
BV_PARITY_REPORT_ONLY=1 timeout 900 python -m pytest tests/test_siglip_step_gpu.py tests/test_train_step_gpu.py tests/test_vit_tower_gpu.py tests/test_evaluators_gpu.py -q -x 2>&1 | tail -8 > $O/pytest_e2e.txt; cat $O/pytest_e2e.txt
cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl
