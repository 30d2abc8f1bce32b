#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_two_ranks_gpu.py tests/test_dp_nccl_gpu.py tests/test_adafactor_gpu.py -q -m gpu -k "kw5 or kw6 or fsdp or FSDP or adafactor or batched" -x 2>&1 | tail -12 > $O/pytest.txt; cat $O/pytest.txt
