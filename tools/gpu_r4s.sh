#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
timeout 900 python -m pytest tests/test_adafactor_gpu.py tests/test_gemm256_gpu.py -q -m gpu -x -k "adafactor or grouped" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 600 python -m pytest tests/test_dp_two_ranks_gpu.py -q -m gpu -x -k "kw6 or FSDP_AF or adafactor" 2>&1 | tail -5 | tee -a $O/pytest.txt
