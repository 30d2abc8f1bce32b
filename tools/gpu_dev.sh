#!/bin/bash
# Short development calls for `gpurun` (each part: well under a GPU-minute of run time; results under gpurun_out/dev).
#   bash tools/gpu_dev.sh attn        the attention kernel tests + tools/attn_bench.py (all implementations, step shapes)
#   bash tools/gpu_dev.sh attn5       the one-launch attention backward probe (tools/probes/attn5_probe.hip: stamps, ablations)
#   bash tools/gpu_dev.sh gemm_order  the k-major tile order A/B (bv_gemm_group_n): per launch, FETCH_SIZE, whole step
#   bash tools/gpu_dev.sh quick       headline line (3 steps, no CPU baseline / configs) + the n = 512 rank shape
#   bash tools/gpu_dev.sh small       the GPU tests of the small serial kernels and of the optimizers
# The measurement set of a round is tools/gpu_final.sh.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/dev; mkdir -p $O
if [[ " $* " == *" attn "* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention and not other_head and not map_" -x 2>&1 | tail -5 | tee $O/pytest_attn.txt
  timeout 300 python tools/attn_bench.py 2>&1 | tee $O/attn_bench.txt | cut -c1-400
fi
if [[ " $* " == *" attn5 "* ]]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=0 -I big_vision_amd/csrc -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o /tmp/attn5_probe.out 2> /dev/null
  for L in 196 64; do timeout 120 /tmp/attn5_probe.out 2048 $L; done 2>&1 | tee $O/attn5_probe.txt
fi
if [[ " $* " == *" gemm_order "* ]]; then
  timeout 400 python tools/gemm_group_ab.py 2> /dev/null | tee $O/group_ab.txt
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python tools/gemm_group_pmc.py run > /dev/null 2> $O/group_pmc.err
  python tools/gemm_group_pmc.py parse $(find $O/pmc -name "*counter_collection.csv" | head -1) 2>&1 | tee $O/group_pmc.txt
  rm -rf $O/pmc
  for g in 4 0 4 0; do
    timeout 300 python tools/gemm_group_ab.py --step $g --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs > $O/step_g$g.json 2> /dev/null
    echo "group_n=$g $(python -c "import json; d=json.load(open('$O/step_g$g.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")" | tee -a $O/group_step.txt
  done
fi
if [[ " $* " == *" quick "* ]]; then
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-stream --no-configs > $O/headline.json 2> $O/headline.err
  python -c "import json; d=json.load(open('$O/headline.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'))"
  timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-stream --no-configs --no-roofline > $O/n512.json 2> $O/n512.err
  python -c "import json; d=json.load(open('$O/n512.json')); print('n512', d['value'], d['ms_per_step'])"
fi
if [[ " $* " == *" small "* ]]; then
  timeout 900 python -m pytest tests/test_adafactor_gpu.py tests/test_gemm256_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "adafactor or grouped or sgemm or embed or colsum or adam" 2>&1 | tail -5 | tee $O/pytest_small.txt
fi
