#!/bin/bash
# GPU call 2 (round 3): GEMM + step tests after the epilogue rewrite, bench lines (N=1 default, n=512 rank shape),
# rocprofv3 kernel stats of the bench command, name of the hipBLASLt kernel behind the yardstick.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm256_gpu.py -x -q 2>&1 | tail -6 > $O/pytest_gemm.txt; cat $O/pytest_gemm.txt
timeout 900 python -m pytest tests/test_siglip_step_gpu.py -x -q -k "tiny_two_towers or b16_siglip_step_small or microbatched or bench_mode" 2>&1 | tail -8 > $O/pytest_step.txt; cat $O/pytest_step.txt
timeout 400 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench_line.json
timeout 300 python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n512.json 2> $O/bench_n512.err; cat $O/bench_n512.json | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-stream > $O/bench_line_profiled.json 2> $O/stats.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
cat > /tmp/mm.py <<'PY'
import torch
torch.backends.cuda.preferred_blas_library("hipblaslt")
for (M,N,K) in [(131072,2304,768),(131072,768,3072),(401408,3072,768)]:
  a=torch.randn(M,K,device="cuda",dtype=torch.bfloat16); b=torch.randn(N,K,device="cuda",dtype=torch.bfloat16)
  for _ in range(3): c=a@b.t()
  torch.cuda.synchronize()
PY
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mm -- python /tmp/mm.py > /dev/null 2> $O/mm.err
find $O/mm -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vendor_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
cut -c1-300 $O/vendor_kernel_stats.csv | head -8
ls $O
