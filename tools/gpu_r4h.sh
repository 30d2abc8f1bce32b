#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_two_ranks_gpu.py tests/test_dp_nccl_gpu.py tests/test_bert_gpu.py tests/test_train_step_gpu.py tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4h/bench_line.json'))
print({k:(v if k not in ('config','roofline','configs','bf16_stream','cpu_baseline') else '...') for k,v in d.items()})
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','step_frac','share_of_step_time','traffic_measured_in_this_run')})
for k,v in d.get('configs',{}).items(): print(k, v)
print('bf16', d.get('bf16_stream',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
