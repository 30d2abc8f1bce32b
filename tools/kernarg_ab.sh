cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v rank512:"; HIP_FORCE_DEV_KERNARG=$v python bench.py --global-batch 512 --steps 12 --warmup 3 --no-cpu-baseline --no-bf16-stream --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_idle_gpu'))"
done
done
for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v headline:"; HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-bf16-stream --no-roofline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('host_enqueue_ms_idle_gpu'))"
done
