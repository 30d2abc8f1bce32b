"""Headline benchmark: image-text pairs/sec of one full SigLIP training step.

Workload (BASELINE.json `metric`, SURVEY.md §8d / Appendix B "C3"): ViT-B/16@224
image tower (MAP pooling) + 12-layer text transformer B (64 tokens, vocab 32 000),
pairwise sigmoid loss over the GLOBAL batch 4096, Adam + clip + wd + cosine
schedule.  One "step" = forward + backward + gradient sync + optimizer update
on one batch of synthetic pairs already resident in HBM.  The global batch is
fixed at 4096 for every N ("strong" scaling): each of the N ranks owns 4096/N
pairs and processes them in micro-batches of 2048 (two-pass embedding scheme of
big_vision_amd/trainers/proj/image_text/siglip.py when 4096/N > 2048, i.e. N = 1;
single pass for N >= 2).

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     : the dominant kernel (the 256x256 k-major bf16 MFMA GEMM: forward + dX) timed
                 live with HIP events on the launch stream over the timed steps;
                 achieved = algorithmic FLOPs (2*M*N*K per launch) / event time.
  cpu_baseline : the CPU oracle (oracle/bv_oracle.py, kind "port") running the
                 same model's full step on a bounded sample of pairs on the
                 host cores (rank 0, N=1 only).  Checker/baseline only — the
                 timed product path never touches oracle/.
  configs      : (N = 1, after the headline, outside its timed region) the other BASELINE.json
                 workloads and the per-rank shapes of the headline on this one GPU, a few steps each:
                 c2 (ViT-B/16 tower fwd+bwd, batch 256), c4_rank (SigLIP L/16@336, the 1024 pairs one of
                 8 ranks owns), c5b (LiT step with the BERT-base text tower, batch 512), rank512 /
                 rank1024 (the 4096 / 8 and 4096 / 4 pairs one rank of the headline owns; *_one_stream: the same with
                 config.tower_streams = 1, the A/B partner of the two-stream default), rank512_rccl.  Each entry:
                 value, unit, ms_per_step, steps, roofline_frac (the same GEMM family, live HIP
                 events), step_frac where the workload's matmul FLOPs are known, peak_hbm_gb.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

# Kernel arguments in device memory (read by the HIP runtime when it initialises, i.e. at the first GPU call): a step is
# ~1600 launches; measured -0.5 % at the 512-pair rank shape (86.6 -> 86.2 ms, profiles/r06_kernarg_ab.txt), nothing on the
# headline.  A deployment sets it in its own environment; an explicit value is left alone.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

GLOBAL_BATCH = 4096
MICRO = 2048   # pairs per micro-batch and rank (ranks with fewer pairs run a single pass)
RES, SEQ, VOCAB, EMB = 224, 64, 32_000, 768
IMAGE_CFG = dict(variant="B/16", pool_type="map")
TEXT_CFG = dict(variant="B", vocab_size=VOCAB)
BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
DOMINANT = (("bv_gemm_bf16", "bv_gemm_bf16_colsum"), 1, 1)   # k-major ("NT") GEMM: forward (W^T shadow) and dX projections
# rocprofv3 --pmc passes of this command (tools/pmc_summary.py), newest first: the headline and the rank shapes of N = 2 / 4 / 8
PMC_PROFILES = ("r06_pmc_traffic.json", "r06_pmc_traffic_rank2048.json", "r06_pmc_traffic_rank1024.json",
                "r06_pmc_traffic_rank512.json", "r05_pmc_traffic.json")
DOMINANT_KERNEL = "gemm256_kernel<true> + gemm256r_kernel (256x256 k-major bf16 MFMA GEMM, all epilogues)"


class GemmObserver:
  """Brackets every launch of the dominant kernel with HIP events on the stream it is launched on (torch's current
  stream at the call: the main stream for the image tower, the side stream for the text tower since the towers run on
  two streams by default) and reports the family's EXCLUSIVE BUSY TIME: every event is placed on one time line
  (elapsed time from a base event recorded when the timed region starts) and the launch intervals are merged - the
  time during which at least one launch of the family was in flight.  With a single stream this is the sum of the
  launch durations; with two streams two overlapping launches (an image-tower GEMM beside a text-tower GEMM) count
  their FLOPs twice and their common time once, which is what the chip delivered.  A family launch that overlaps a
  kernel of another family on the other stream is charged the whole interval (the figure errs low, never high)."""

  def __init__(self):
    self.recs = []
    self.active = False
    self.base = None

  def start(self):
    """Call with the device idle, right before the timed region."""
    self.base = torch.cuda.Event(enable_timing=True)
    self.base.record()
    self.active = True

  def begin(self, name, args):
    if not self.active or name not in DOMINANT[0] or (args[0], args[1]) != DOMINANT[1:]:
      return None
    if (args[9] & 255) or (args[10] & 255) or (args[11] & 63):
      return None   # small/ragged problems run on the general 128x128 kernel, not the dominant one
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    M, N, K, epi, out_f32 = args[9], args[10], args[11], args[12], args[8]
    # algorithmic HBM bytes of the launch: A + B read once, C written once, plus the epilogue's
    # auxiliary operand (fp32 residual / bf16 pre-activation) and second output (GELU / EMIT)
    nbytes = 2.0 * M * K + 2.0 * N * K + M * N * (4.0 if out_f32 else 2.0)
    nbytes += {1: 4.0, 4: 2.0, 6: 2.0}.get(epi, 0.0) * M * N      # aux
    nbytes += {3: 2.0, 6: 2.0}.get(epi, 0.0) * M * N              # C2
    # (kernel variant of the launch, as rocprofv3 names it: gemm256_kernel<true, 0, EPI, OUTF32> / the rolling-epilogue
    # gemm256r_kernel for the plain, GELU and fp32 +residual epilogues - the split is the library's, the key here is
    # just (epilogue id, fp32 output))
    return (e0, e1, 2.0 * M * N * K, nbytes, (int(epi), bool(out_f32)))

  def end(self, tok):
    tok[1].record()
    self.recs.append(tok)

  def intervals(self):
    """[(start_ms, end_ms)] of every launch on the common time line."""
    base = self.base
    if base is None:
      return [(0.0, r[0].elapsed_time(r[1])) for r in self.recs]   # (no base: durations only, laid end to end below)
    return [(base.elapsed_time(r[0]), base.elapsed_time(r[1])) for r in self.recs]

  def by_epilogue(self):
    """{"epi<id>[_f32]": {launches, algorithmic bytes per launch, GFLOP per launch}}: the algorithmic side of the
    per-variant PMC bytes in profiles/rNN_pmc_traffic.json (epilogue ids: include/bvhip.h BV_EPI_*)."""
    acc = {}
    for r in self.recs:
      epi, f32 = r[4] if len(r) > 4 else (-1, False)
      a = acc.setdefault(f"epi{epi}" + ("_f32" if f32 else ""), [0, 0.0, 0.0])
      a[0] += 1; a[1] += r[3]; a[2] += r[2]
    return {k: {"launches": n, "algorithmic_bytes_per_launch": b / n, "gflop_per_launch": f / n / 1e9} for k, (n, b, f) in sorted(acc.items())}

  def summary(self):
    """(launches, exclusive busy ms, FLOPs, algorithmic bytes, sum of the per-launch event times in ms)."""
    iv = self.intervals()
    total = sum(b - a for a, b in iv)
    busy = union_ms(iv) if self.base is not None else total
    return len(self.recs), busy, sum(r[2] for r in self.recs), sum(r[3] for r in self.recs), total


def union_ms(intervals):
  """Length of the union of [start, end) intervals (ms)."""
  busy, cur_a, cur_b = 0.0, None, None
  for a, b in sorted(intervals):
    if cur_b is None or a > cur_b:
      if cur_b is not None:
        busy += cur_b - cur_a
      cur_a, cur_b = a, b
    elif b > cur_b:
      cur_b = b
  if cur_b is not None:
    busy += cur_b - cur_a
  return busy


def roofline_object(obs, wall_s):
  """The `roofline` object of a timed region from its observer (wall_s: wall time of that region, seconds)."""
  launches, ms, flops, nbytes, sum_ms = obs.summary()
  ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
  return {"bound": "mfma", "kernel": DOMINANT_KERNEL, "achieved": ach, "peak": BF16_DENSE_PEAK_TFLOPS,
          "unit": "TFLOP/s", "frac": ach / BF16_DENSE_PEAK_TFLOPS, "traffic": None,
          "timing": "exclusive busy time of the kernel family: union of its launch intervals, HIP events on each launch "
                    "stream over the timed region (two streams: overlapping launches share their common time)",
          "algorithmic_bytes_per_launch": nbytes / max(1, launches), "launches": launches,
          "avg_launch_us": 1e3 * ms / max(1, launches),
          "sum_of_launch_event_ms": sum_ms, "busy_ms": ms,
          "stream_overlap_factor": (sum_ms / ms) if ms > 0 else None,
          "share_of_step_time": ms / (1e3 * wall_s) if wall_s > 0 else None,
          "by_epilogue": obs.by_epilogue()}


def pmc_traffic(world, per_gpu_batch, micro):
  """HBM bytes per launch of the dominant kernel family from the COMMITTED rocprofv3 --pmc passes (counters cannot be
  read from inside the process; N = 1 replaces this with passes of its own, live_pmc_traffic).  The profile is picked
  by what ONE GPU runs - the pairs per GPU and the micro-batch size decide every launch shape - so an N > 1 line uses
  the passes taken on one GPU at its rank shape (`bench.py --global-batch <pairs per GPU>`; no RCCL beside the
  kernels in that profile, which the `traffic_source` string says).  (None, None) when no profile matches."""
  passes = min(per_gpu_batch, micro)
  for name in PMC_PROFILES:
    path = os.path.join(ROOT, "profiles", name)
    try:
      with open(path) as f:
        d = json.load(f)
      if d.get("per_gpu_batch", GLOBAL_BATCH) != per_gpu_batch or min(d.get("microbatch", MICRO), per_gpu_batch) != passes:
        continue
      src = os.path.relpath(path, ROOT)
      if world != d.get("n_gpus", 1):
        src += f" (one GPU at the {per_gpu_batch}-pair rank shape of N = {world})"
      return d["kernels"][DOMINANT_KERNEL]["hbm_bytes"], src
    except (OSError, KeyError, ValueError):
      continue
  return None, None


def pmc_family_totals(csv_path):
  """(sum of Counter_Value in KiB, number of launches) over the dominant kernel family in one rocprofv3
  counter_collection.csv (one counter per file)."""
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import pmc_summary
  t, c, _ = pmc_summary.load(csv_path)
  fam = [k for k in t if k.startswith("gemm256_kernel<true") or k.startswith("gemm256r_kernel")]
  return sum(t[k] for k in fam), sum(c[k] for k in fam)


def live_pmc_traffic(micro, timeout_s=120):
  """HBM bytes per launch of the dominant kernel family measured IN THIS RUN: two short rocprofv3 passes of this very
  command (`--steps 1 --warmup 0`, everything optional switched off) as child processes on the same GPU, one per
  counter - FETCH_SIZE and WRITE_SIZE in SEPARATE passes with --kernel-trace only, as MI355X_MICROARCH.md (HBM /
  rocprofv3 section) prescribes - corrected as tools/pmc_summary.py does (FETCH_SIZE x 2 on gfx950).  Called after the
  timed region with this process's HBM released.  Returns (bytes per launch | None, detail dict)."""
  import glob
  import shutil
  import subprocess
  import tempfile
  rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
  if not os.path.exists(rocprof):
    return None, {"error": "rocprofv3 not found"}
  tot = {}
  t_all = time.perf_counter()
  for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out = tempfile.mkdtemp(prefix=f"bv_pmc_{counter.lower()}_", dir="/tmp")
    cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--microbatch", str(micro), "--no-roofline",
           "--no-cpu-baseline", "--no-bf16-stream", "--no-configs", "--no-live-pmc"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    try:
      pr = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
      shutil.rmtree(out, ignore_errors=True)
      return None, {"error": f"rocprofv3 --pmc {counter} pass exceeded {timeout_s} s"}
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if pr.returncode != 0 or not files:
      shutil.rmtree(out, ignore_errors=True)
      return None, {"error": f"rocprofv3 --pmc {counter} pass failed (rc {pr.returncode}): {pr.stderr[-300:]}"}
    tot[counter] = pmc_family_totals(files[0])
    shutil.rmtree(out, ignore_errors=True)
  (fk, fn), (wk, wn) = tot["FETCH_SIZE"], tot["WRITE_SIZE"]
  if not fn or not wn:
    return None, {"error": "no launches of the dominant kernel in the counter files"}
  rd, wr = fk * 1024 * 2 / fn, wk * 1024 / wn
  return rd + wr, {"read_bytes": rd, "write_bytes": wr, "launches_per_step": fn, "passes_s": round(time.perf_counter() - t_all, 1),
                   "command": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE} --kernel-trace -- python bench.py --steps 1 --warmup 0 "
                              "--no-roofline --no-cpu-baseline --no-bf16-stream --no-configs (one child process per counter)",
                   "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE KiB x 1024"}


# Activations between the encoder blocks (config.residual_stream).  `value` is measured on "float32": the
# reference's arithmetic (models/vit.py:92-110 keeps the residual stream in fp32 even with dtype_mm=bfloat16,
# only matmul inputs are cast).  "bfloat16" (LayerNorm inputs, the +residual GEMM epilogues, the
# saved block inputs and the gradient stream in bf16; statistics / softmax / loss / optimizer fp32) is an
# opt-in trainer mode that is NARROWER than the reference: at N = 1 the line carries it as a second,
# clearly separate object ("bf16_stream": value, ms_per_step and the worst per-tensor gradient rel-L2 the
# -m gpu step tests measured for that mode), never as `value`.
RESIDUAL_STREAM = "float32"
# the -m gpu case that runs exactly the bf16 object's mode: B/16 + text-B through micro-batches with gelu(h)-free contexts
BF16_STREAM_PARITY = ("profiles/r05_parity_report.jsonl", "siglip B/16 n=32 microbatch=8 gelu(h)-free contexts, bfloat16 stream")


# siglip.make_update_fn's state_cache["light"] -> what a kept micro-batch context holds
CTX_KIND = {False: "full", None: "full", "g": "gelu(h)-free (re-emitted by the fc2 dX GEMM)", True: "light",
            "light": "light (no gelu(h), no LayerNorm outputs: both re-derived in the backward)"}


def make_config(total_steps):
  from big_vision_amd.compat.ml_collections import ConfigDict
  c = ConfigDict()
  # Optimizer settings of the only in-repo SigLIP-trainer config
  # (configs/proj/image_text/siglip_lit_coco.py:93-106), see SURVEY.md App. B.
  c.optax_name = "scale_by_adam"
  c.lr = 1e-3
  c.wd = 1e-2
  # siglip_lit_coco.py:100: warmup = max(0.03 * total, 100) steps.
  c.schedule = dict(decay_type="cosine", warmup_steps=max(100, int(0.03 * total_steps)))
  c.grad_clip_norm = 1.0
  c.total_steps = total_steps
  c.microbatch = MICRO
  if os.environ.get("BV_TOWER_STREAMS"):      # A/B of the trainer option config.tower_streams (default: the trainer's)
    c.tower_streams = int(os.environ["BV_TOWER_STREAMS"])
  return c


def synthetic_batch(n, dev, seed):
  """U(-1,1) NHWC fp32 images + sticky-EOS int32 tokens, on device."""
  g = torch.Generator(device=dev).manual_seed(seed)
  image = torch.rand((n, RES, RES, 3), generator=g, device=dev, dtype=torch.float32) * 2 - 1
  text = torch.randint(2, VOCAB, (n, SEQ), generator=g, device=dev, dtype=torch.int32)
  lens = torch.randint(4, SEQ, (n,), generator=g, device=dev)
  pos = torch.arange(SEQ, device=dev)[None, :]
  text = torch.where(pos >= lens[:, None], torch.ones_like(text), text)
  return image.contiguous(), text.contiguous()


def cpu_baseline(sample_pairs, samples=2):
  """Full SigLIP step (fwd + bwd + Adam chain) of the SAME model on the CPU oracle."""
  sys.path.insert(0, os.path.join(ROOT, "oracle"))
  import bv_oracle as O
  cores = torch.get_num_threads()
  params = O.init_two_towers(0, (RES, RES), SEQ, image_cfg=IMAGE_CFG, text_cfg=TEXT_CFG,
                             out_dim=(None, EMB), temperature_init=10.0, bias_init=-10.0,
                             dtype=torch.float32)
  params = O.tree_map(lambda v: v.requires_grad_(True), params)
  cfg = make_config(20_000).to_dict()
  cfg.pop("microbatch")
  tx = O.OptaxOracle(cfg, O.tree_map(lambda v: v.detach(), params),
                     sched_kw=dict(total_steps=20_000, batch_size=sample_pairs))

  def step(n):
    image, text = O.synthetic_batch(1, n, RES, SEQ, VOCAB)
    loss, _ = O.siglip_step_loss(params, image, text, image_cfg=IMAGE_CFG, text_cfg=TEXT_CFG,
                                 out_dim=(None, EMB))
    loss.backward()
    grads = O.tree_map(lambda v: v.grad, params)
    with torch.no_grad():
      upd = tx.update(grads, O.tree_map(lambda v: v.detach(), params))
      for (_, p), (_, du) in zip(O.tree_flatten_with_names(params), O.tree_flatten_with_names(upd)):
        p.add_(du)
        p.grad = None
    return float(loss.detach())

  step(2)  # page in / thread-pool warm-up
  # two sample sizes ("n = 8 / 16 / 32 ... scales ~linearly in n"): half and one-and-a-half times the
  # requested sample, together the CPU work of two samples of it (~30-40 s); `value` is the larger one's rate
  sizes = sorted({max(2, sample_pairs // 2), max(2, sample_pairs * 3 // 2)}) if samples > 1 and sample_pairs > 2 else [max(2, sample_pairs)]   # (--cpu-sample 2: smoke runs, one sample)
  times = []
  for n in sizes:
    if times and times[-1] / sizes[len(times) - 1] * n > 90.0:
      break          # a slow host: the larger sample would take more than 90 s - report the smaller one alone
    t0 = time.perf_counter()
    step(n)
    times.append(time.perf_counter() - t0)
  sizes = sizes[:len(times)]
  n, dt = sizes[-1], times[-1]
  return {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "samples": len(times),
          "sample_pairs": sizes, "sample_seconds": [round(t, 2) for t in times],
          "pairs_per_s_by_sample": {str(k): round(k / t, 3) for k, t in zip(sizes, times)},
          "sample": f"one full step (fwd+bwd+Adam) of the same ViT-B/16+text-B model on {n} pairs, fp32 torch-CPU oracle "
                    f"(a port: the reference's CPU-JAX path cannot run on this host), {dt:.1f} s"
                    + (f"; a second sample on {sizes[0]} pairs shows the rate's dependence on the sample size" if len(sizes) > 1 else "")}


def bf16_stream_parity():
  """Worst per-tensor gradient rel-L2 / cosine of the bf16-stream B/16 n=32 micro-batched step vs the fp64
  oracle, as the -m gpu suite last measured it (the committed report; bench.py never runs the oracle on
  the timed path)."""
  path, case = BF16_STREAM_PARITY
  try:
    with open(os.path.join(ROOT, path)) as f:
      rows = [json.loads(l) for l in f if l.strip()]
    row = [x for x in rows if x.get("case") == case][-1]
    return {"parity_worst_rel_l2": row["worst_rel"], "parity_worst_cos": row["worst_cos"],
            "parity_case": case, "parity_source": path}
  except (OSError, IndexError, KeyError, ValueError):
    return {"parity_worst_rel_l2": None, "parity_source": None}


def rccl_info(comm, dev):
  """Evidence that the N ranks of this line really joined one RCCL communicator: backend name, RCCL
  version, and the result of an all-reduce of ones over the group (= the number of ranks that took part)."""
  import torch.distributed as dist
  info = {"ranks": comm.size, "backend": None, "version": None, "allreduce_of_ones": None}
  if dist.is_available() and dist.is_initialized():
    info["backend"] = dist.get_backend()
    try:
      v = torch.cuda.nccl.version()
      info["version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
      pass
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    info["allreduce_of_ones"] = float(one.item())
    info["max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
  # the two knobs of the overlapped gradient sync and the environment variables that override them without a code
  # change (A/B on the first multi-GPU node: BV_RESERVED_CUS=4|8|0, BV_NCCL_MAX_NCHANNELS=4|8|uncapped)
  from big_vision_amd import dp
  info["reserved_cus"] = dp.RESERVED_CUS
  info["env_overrides"] = {k: os.environ.get(k) for k in ("BV_RESERVED_CUS", "BV_NCCL_MAX_NCHANNELS", "BV_TOWER_STREAMS")}
  return info


def _free_port():
  import socket
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    return sk.getsockname()[1]


def spawn_ranks(n):
  """`python bench.py --gpus N` without a launcher: re-run this command as N ranks (one process per
  GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, rendezvous on 127.0.0.1), exactly
  what `python -m torch.distributed.run --nproc-per-node N bench.py ...` would start."""
  import subprocess
  # BV_BENCH_SHARE_GPU=1 (with BV_DP_BACKEND=gloo): every rank on GPU 0 - a functional run of the N > 1 line on a one-GPU
  # box (tests/test_dp_nccl_gpu.py); its `value` measures nothing and the line says so (config.shared_gpu)
  share = os.environ.get("BV_BENCH_SHARE_GPU") == "1"
  if torch.cuda.is_available() and torch.cuda.device_count() < n and not share:
    raise RuntimeError(f"--gpus {n} but only {torch.cuda.device_count()} GPU(s) are visible")
  port = str(_free_port())
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if share else str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
  # Poll ALL children: a rank that dies while the others sit in an RCCL collective would otherwise leave a
  # sequential wait() blocked forever on a healthy-looking rank.  First non-zero exit -> terminate the rest.
  rc = 0
  live = list(procs)
  while live and not rc:
    time.sleep(0.2)
    for p in list(live):
      if p.poll() is not None:
        live.remove(p)
        rc = rc or p.returncode
  if rc:
    for p in live:
      p.terminate()
    deadline = time.time() + 10
    for p in live:
      try:
        p.wait(timeout=max(0.1, deadline - time.time()))
      except subprocess.TimeoutExpired:
        p.kill()
    raise SystemExit(rc)


# ------------------------------------------------------------------ other BASELINE workloads (N = 1 `configs` object) --
MATMUL_GFLOP_PER_PAIR = 139.3       # B/16 + text-B, forward + backward (3 x (35.42 image + 11.02 text + 0.006 loss)), DESIGN.md 4
MATMUL_GFLOP_PER_IMAGE_B16 = 3 * 35.42


def _timed(fn, steps, warmup):
  """(seconds per step, roofline object of the k-major 256x256 GEMM family over the same timed region)."""
  from big_vision_amd import _lib
  for _ in range(warmup):
    fn()
  obs = GemmObserver()
  _lib.observer = obs
  torch.cuda.synchronize()
  obs.start()
  t0 = time.perf_counter()
  for _ in range(steps):
    fn()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  obs.active = False
  _lib.observer = None
  roof = roofline_object(obs, dt * steps)
  return dt, roof


def workload_c2(dev, steps, stream="float32"):
  """BASELINE configs[1]: ViT-B/16 image tower alone, forward + backward, batch 256."""
  from big_vision_amd import engine as E
  from big_vision_amd.models import vit
  from big_vision_amd.params import ParamStore
  old = E.set_residual_stream(stream)
  try:
    n, res = 256, 224
    model = vit.Model(None, variant="B/16", pool_type="map")
    hw = model.grid((n, res, res, 3))
    store = ParamStore(model.entries("", hw), dev)
    store.init_random(0); store.refresh_shadow(); store.want_grads = True
    image = torch.rand((n, res, res, 3), device=dev) * 2 - 1
    ex = model.executor(store, "", hw)

    def step():
      store.zero_grad()
      z, _, ctx = ex.fwd(image, save=True)
      ex.bwd(ctx, (z / n).contiguous())          # dL/dz of L = 0.5 mean |z|^2
    dt, roof = _timed(step, steps, 2)
  finally:
    E.set_residual_stream(old)
  flops = MATMUL_GFLOP_PER_IMAGE_B16 * 1e9 * n
  return {"metric": "images/sec, ViT-B/16 image tower forward+backward, batch 256 (BASELINE configs[1])",
          "value": n / dt, "unit": "images/s", "ms_per_step": 1e3 * dt, "tflops_algorithmic": flops / dt / 1e12,
          "step_frac": flops / dt / 1e12 / BF16_DENSE_PEAK_TFLOPS, "roofline": roof,
          "config": {"workload": "ViT-B/16@224 MAP tower, fwd+bwd, no optimizer", "batch": n, "residual_stream": stream}}


def workload_siglip(dev, steps, image_cfg, text_cfg, emb, n, res, seq, micro, schedule=None, label="", text_model=None,
                    vocab=32_000, stream="float32", comm=None, gflop_per_pair=None, tower_streams=None):
  """One SigLIP training step (trainers.proj.image_text.siglip) on n pairs of synthetic data, this device only."""
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, emb), temperature_init=10.0,
                           bias_init=-10.0 if schedule is None else -2.71,
                           **({"text_model": text_model} if text_model else {}))
  config = make_config(20_000)
  config.microbatch = micro
  config.residual_stream = stream
  if tower_streams:
    config.tower_streams = tower_streams
  if schedule is not None:
    config.schedule = schedule
  g = torch.Generator(device=dev).manual_seed(1)
  image = torch.rand((n, res, res, 3), generator=g, device=dev) * 2 - 1
  text = torch.randint(2, vocab, (n, seq), generator=g, device=dev, dtype=torch.int32)
  kw = {} if comm is None else {"comm": comm}
  state, _ = siglip.make_train_state(model, config, (n, res, res, 3), (n, seq), rng=0, total_steps=20_000, device=dev, **kw)
  fn = siglip.make_update_fn(model, config, **kw)
  box = {"s": state}

  def step():
    box["s"], box["m"] = fn(box["s"], None, {"image": image, "labels": text})
  dt, roof = _timed(step, steps, 2)
  siglip.check_finite(box["m"])
  r = {"value": n / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt, "roofline": roof,
       "config": {"workload": label, "per_gpu_batch": n, "microbatch": micro, "residual_stream": stream,
                  "final_loss": float(box["m"]["training_loss"].item())}}
  if gflop_per_pair:
    r["step_frac"] = gflop_per_pair * 1e9 * n / dt / 1e12 / BF16_DENSE_PEAK_TFLOPS
  return r


LIT_SCHEDULE = [("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=150))]


def workload_c4(dev, steps, stream="float32"):
  """BASELINE configs[3]: SigLIP ViT-L/16@336 + text-L, global batch 8192 on 8 GPUs: ONE rank's 1024 pairs."""
  r = workload_siglip(dev, steps, dict(variant="L/16", pool_type="map"), dict(variant="L", vocab_size=32_000), 1024,
                      n=1024, res=336, seq=64, micro=256, stream=stream,   # (A/B round 6: 512 -> 769 pairs/s, 256 -> 823)
                      gflop_per_pair=981.4,   # matmul FLOPs of one L/16@336 + text-L pair per step
                      label="SigLIP ViT-L/16@336 + text-L: one rank's 1024 pairs of the global batch 8192 (loss over the "
                            "local 1024 only: no peers on a single device), micro-batches of 256, Adam+clip+wd+cosine")
  r["metric"] = "image-text pairs/sec per GPU, SigLIP ViT-L/16@336 training step at 1024 pairs per GPU (BASELINE configs[3])"
  return r


def workload_c5(dev, steps, stream="float32"):
  r = workload_siglip(dev, steps, dict(variant="B/16", pool_type="tok", head_zeroinit=False),
                      dict(variant="B", vocab_size=32_000), 768, n=512, res=224, seq=16, micro=2048, schedule=LIT_SCHEDULE,
                      stream=stream, label="LiT (siglip_lit_coco.py): frozen ViT-B/16 cls-token tower + trainable text-B, "
                                           "16 tokens, batch 512, text-only backward")
  r["metric"] = "image-text pairs/sec, LiT locked-image step, batch 512 (BASELINE configs[4])"
  return r


def workload_c5b(dev, steps, stream="float32"):
  """The literal siglip_lit_coco.py: text_model='proj.flaxformer.bert', config 'base' (:78,84-87)."""
  r = workload_siglip(dev, steps, dict(variant="B/16", pool_type="tok", head_zeroinit=False),
                      dict(config="base", head_zeroinit=False), 768, n=512, res=224, seq=16, micro=2048,
                      schedule=LIT_SCHEDULE, text_model="proj.flaxformer.bert", vocab=30522, stream=stream,
                      label="LiT (siglip_lit_coco.py as written): frozen ViT-B/16 cls-token tower + trainable BERT-base "
                            "text tower, 16 tokens, batch 512, text-only backward")
  r["metric"] = "image-text pairs/sec, LiT locked-image step with the BERT-base text tower, batch 512 (BASELINE configs[4])"
  return r


def workload_rank_shape(dev, steps, n, stream="float32", tower_streams=None):
  """The pairs ONE rank of the headline owns at N = 4096 / n GPUs (single pass, full contexts, loss over the local
  pairs only: no peers on a single device).  tower_streams = 1: both towers on one stream (the trainer's default since
  round 6 is 2: text tower on a side stream beside the image tower; identical results) - the A/B partner of the
  default entry."""
  r = workload_siglip(dev, steps, IMAGE_CFG, TEXT_CFG, EMB, n=n, res=RES, seq=SEQ, micro=MICRO, stream=stream,
                      gflop_per_pair=MATMUL_GFLOP_PER_PAIR, tower_streams=tower_streams,
                      label=f"headline model, the {n} pairs one of {GLOBAL_BATCH // n} ranks owns (no collectives)"
                            + (", config.tower_streams = 1" if tower_streams == 1 else ""))
  r["metric"] = (f"image-text pairs/sec per GPU at {n} pairs per GPU (rank shape of the headline at N = {GLOBAL_BATCH // n})"
                 + (", both towers on ONE stream (config.tower_streams = 1)" if tower_streams == 1 else ""))
  return r


def workload_rank_shape_rccl(dev, steps, n):
  """workload_rank_shape with the step's collectives IN the loop: a one-rank RCCL group and dp.Comm(force=True), so
  that all_gather_into_tensor / reduce_scatter_tensor / the bucketed gradient all-reduce on the side stream / the
  scalar all-reduce are issued exactly as N > 1 issues them (the call path the driver's N = 1 box can execute)."""
  import torch.distributed as dist
  from big_vision_amd import dp
  own = not dist.is_initialized()
  if own:
    port = _free_port()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_MAX_NCHANNELS", str(dp.RESERVED_CUS))
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
  try:
    comm = dp.Comm(force=True)
    r = workload_siglip(dev, steps, IMAGE_CFG, TEXT_CFG, EMB, n=n, res=RES, seq=SEQ, micro=MICRO, comm=comm,
                        gflop_per_pair=MATMUL_GFLOP_PER_PAIR,
                        label=f"headline model, the {n} pairs one rank owns, collectives issued on a one-rank RCCL group")
    r["metric"] = (f"image-text pairs/sec per GPU at {n} pairs per GPU with the RCCL collectives of the step in the loop "
                   "(one-rank group)")
    r["rccl"] = rccl_info(comm, dev)
  finally:
    if own:
      dist.destroy_process_group()
  return r


def configs_object(dev, steps=3):
  """The `configs` object of the N = 1 line: every entry is measured here, after the headline, on the same device."""
  import gc
  out = {}
  # the short steps (36-170 ms) are timed over at least 8 of them: three steps of the 512-pair shape with the one-rank RCCL
  # group read 85.9 ms on one box and 92.9 on the next, ten steps 86.4 +- 0.1 on both (profiles/NOTES_r06.md)
  few, more = steps, max(steps, 8)
  for key, k, fn in (("c2", more, lambda: workload_c2(dev, more)), ("c4_rank", few, lambda: workload_c4(dev, few)),
                     ("c5b", more, lambda: workload_c5b(dev, more)), ("rank512", more, lambda: workload_rank_shape(dev, more, 512)),
                     ("rank1024", more, lambda: workload_rank_shape(dev, more, 1024)),
                     ("rank512_rccl", more, lambda: workload_rank_shape_rccl(dev, more, 512)),
                     ("rank512_one_stream", more, lambda: workload_rank_shape(dev, more, 512, tower_streams=1)),
                     ("rank1024_one_stream", more, lambda: workload_rank_shape(dev, more, 1024, tower_streams=1))):
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    try:
      r = fn()
      out[key] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                  "steps": k, "roofline_frac": r["roofline"]["frac"], "step_frac": r.get("step_frac"),
                  "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                  "final_loss": r["config"].get("final_loss")}
      if "rccl" in r:
        out[key]["rccl"] = r["rccl"]
    except Exception as e:   # the headline must not depend on the extra workloads
      out[key] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=4)
  ap.add_argument("--warmup", type=int, default=1)
  ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH)
  ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch HIP events")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-bf16-stream", action="store_true",
                  help="skip the second (bf16 residual stream) measurement that N = 1 appends as `bf16_stream`")
  ap.add_argument("--no-configs", action="store_true",
                  help="skip the `configs` object (the other BASELINE workloads and rank shapes, N = 1 only)")
  ap.add_argument("--no-live-pmc", action="store_true",
                  help="do not measure roofline.traffic with two rocprofv3 --pmc child passes (N = 1); the committed profile is used")
  ap.add_argument("--configs-steps", type=int, default=3)
  ap.add_argument("--cpu-sample", type=int, default=16)
  ap.add_argument("--microbatch", type=int, default=MICRO, help="pairs per micro-batch and rank")
  ap.add_argument("--residual-stream", default=RESIDUAL_STREAM, choices=("float32", "bfloat16"),
                  help="dtype of the activations between the encoder blocks (config.residual_stream)")
  args = ap.parse_args()

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    return spawn_ranks(args.gpus)   # plain `python bench.py --gpus N`: one process per GPU, started here

  # stdout carries exactly ONE line (rank 0's JSON).  Libraries write to fd 1 too (RCCL prints its
  # version banner there when a communicator is created): from here on fd 1 points at stderr and the
  # JSON line goes to the saved descriptor.
  sys.stdout.flush()
  json_fd = os.dup(1)
  os.dup2(2, 1)

  from big_vision_amd import _lib, dp
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip

  if not torch.cuda.is_available():
    raise RuntimeError("bench.py needs a GPU: the product path has no CPU fallback")
  # N > 1: rank 0 times the CPU oracle FIRST, before the process group exists (the peers wait in the rendezvous):
  # after the timed region the ranks leave at different times, and a rank that computes for half a minute while its
  # peers tear their communicators down is the one asymmetry this script can avoid.  N = 1: after the GPU work.
  cpu_early = None
  if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0 and not args.no_cpu_baseline:
    cpu_early = cpu_baseline(args.cpu_sample)
  comm = dp.init_from_env(overlap_channels=dp.RESERVED_CUS)   # gradient all-reduces overlap the backward's GEMMs
  world = comm.size
  assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
  # (BV_BENCH_SHARE_GPU=1: every rank on GPU 0 also under a launcher that numbers LOCAL_RANK itself - torchrun on a one-GPU box)
  local = 0 if os.environ.get("BV_BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
  torch.cuda.set_device(local)
  dev = torch.device("cuda", local)
  assert args.global_batch % world == 0
  n = args.global_batch // world

  total_steps = max(20_000, args.steps + args.warmup)
  image, text = synthetic_batch(n, dev, seed=1 + comm.rank)
  batch = {"image": image, "labels": text}

  def measure(stream, steps, warmup, roofline):
    """W untimed + K timed steps of the full training step on `stream`; everything it allocates is freed
    on return (the second, bf16-stream measurement at N = 1 needs the HBM back)."""
    model = two_towers.Model(image=IMAGE_CFG, text=TEXT_CFG, out_dim=(None, EMB),   # (the model caches its
                             temperature_init=10.0, bias_init=-10.0)                # executors: one per measurement)
    config = make_config(total_steps)
    config.microbatch = args.microbatch
    config.residual_stream = stream
    state, _ = siglip.make_train_state(model, config, (n, RES, RES, 3), (n, SEQ), rng=0, comm=comm,
                                       total_steps=total_steps, device=dev)
    update_fn = siglip.make_update_fn(model, config, comm=comm)
    obs = GemmObserver()
    meas = None
    for _ in range(warmup):
      state, meas = update_fn(state, None, batch)
    # Host cost of one step = wall time to ENQUEUE it on an idle GPU with an empty launch queue (the
    # average over the timed steps below also contains the time the host spends blocked on a full
    # queue while the GPU is the bottleneck, so it says nothing about host-boundness).  Untimed extra
    # step, outside the timed region.
    host_unblocked_ms = None
    if warmup > 0:
      comm.barrier()
      torch.cuda.synchronize()
      h0 = time.perf_counter()
      state, meas = update_fn(state, None, batch)
      host_unblocked_ms = 1e3 * (time.perf_counter() - h0)
    if roofline:
      _lib.observer = obs
    comm.barrier()
    torch.cuda.synchronize()
    obs.start() if roofline else None
    t0 = time.perf_counter()
    for _ in range(steps):
      state, meas = update_fn(state, None, batch)
    host_dt = time.perf_counter() - t0   # host enqueue time (diagnostic: host- vs GPU-bound)
    torch.cuda.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    obs.active = False
    _lib.observer = None
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
      torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(meas["training_loss"].item())
    siglip.check_finite(meas)
    # One EXTRA, untimed step with both towers on one stream and the same per-launch events: launches are serialised,
    # so the family's busy time there is exclusive by construction.  Reported as `roofline.one_stream` beside the
    # timed region's figure (on two streams a family launch that runs beside the other tower's LayerNorm / attention
    # kernels in its last, ragged round is charged that whole interval).
    obs1 = None
    if roofline and int(config.get("tower_streams", 2) or 1) == 2:
      try:
        torch.cuda.empty_cache()          # blocks cached by the side stream's pool are of no use to a one-stream step
        config.tower_streams = 1
        obs1 = GemmObserver()
        _lib.observer = obs1
        comm.barrier()
        torch.cuda.synchronize()
        obs1.start()
        t1 = time.perf_counter()
        state, meas1 = update_fn(state, None, batch)
        torch.cuda.synchronize()
        obs1.wall_s = time.perf_counter() - t1
        obs1.active = False
        del meas1
      except Exception as e:   # the headline must not depend on the extra step
        obs1 = None
        print(f"one-stream roofline step failed: {type(e).__name__}: {e}", file=sys.stderr)
      finally:
        _lib.observer = None
        config.tower_streams = 2
    res = dict(dt=dt, host_dt=host_dt, host_unblocked_ms=host_unblocked_ms, loss=loss, obs=obs, obs_one_stream=obs1,
               keep_n=update_fn.state_cache["keep_n"], light=update_fn.state_cache["light"],
               peak=torch.cuda.max_memory_allocated(dev), tower_streams=int(config.get("tower_streams", 2) or 1))
    del state, update_fn, meas, model
    return res

  r = measure(args.residual_stream, args.steps, args.warmup, not args.no_roofline)
  dt, host_dt, host_unblocked_ms, loss, obs = r["dt"], r["host_dt"], r["host_unblocked_ms"], r["loss"], r["obs"]
  # N = 1 only: the opt-in bf16 residual stream on the same workload, as a separate object of the line
  bf16_line = None
  if world == 1 and args.residual_stream == "float32" and not args.no_bf16_stream:
    try:
      import gc
      gc.collect()
      torch.cuda.empty_cache()
      torch.cuda.reset_peak_memory_stats(dev)
      k2 = max(1, min(args.steps, 2))
      r2 = measure("bfloat16", k2, 1, False)
      bf16_line = {"value": args.global_batch * k2 / r2["dt"], "unit": "pairs/s", "ms_per_step": 1e3 * r2["dt"] / k2,
                   "steps": k2, "warmup": 1, "final_loss": r2["loss"], "peak_hbm_gb": round(r2["peak"] / 1e9, 1),
                   "contexts": CTX_KIND[r2["light"]] if n > args.microbatch else "full",
                   "note": "config.residual_stream='bfloat16': narrower than the reference's fp32 residual stream "
                           "(models/vit.py:92-110); reported beside `value`, never as `value`"}
      bf16_line.update(bf16_stream_parity())
    except Exception as e:   # the headline must not depend on the optional second measurement
      bf16_line = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}

  # (a collective: every rank takes part, then the others are done)
  rccl = rccl_info(comm, dev) if (world > 1 or os.environ.get("BV_DP_FORCE_COLLECTIVES") == "1") else None
  if comm.rank != 0:
    return
  roof = None
  if not args.no_roofline:
    roof = roofline_object(obs, dt)
    if r.get("obs_one_stream") is not None:
      one = roofline_object(r["obs_one_stream"], r["obs_one_stream"].wall_s)
      roof["one_stream"] = {"frac": one["frac"], "achieved": one["achieved"], "avg_launch_us": one["avg_launch_us"],
                            "launches": one["launches"], "ms_of_this_step": 1e3 * r["obs_one_stream"].wall_s,
                            "what": "the same launches serialised: one extra, untimed step with config.tower_streams = 1 "
                                    "after the timed region (per-launch HIP events are exclusive there)"}
    traffic, traffic_src = pmc_traffic(world, n, args.microbatch)
    traffic_live, traffic_detail = False, None
    if world == 1 and not args.no_live_pmc and args.global_batch == GLOBAL_BATCH:
      # counters cannot be read in-process: two child rocprofv3 passes of this command, now, on this GPU (the HBM
      # of this process is released first: the child needs all of it)
      import gc
      batch.clear()
      image = text = None   # the synthetic batch (2.5 GB): every later object builds its own inputs
      gc.collect()
      torch.cuda.empty_cache()
      try:
        live, traffic_detail = live_pmc_traffic(args.microbatch)
      except Exception as e:   # the headline must not depend on the profiler
        live, traffic_detail = None, {"error": f"{type(e).__name__}: {e}"[:300]}
      if live is not None:
        traffic, traffic_src, traffic_live = live, "rocprofv3 --pmc child passes of this run", True
    roof.update(traffic=traffic, traffic_source=traffic_src, traffic_measured_in_this_run=traffic_live,
                traffic_detail=traffic_detail)
  configs = None
  if world == 1 and not args.no_configs and args.global_batch == GLOBAL_BATCH:
    configs = configs_object(dev, args.configs_steps)
  # rank 0 of ANY world size carries the CPU oracle's rate (N > 1: timed above, before the rendezvous)
  cpu = cpu_early if cpu_early is not None else (None if args.no_cpu_baseline else cpu_baseline(args.cpu_sample))
  line = assemble_line(args, world, n, r, roof, bf16_line, rccl, configs, cpu)
  sys.stdout.flush()
  os.write(json_fd, (json.dumps(line) + "\n").encode())


def assemble_line(args, world, n, r, roof, bf16_line, rccl, configs, cpu):
  """The ONE JSON line of rank 0 from the measured pieces (pure: tests/test_bench_line_cpu.py feeds it a faked
  N = 2 measurement).  r: measure()'s result dict; roof: roofline_object() (+ traffic fields) or None."""
  dt = r["dt"]
  value = args.global_batch * args.steps / dt
  line = {
      "metric": "image-text pairs/sec training step, ViT-B/16 SigLIP bs4096",
      "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
      "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
      "config": {"workload": "SigLIP ViT-B/16@224 (MAP) + text-B 12L/64tok/vocab32k, sigmoid loss, "
                             "Adam+clip+wd+cosine, random-init weights (BASELINE configs[2])",
                 "global_batch": args.global_batch, "per_gpu_batch": n, "microbatch": args.microbatch,
                 "residual_stream": args.residual_stream,
                 "tower_streams": r.get("tower_streams"),
                 **({"shared_gpu": "all ranks on ONE GPU over gloo: a functional run, not a measurement"}
                    if os.environ.get("BV_BENCH_SHARE_GPU") == "1" else {}),
                 "recompute": (f"{max(0, n // args.microbatch - r['keep_n'])} of "
                               f"{n // args.microbatch} micro-batches re-run their forward in pass 2 "
                               f"(the others keep {CTX_KIND[r['light']]} "
                               "activation contexts in HBM)")
                              if n > args.microbatch else "none",
                 "parallelism": f"dp{world}", "final_loss": r["loss"],
                 "img_per_sec_per_core": value / world,   # the reference's own rate figure (utils.py:506, Chrono.tick)
                 "peak_hbm_gb": round(r["peak"] / 1e9, 1),
                 "host_enqueue_ms_idle_gpu": r["host_unblocked_ms"],
                 "host_wall_ms_per_step_incl_queue_backpressure": 1e3 * r["host_dt"] / args.steps},
  }
  if roof is not None:
    roof = dict(roof)
    roof["traffic_unit"] = "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)"
    # whole step, PER GPU: algorithmic matmul FLOPs of the workload (forward + backward of the kept contexts; the
    # recompute FLOPs of re-run micro-batches are NOT counted) / wall time / (world x one GPU's peak)
    roof["step_frac"] = (MATMUL_GFLOP_PER_PAIR * 1e9 * args.global_batch * args.steps / dt / 1e12
                         / (BF16_DENSE_PEAK_TFLOPS * world))
    roof["step_frac_is"] = "per GPU (job FLOP/s / (n_gpus x 2.5 PF))"
    if world > 1:
      roof["measured_on"] = "rank 0 (every rank runs the same launches on its 1/N of the batch)"
    line["roofline"] = roof
  if bf16_line is not None:
    line["bf16_stream"] = bf16_line
  if rccl is not None:
    line["rccl"] = rccl
  if configs is not None:
    line["configs"] = configs
  if cpu is not None:
    line["cpu_baseline"] = cpu
  return line


if __name__ == "__main__":
  try:
    main()
  finally:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
      torch.distributed.destroy_process_group()
