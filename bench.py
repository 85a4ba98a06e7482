#!/usr/bin/env python3
"""Headline benchmark: H2O scored-decode hot path, Qwen2.5-7B shapes, 128k-context rows.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): sparse_method=h2o, decode_budget=4096, interval=128,
28 layers, 28 q heads / 4 kv heads, head_dim 128, bf16.  Every GPU hosts B sequences whose
131072-token prompts have already been reduced by chunked prefill to 4096-token physical rows
(per layer) scattered over a randomly permuted paged KV pool.  One "step" = one decode step of
the hot path for the whole batch: slot allocation, then per layer {store_kvcache, scored
split-KV decode stage 1, stage 2 merge, score normalise + accumulate}, then the eviction
trigger; every 128th step runs the burst (exact H2O selection + slot-table compaction on all
28 layers).  K steps are timed between barriers; `value` = tokens of all GPUs / max-over-ranks
time.  The model's dense layers are outside this build (hot path only).

One JSON line is printed by rank 0.  `roofline` times every stage-1 launch of the timed
region with HIP events on the launch stream; `cpu_baseline` times the numpy oracle on a
bounded sample of the same workload on the host (rank 0, N=1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK = 8.0e12   # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="sequences per GPU (a 288 GB MI355X holds far more 4224-token "
                    "rows than this; 256 rows = one unsplit stage-1 workgroup per CU; "
                    "64 = the round-1 default, see profiles/ for 64 / 128 / 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of hipGraph replay")
    ap.add_argument("--event-steps", type=int, default=8, help="eager steps timed per launch for the roofline leg")
    ap.add_argument("--max-model-len", type=int, default=131072, help="row stride of the slot table / score tensor")
    ap.add_argument("--no-paths", action="store_true", help="skip the other configurations' decode-step timings")
    ap.add_argument("--path-steps", type=int, default=24)
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (tools/e2e_decoder.py in child processes)")
    ap.add_argument("--e2e", action="store_true", help="run ONLY the end-to-end leg and print its lines (one per batch / mode)")
    ap.add_argument("--tp", type=int, default=0, help="tensor-parallel degree of the end-to-end leg (0: 1 on one GPU, else "
                    "the largest of 2 / 4 that divides the GPU count: Qwen2.5-7B has 4 KV heads)")
    ap.add_argument("--e2e-batches", default="", help="comma list; default 1,64,256 on one GPU, 64 under tensor parallelism")
    ap.add_argument("--stub-driver", action="store_true",
                    help="tests only: a stand-in decode driver that never touches a GPU (gloo ranks, a step = a fixed sleep of "
                         "rank + 1 ms) so that the launch / rendezvous / aggregation plumbing of --gpus N runs on CPU")
    return ap.parse_args()


class _StubDriver:
    """`--stub-driver`: the interface of SparseDecodeDriver that main() uses, with no device behind it."""

    def __init__(self, batch: int, rank: int, interval: int):
        from types import SimpleNamespace
        self.B, self.rank, self.interval, self.steps_done = int(batch), int(rank), int(interval), 0
        self.cache_manager = SimpleNamespace(_h2o_counters={"decode_eviction_bursts": 0}, permute_free_slots=lambda seed: None)
        self.config = SimpleNamespace(decode_cuda_graph=False)

    def admit_resident_rows(self, *a, **kw):
        return None

    def random_step_inputs(self, seed: int = 0):
        return None, None, None

    def enable_decode_graph(self):
        self.config.decode_cuda_graph = True

    def step(self, q, k, v, outputs=None, *, after_layers=None):
        time.sleep(1e-3 * (self.rank + 1))
        self.steps_done += 1
        if self.steps_done % self.interval == 0:
            self.cache_manager._h2o_counters["decode_eviction_bursts"] += self.B


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _cpu_worker(job):
    """One host core: the numpy oracle on ONE sequence x 28 layers x `steps` decode steps at row length 4160 (the mean of
    the 4096..4224 cycle), then one burst selection over the sequence's 28 rows.  -> (seconds of steps, seconds of burst)"""
    seed, steps = job
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:      # pragma: no cover
        pass
    from oracle import bf16_round
    from oracle import decode_attention as oda
    from oracle import h2o as oh
    rng = np.random.default_rng(seed)
    B, Hq, Hkv, D, Lrow, layers = 1, 28, 4, 128, 4160, 28
    slots = B * 4224 + 64
    k = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.3).astype(np.float32))
    v = bf16_round((rng.standard_normal((slots, Hkv, D)) * 0.3).astype(np.float32))
    q = bf16_round((rng.standard_normal((B, Hq, D)) * 0.3).astype(np.float32))
    req = rng.permutation(slots)[: B * 4224].reshape(B, 4224).astype(np.int32)
    rows = np.arange(B, dtype=np.int32)
    lens = np.full(B, Lrow, np.int32)
    cum = rng.random((B, Lrow - 1)).astype(np.float32)
    t0 = time.perf_counter()
    for _ in range(steps):
        for _l in range(layers):
            raw = np.full((B, Lrow), -1e20, np.float32)
            mid, lse = oda.flash_decode_stage1(q, k, v, req, rows, lens, Lrow, 256, attn_score=raw)
            oda.flash_decode_stage2(mid, lse, lens, 256)
            norm = oda.h2o_normalize_decode_scores(raw, D)
            oh.update_decode_scores(cum, norm, Lrow)
    t_steps = time.perf_counter() - t0
    s = rng.random((layers * B, 4224)).astype(np.float32)
    t0 = time.perf_counter()
    oh.select_h2o_indices_batch(s, budget=4096, recent_ratio=0.5)
    return t_steps, time.perf_counter() - t0


def _usable_cores() -> int:
    """Cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota (a container can see 256
    CPUs in os.cpu_count() and own far fewer)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:      # pragma: no cover
        n = int(os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], parts[1]
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().split()[0]
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / float(period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(seed: int = 20260625, steps: int = 4, workers: int | None = None):
    """The oracle (numpy restatement of the reference) on ALL host cores: one worker process per core, one sequence per
    worker (the path shards by sequence on the CPU exactly as on the GPUs), 28 layers x `steps` decode steps each, plus
    one burst selection per sequence amortised over the 128-step interval.  Must run before this process touches the
    GPU (the workers are forked)."""
    import multiprocessing as mp
    cores = int(workers) if workers else _usable_cores()
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(seed + i, steps) for i in range(cores)], chunksize=1)
    wall = time.perf_counter() - t0
    # every worker decodes its sequence concurrently: a step of the whole batch takes as long as the slowest worker's
    per_step = max(r[0] for r in res) / steps + max(r[1] for r in res) / 128.0
    return {
        "value": cores / per_step, "unit": "tokens/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
        "cpus_visible": int(os.cpu_count() or 0),
        "sample": f"numpy oracle, {cores} worker processes (one per usable host core) x 1 sequence x 28 layers x {steps} decode "
                  f"steps at row length 4160 (+1 burst selection over each sequence's 28 rows / 128 steps); slowest worker "
                  f"{max(r[0] for r in res):.1f} s of steps, {wall:.1f} s wall incl. process start; single worker alone: "
                  f"{steps / min(r[0] for r in res):.2f} tokens/s",
    }


def _paths_in_child(names, steps: int, timeout_s: int = 240) -> list[dict]:
    """tools/pathbench.py in a CHILD process (one per call, all configurations): whatever a side leg does - a Python
    exception, an abort inside a library, a hang - the headline line of this process is already computed and still gets
    printed.  Each configuration comes back as one JSON line; a configuration the child did not report is an error entry."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "pathbench.py"), "--graph", "--configs", ",".join(names),
           "--steps", str(int(steps)), "--warmup", "4"]
    got, note = {}, None

    def take(stdout):
        for line in (stdout or "").splitlines():
            if line.startswith("{"):
                try:
                    d = json.loads(line)
                    got[d.get("config")] = d
                except ValueError:
                    pass

    try:
        # well below the 400 s / 600 s `timeout` wrappers of tools/refresh_profiles.sh: a hung side leg must not take the
        # headline line (printed after this returns) with it
        r = subprocess.run(cmd, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
        take(r.stdout)
        if r.returncode != 0:
            tail = [l for l in r.stderr.strip().splitlines() if l and "amdgpu.ids" not in l][-1:] or [""]
            note = f"pathbench child exited with {r.returncode}: {tail[0][:200]}"
    except subprocess.TimeoutExpired as e:
        out = e.stdout
        take(out.decode(errors="replace") if isinstance(out, bytes) else out)      # keep the configurations that finished
        note = f"pathbench child timed out after {timeout_s} s"
    except OSError as e:
        note = f"pathbench child failed to start: {e}"
    return [got.get(n) or {"config": n, "error": note or "not reported by the pathbench child"} for n in names]


def _traffic_in_child(batch: int, timeout_s: int = 150) -> dict:
    """HBM bytes of one stage-1 launch from the PMC counters, measured in this run: `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` as two separate passes (counters only, the program itself after `--`) over tools/kbench.py at the
    headline batch and row length 4224, with the micro-architecture guide's gfx950 correction (FETCH_SIZE counts 128-byte
    requests as 64: read bytes = 2 x FETCH_SIZE x 1024).  -> per-launch figures, or {"error": ...} (then the committed
    profiles/stage1_traffic.json stays the source)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return {"error": "rocprofv3 not on PATH"}
    L = 4224
    per_launch = {}
    tmp = tempfile.mkdtemp(prefix="svk_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, ctr)
            cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", out_dir, "--", sys.executable,
                   os.path.join(ROOT, "tools", "kbench.py"), "--batches", str(batch), "--block-seqs", str(L), "--modes", "2", "--iters", "3"]
            try:
                subprocess.run(cmd, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            except (subprocess.TimeoutExpired, OSError) as e:
                return {"error": f"{ctr} pass: {type(e).__name__}"}
            vals = []
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    vals += [float(r["Counter_Value"]) for r in csv.DictReader(fh)
                             if r["Counter_Name"] == ctr and "decode_stage1_kernel" in r["Kernel_Name"]]
            if not vals:
                return {"error": f"{ctr} pass: no stage-1 rows in the counter output"}
            per_launch[ctr] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = 2.0 * per_launch["FETCH_SIZE"][0] * 1024.0, per_launch["WRITE_SIZE"][0] * 1024.0
    return {"hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "row_len": L,
            "batch": batch, "launches": per_launch["FETCH_SIZE"][1], "algorithmic_bytes_per_launch": batch * L * (2 * 4 * 128 * 2 + 4 + 4),
            "measured_in": "children of this bench.py run"}


def _e2e_in_child(args, world: int, rank: int, local_rank: int, timeout_s: int = 200) -> list[dict]:
    """The end-to-end leg (tools/e2e_decoder.py) in a CHILD process per rank, after this process has released its GPU
    memory: a random-weight Qwen2.5-7B-shaped decoder (torch dense layers) around this build's attention path.  One GPU:
    tp=1 at B in {1, 64, 256}.  N GPUs: tensor-parallel groups (RCCL all-reduce after o_proj / down_proj) x replicas, the
    children rendezvous on their own port.  Bounded by `timeout_s`; whatever happens in there - an exception, a hang in a
    collective - the headline measurement of this process is already taken and is printed afterwards.  -> parsed lines."""
    import subprocess
    tp = int(args.tp) or (1 if world == 1 else (4 if world % 4 == 0 else 2))
    batches = args.e2e_batches or ("1,64,256" if tp == 1 else "64")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_decoder.py"), "--tp", str(tp), "--batches", batches,
           "--steps", "24", "--warmup", "3", "--modes", "graph" if world == 1 else "eager,graph"]
    env = dict(os.environ)
    if world > 1:
        env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 23))
    else:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    lines, note = [], None

    def take(stdout):
        for line in (stdout or "").splitlines():
            if line.startswith("{"):
                try:
                    lines.append(json.loads(line))
                except ValueError:
                    pass

    try:
        r = subprocess.run(cmd, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True,
                           env=env)
        take(r.stdout)
        if r.returncode != 0:
            tail = [l for l in r.stderr.strip().splitlines() if l and "amdgpu.ids" not in l][-1:] or [""]
            note = f"e2e child exited with {r.returncode}: {tail[0][:200]}"
    except subprocess.TimeoutExpired as e:
        out = e.stdout
        take(out.decode(errors="replace") if isinstance(out, bytes) else out)
        note = f"e2e child timed out after {timeout_s} s"
    except OSError as e:
        note = f"e2e child failed to start: {e}"
    if note:
        lines.append({"e2e": True, "error": note})
    return lines


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks ourselves (child processes, before anything here touches a GPU)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:]))
    if args.gpus != world:
        print(f"[bench] --gpus {args.gpus} does not match WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.e2e:
        for line in _e2e_in_child(args, world, rank, local_rank, timeout_s=900):
            if rank == 0:
                print(json.dumps(line), flush=True)
        return
    stub = bool(args.stub_driver)
    if stub:
        args.no_cpu_baseline = args.no_kernel_events = args.no_paths = True
    if use_dist:
        import torch.distributed as dist
        if stub:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    n_gpus = world if use_dist else 1
    device = "cpu" if stub else f"cuda:{local_rank}"
    # the CPU leg forks one worker per host core, so it runs before this process creates a GPU context
    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:           # the CPU leg is a reported baseline: it must not take the measurement with it
            cpu = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}

    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
    import sparse_vllm_amd.layers.attention as attn_mod

    B = args.batch
    budget, interval = 4096, 128
    conf = Config.from_kwargs(
        sparse_method="h2o", num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4, head_dim=128,
        max_model_len=int(args.max_model_len), max_num_seqs_in_gpu=B, num_kvcache_slots=B * (budget + interval) + 4096,
        h2o_decode_budget=budget, h2o_decode_eviction_interval=interval, h2o_prefill_budget=8192,
        engine_prefill_chunk_size=8192, device=device)
    drv = _StubDriver(B, rank, interval) if stub else SparseDecodeDriver(conf)
    cm = drv.cache_manager
    cm.permute_free_slots(20260625 + rank)
    drv.admit_resident_rows(B, budget, logical_len=131072, seed=20260625 + rank, device_rng=True)
    q, k, v = drv.random_step_inputs(seed=7 + rank)

    # ---- stage-1 launch timing with HIP events on the launch stream (torch's current stream).
    # Eager mode: the arguments of every stage-1 launch of a step are captured, and right after the
    # step's layer loop - BEFORE post_forward, so before a burst can compact the slot table or change
    # the lengths - the same 28 launches are re-issued back to back between ONE pair of events: same
    # K/V rows, slot table, lengths and queries as the launches of the step itself.  (Bracketing each
    # launch separately measures the event markers' own system-scope cache flushes and the Python launch
    # latency, 2x the kernel time on this box.  The re-issue is idempotent for the partials and the
    # fused store; the raw-score buffer already holds this layer's raw scores, the max-combine's
    # traffic is the same.)
    events = []          # (total_ms, n_launches, row_len)
    record = {"on": False, "calls": []}
    orig = attn_mod.flash_decode_stage1_with_score

    def capturing_stage1(*a, **kw):
        if record["on"]:
            record["calls"].append((a, dict(kw)))
        return orig(*a, **kw)

    def time_captured_launches():
        calls, record["calls"] = record["calls"], []
        if not calls:
            return
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for a, kw in calls:
            orig(*a, **kw)
        e1.record()
        torch.cuda.synchronize()
        events.append((e0.elapsed_time(e1), len(calls), int(drv.row_len()[0])))

    if not args.no_kernel_events:
        attn_mod.flash_decode_stage1_with_score = capturing_stage1

    def barrier():
        if not stub:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    if not args.no_graph:
        drv.enable_decode_graph()
    # priming (part of setup, untimed): one full eviction interval, so that the first burst's one-time costs (code
    # object load of the selection / compaction kernels, first-use allocations: ~45 ms) are not billed to the K timed
    # steps.  The timed region still contains its bursts (one per 128 steps per sequence) at steady-state cost.
    for _ in range(int(conf.h2o_decode_eviction_interval) + 2):
        drv.step(q, k, v)
    warmup = max(args.warmup, 2 if not args.no_graph else 0)   # graph capture happens in the first steps
    for _ in range(warmup):
        drv.step(q, k, v)
    barrier()
    bursts0 = int(cm._h2o_counters["decode_eviction_bursts"])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.step(q, k, v)
    barrier()
    elapsed = time.perf_counter() - t0
    bursts_in_window = (int(cm._h2o_counters["decode_eviction_bursts"]) - bursts0) // B
    # ---- the burst on its own: run on to the next eviction step, time the plain steps before it and the step that carries
    # the burst (select + compact over all 28 layers x B rows); every sequence pays it once per `interval` steps
    burst = None
    if not args.no_graph and not stub:
        plain = []
        for _ in range(2 * interval + 2):
            n0 = int(cm._h2o_counters["decode_eviction_bursts"])
            torch.cuda.synchronize()
            ts = time.perf_counter()
            drv.step(q, k, v)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - ts) * 1e3
            if int(cm._h2o_counters["decode_eviction_bursts"]) > n0:
                base = float(np.median(plain[-16:])) if plain else float("nan")
                burst = {"ms": dt - base, "step_ms_with_burst": dt, "plain_step_ms": base, "per_steps": interval}
                break
            plain.append(dt)
    if not args.no_kernel_events:
        # roofline leg: the same workload continues for a few eagerly launched steps
        drv.config.decode_cuda_graph = False
        def after_layers():
            record["on"] = False
            time_captured_launches()

        for _ in range(args.event_steps):
            record["on"] = True
            drv.step(q, k, v, after_layers=after_layers)
    from sparse_vllm_amd.replicas import aggregate_throughput, gather_per_rank
    per_rank_s = gather_per_rank(elapsed, device=device)          # every replica's own clock (N = 1: one entry)
    tokens, elapsed = aggregate_throughput(B * args.steps, elapsed, device=device)
    out = {
        "metric": "decode tokens/s at 128k ctx, H2O budget=4k, Qwen2.5-7B (sparse attention hot path)",
        "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {
            "workload": "Qwen2.5-7B h2o 128k ctx decode_budget=4096 interval=128 (BASELINE.json configs[2]); "
                        "hot path only: alloc + 28 x {store_kvcache, scored stage1, stage2 + score update} + burst "
                        "eviction every 128 steps; dense model layers not included",
            "seqs_per_gpu": B, "global_batch": n_gpus * B, "launch": "eager" if args.no_graph else "hipGraph replay", "resident_row_len": "4096..4224", "layers": 28,
            "heads": "28q/4kv x 128", "parallelism": f"replicas x{n_gpus} (sequence-sharded, no collective)",
            "max_model_len": int(args.max_model_len), "logical_context": 131072,
            "bursts_in_timed_window": int(bursts_in_window),
        },
        # replicas: what every rank measured on its own clock (value uses the slowest one)
        "ranks_seen": len(per_rank_s),
        "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank_s],
        "per_gpu_tokens_per_s": {"min": B * args.steps / max(per_rank_s), "max": B * args.steps / min(per_rank_s)},
    }
    if burst is not None:
        # what the timed window would read with exactly one burst per `interval` steps, whatever K the driver chose
        plain_ms = (elapsed * 1e3 - bursts_in_window * burst["ms"]) / args.steps
        burst["value_amortised"] = n_gpus * B / ((plain_ms + burst["ms"] / interval) * 1e-3)
        out["burst"] = burst
    if events:
        ms_total = sum(e[0] for e in events)
        n_launch = sum(e[1] for e in events)
        bytes_total = float(sum(e[1] * B * e[2] * (2 * 4 * 128 * 2 + 4 + 4) for e in events))
        achieved = bytes_total / (ms_total * 1e-3)
        traffic = traffic_meas = None
        pmc = os.path.join(ROOT, "profiles", "stage1_traffic.json")
        mean_len = float(sum(e[1] * e[2] for e in events)) / n_launch
        if os.path.exists(pmc):
            try:
                tr = json.load(open(pmc))
                # measured per launch at one batch size and row length (tools/make_traffic_json.py): reported for that batch
                # only, and SCALED to the mean row length of the launches timed here (traffic is proportional to the rows'
                # tokens: K + V + slot id + score per token), so that it compares like for like with
                # algorithmic_bytes_per_launch; the measured pair stays beside it
                if int(tr.get("batch", 64)) == B:
                    traffic_meas = {"hbm_bytes_per_launch": tr.get("hbm_bytes_per_launch"), "row_len": tr.get("row_len"),
                                    "algorithmic_bytes_per_launch": tr.get("algorithmic_bytes_per_launch")}
                    traffic = float(tr["hbm_bytes_per_launch"]) * mean_len / float(tr["row_len"])
            except Exception:
                traffic = traffic_meas = None
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_measured": traffic_meas, "mean_row_len": mean_len,
            "traffic_source": "profiles/stage1_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/kbench.py "
                              "at this batch (builder-run, not re-measured in this process), scaled by mean_row_len / its row_len",
            "kernel": "decode_stage1_kernel_v3<128,7,HEADMAX,nt,off32> (scored GQA split-KV decode)",
            "launches_timed": n_launch, "avg_launch_us": ms_total * 1e3 / n_launch,
            "timing": (f"{args.event_steps} eagerly launched steps continue the timed region; after each step's layer "
                       "loop and before its post_forward, its 28 stage-1 launches are re-issued back to back on the same "
                       "data and state between one pair of HIP events on the launch stream"),
            "algorithmic_bytes_per_launch": bytes_total / n_launch,
        }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0 and n_gpus == 1 and not args.no_paths:
        # the other configurations of BASELINE.json (and the B=64 H2O point of SURVEY 8(d)) as decode-step timings of the
        # same build, a few seconds each: hipGraph replay, synthetic resident state (tools/pathbench.py)
        attn_mod.flash_decode_stage1_with_score = orig
        del drv, cm, q, k, v
        record["calls"].clear()
        torch.cuda.empty_cache()
        names = ("h2o_b64", "h2o_b8", "h2o_b1", "streamingllm", "streamingllm_b1", "quest", "quest_b8", "quest_b1", "deltakv",
                 "deltakv_b4")
        out["paths"] = _paths_in_child(names, args.path_steps)
        # SURVEY 8(d) quotes the H2O metric at B in {1, 8, 32, 64}: the B=64 point inside `config`, where the driver's
        # parser keeps it (same workload and build as the headline, tools/pathbench.py `h2o_b64`)
        b64 = next((p for p in out["paths"] if p.get("config") == "h2o_b64" and "error" not in p), None)
        if b64 is not None:
            # flat scalar keys: the driver's parser keeps scalars of `config` and drops nested objects
            out["config"].update({
                "b64_ms_per_step": b64.get("ms_per_step"), "b64_tokens_per_s": b64.get("tokens_per_s"),
                "b64_step_frac_of_hbm_peak": b64.get("roofline_frac"), "b64_stage1_kernel_us": b64.get("kernel_us"),
                "b64_stage1_frac_of_hbm_peak": b64.get("kernel_frac")})
        if "roofline" in out:
            # HBM traffic of the stage-1 launch measured NOW, in children of this process (two counter passes)
            meas = _traffic_in_child(B)
            if meas.get("hbm_bytes_per_launch"):
                rf = out["roofline"]
                rf["traffic"] = float(meas["hbm_bytes_per_launch"]) * rf["mean_row_len"] / float(meas["row_len"])
                rf["traffic_measured"] = meas
                rf["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/kbench.py at "
                                        "this batch, run as children of this bench process; read bytes = 2 x FETCH_SIZE x 1024 "
                                        "(the guide's gfx950 correction), scaled by mean_row_len / row_len")
            else:
                out["roofline"]["traffic_in_process"] = meas
    if not stub and not args.no_e2e:
        # end-to-end decode tokens/s as BASELINE.json words the metric, beside the unchanged headline: dense
        # layers are plain torch GEMMs (outside this build's scope), the attention is this build's path
        drv = cm = q = k = v = None          # release this process's KV pool before the children start
        record["calls"].clear()
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if use_dist:
            dist.barrier()
        e2e = _e2e_in_child(args, n_gpus, rank, local_rank)
        if use_dist:
            dist.barrier()
        out["e2e"] = {"note": "random-weight Qwen2.5-7B-shaped decoder, torch F.linear / rms_norm / RoPE around this build's "
                              "attention path (tools/e2e_decoder.py); bound_ms = (14.1 GB + B x 0.243 GB) / tp / 8 TB/s "
                              "(BASELINE.md section 2); the dense layers are not part of this build", "runs": e2e}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
