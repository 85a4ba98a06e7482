#!/usr/bin/env python3
"""Decode-path timings of the non-headline configurations (development / profiling tool, GPU only).

    python tools/pathbench.py [--configs streamingllm,quest,deltakv,deltakv_raw,vanilla] [--steps 32] [--graph]

BASELINE.json "other configs" (SURVEY section 8(d)), Qwen2.5-7B shapes, synthetic resident state:
  streamingllm  32k prompt reduced to sink 64 + recent 512, rows 576..1151, B=64
  quest         128k context, page 16, token budget 4672 (skip_layers 2 -> dense), B=4
  deltakv       256k context, sink 8 / recent 128 / keep 2048, K=4, latent 256 int4, 6 KIVI-int4 full layers, B=1
  deltakv_raw   same with raw bf16 full layers at 64k context
  vanilla       8k context dense decode, B=16
  h2o_b<N>      the headline configuration at N sequences; quest_b<N> / streamingllm_b<N> / deltakv_b4: the other batch
                points of SURVEY 8(d)
One JSON line per configuration: ms per decode step of the sparse path (all layers, no dense model).
"""
import argparse
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_vllm_amd.config import Config
if os.environ.get("SVK_AB_LIB"):          # developer A/B: another build of the library (e.g. sparse_vllm_amd/libsvk_ab.so)
    from sparse_vllm_amd import _lib as _svk_lib
    _svk_lib.LIB_PATH = os.path.abspath(os.environ["SVK_AB_LIB"])
from sparse_vllm_amd.engine.decode_driver import capture_without_gc
from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver

QWEN = dict(num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4, head_dim=128)


def build(name: str):
    if name.startswith("streamingllm"):
        B = int(name.split("_b")[1]) if "_b" in name else 64
        cfg = Config.from_kwargs(sparse_method="streamingllm", sink_keep_tokens=64, recent_keep_tokens=512,
                                 max_model_len=2048, max_num_seqs_in_gpu=B, num_kvcache_slots=B * 1160 + 64, **QWEN)
        drv = SparseDecodeDriver(cfg)
        drv.cache_manager.permute_free_slots(1)
        drv.admit_resident_rows(B, 576, logical_len=32768, seed=0, device_rng=True)
        return drv, dict(batch=B, context=32768, resident=576)
    if name.startswith("quest"):
        B, ctx = (int(name.split("_b")[1]) if "_b" in name else 4), 131072
        cfg = Config.from_kwargs(sparse_method="quest", sink_keep_tokens=64, decode_keep_tokens=4096, recent_keep_tokens=512,
                                 max_model_len=ctx + 256, max_num_seqs_in_gpu=B, num_kvcache_slots=B * (ctx + 256), **QWEN)
        drv = SparseDecodeDriver(cfg)
        drv.cache_manager.permute_free_pages(1)
        drv.admit_resident_rows(B, ctx, seed=0, device_rng=True)
        return drv, dict(batch=B, context=ctx, token_budget=cfg.quest_token_budget)
    if name in ("deltakv", "deltakv_raw") or name.startswith("deltakv_b"):
        kivi = name != "deltakv_raw"
        B, ctx = (int(name.split("_b")[1]) if name.startswith("deltakv_b") else 1), 8 + 128 * (2048 if kivi else 512)     # tail == recent: room for `recent` decode steps
        cfg = Config.from_kwargs(sparse_method="deltakv", full_attention_layers="0,1,2,8,18,27" if kivi else "0,1,2,8,18",
                                 sink_keep_tokens=8, recent_keep_tokens=128, decode_keep_tokens=2048, deltakv_neighbor_count=4,
                                 deltakv_latent_dim=256, deltakv_latent_quant_bits=4, deltakv_latent_quant_group_size=32,
                                 deltakv_center_ratio=0.03, allow_missing_deltakv_path=True, compressor_intermediate_size=2048,
                                 full_layer_kv_quant_bits=4 if kivi else 0, max_model_len=ctx + 512, max_num_seqs_in_gpu=B,
                                 **QWEN)
        drv = SparseDecodeDriver(cfg)
        drv.cache_manager.permute_free_slots(1)
        drv.admit_compressed_rows(B, ctx, seed=0)
        return drv, dict(batch=B, context=ctx, full_layers=len(drv.cache_manager.full_layer_ids), kivi=kivi)
    if name == "vanilla":
        B, ctx = 16, 8192
        cfg = Config.from_kwargs(sparse_method="", max_model_len=ctx + 256, max_num_seqs_in_gpu=B,
                                 num_kvcache_slots=B * (ctx + 256), **QWEN)
        drv = SparseDecodeDriver(cfg)
        drv.cache_manager.permute_free_slots(1)
        drv.admit_resident_rows(B, ctx, seed=0, device_rng=True)
        return drv, dict(batch=B, context=ctx)
    raise SystemExit(f"unknown config {name}")


def algorithmic_bytes_per_step(name: str, info: dict, mean_row_len: float | None = None) -> float:
    """HBM bytes one decode step of the configuration has to move (SURVEY.md section 8(d) per-unit figures x units):
    K + V rows of the attended tokens (2 KiB per token-layer) + slot ids (+ the fp32 score for H2O), Quest's page
    metadata scan (128 B per context token and sparse layer), DeltaKV's father gathers / latents / scratch writes and the
    712 B per token of a KIVI-int4 full layer."""
    B, L = info["batch"], 28
    if name.startswith("h2o"):
        return B * L * float(mean_row_len) * 2056
    if name.startswith("streamingllm"):
        return B * L * float(mean_row_len) * 2052
    if name == "vanilla":
        return B * L * float(info["context"]) * 2052
    if name.startswith("quest"):
        ctx, budget = info["context"], info["token_budget"]
        return B * ((L - 2) * (ctx * 128 + budget * 2056) + 2 * ctx * 2052)
    if name in ("deltakv", "deltakv_raw") or name.startswith("deltakv_b"):
        ctx, nfull = info["context"], info["full_layers"]
        keep, K, view = 2048, 4, 8 + 2048 + 256
        sparse = keep * (K * 2048 + 160 + 2048) + view * 2052
        full = ctx * (712 if info["kivi"] else 2052)
        return B * ((L - nfull) * sparse + nfull * full)
    raise ValueError(name)


def build_h2o(B: int):
    budget, interval = 4096, 128
    cfg = Config.from_kwargs(sparse_method="h2o", max_model_len=131072, max_num_seqs_in_gpu=B,
                             num_kvcache_slots=B * (budget + interval) + 4096, h2o_decode_budget=budget,
                             h2o_decode_eviction_interval=interval, h2o_prefill_budget=8192, **QWEN)
    drv = SparseDecodeDriver(cfg)
    drv.cache_manager.permute_free_slots(1)
    drv.admit_resident_rows(B, budget, logical_len=131072, seed=0, device_rng=True)
    return drv, dict(batch=B, context=131072, resident=budget)


# ---------------------------------------------------------------------------------------------------------------------
# kernel-level figures of a path: the launches of its main kernels, recorded during ONE eagerly launched step and
# re-issued - between the step's layer loop and its post_forward, on the step's own data - as a hipGraph of back-to-back
# launches per kernel, replayed between one pair of HIP events (the recorded Python arguments keep every tensor alive;
# all of these launches are idempotent).
# ---------------------------------------------------------------------------------------------------------------------
KV_TOKEN_BYTES = 2 * 4 * 128 * 2          # K + V row of one token, Qwen2.5-7B heads, bf16


def _lens_sum(t) -> float:
    return float(t.sum().item())


def _kernel_specs(name: str):
    """-> [(label, module, attribute, group(args, kw) -> suffix, bytes(args, kw) -> algorithmic bytes of the launch)]"""
    import sparse_vllm_amd.layers.attention as attn_mod
    specs = [
        ("decode_stage1_kernel_v3 (scored, svk_flash_decode_stage1)", attn_mod, "flash_decode_stage1_with_score",
         lambda a, kw: "", lambda a, kw: _lens_sum(a[5]) * (KV_TOKEN_BYTES + 8)),
        ("decode_stage1_kernel_v3 (svk_flash_decode_stage1)", attn_mod, "flash_decode_stage1",
         lambda a, kw: " dense rows" if int(a[6]) > 8192 else "", lambda a, kw: _lens_sum(a[5]) * (KV_TOKEN_BYTES + 4)),
        ("kivi_stage1_tile128_kernel (svk_kivi_decode_stage1)", attn_mod, "full_layer_kivi_flash_decode_stage1",
         lambda a, kw: "", lambda a, kw: _lens_sum(kw["context_lens"]) * 712),
    ]
    if name.startswith("quest"):
        from sparse_vllm_amd.engine.cache_manager import quest as qmod
        # the manager calls quest_ops.<fn>: patch the functions on the module object it imported
        specs += [
            ("quest_score_pages_kernel (svk_quest_score_pages)", qmod.quest_ops, "score_pages", lambda a, kw: "",
             lambda a, kw: float(a[0].shape[0]) * int(kw["n_prev"]) * 2 * a[1].shape[1] * a[1].shape[2] * 2),
            ("quest_build_view_kernel (svk_quest_build_view)", qmod.quest_ops, "build_view", lambda a, kw: "",
             lambda a, kw: float(a[5].shape[0]) * (int(kw["n_prev"]) * 4 + int(kw["max_keep"]) * 4)),
        ]
    return specs


class _TimedGraph:
    """Stand-in for the driver's captured graph with HIP events right around every replay.  (A top-level class on purpose:
    a class defined inside `measure` closes over the graph and is cyclic garbage afterwards - the collector then destroys
    the hipGraph at some later allocation, and when that falls inside the NEXT configuration's stream capture the process
    dies with "operation not permitted when stream is capturing".)"""

    def __init__(self, graph):
        self.graph, self.pairs = graph, []

    def replay(self):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.graph.replay()
        e1.record()
        self.pairs.append((e0, e1))


def kernel_timings(drv, q, k, v, name: str, *, replays: int = 5) -> list[dict]:
    specs = _kernel_specs(name)
    calls: dict[str, list] = {}
    recording = {"on": False}
    saved = []
    for label, mod, attr, group, nbytes in specs:
        orig = getattr(mod, attr)

        def wrapper(*a, _orig=orig, _label=label, _group=group, _nbytes=nbytes, **kw):
            if recording["on"]:
                calls.setdefault(_label + _group(a, kw), []).append((_orig, a, dict(kw), _nbytes))
            return _orig(*a, **kw)

        saved.append((mod, attr, orig))
        setattr(mod, attr, wrapper)
    results: list[dict] = []

    def after_layers():
        recording["on"] = False
        for label, lst in calls.items():
            nb = sum(f(a, kw) for _o, a, kw, f in lst) / len(lst)
            torch.cuda.synchronize()
            for orig, a, kw, _f in lst:          # once eagerly: first-use allocations of the wrappers happen outside the capture
                orig(*a, **kw)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture_without_gc(), torch.cuda.graph(g):
                for orig, a, kw, _f in lst:
                    orig(*a, **kw)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(replays):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (replays * len(lst))
            results.append(dict(kernel=label, launches_per_step=len(lst), kernel_us=round(us, 2),
                                algorithmic_mb_per_launch=round(nb / 1e6, 3), frac=round(nb / (us * 1e-6) / 8.0e12, 4)))
            del g
        calls.clear()

    try:
        was_graph = bool(getattr(drv.config, "decode_cuda_graph", False))
        drv.config.decode_cuda_graph = False
        recording["on"] = True
        drv.step(q, k, v, after_layers=after_layers)
        torch.cuda.synchronize()
        drv.config.decode_cuda_graph = was_graph
    finally:
        for mod, attr, orig in saved:
            setattr(mod, attr, orig)
    return sorted(results, key=lambda r: -r["kernel_us"] * r["launches_per_step"])


def measure(name: str, *, steps: int = 32, warmup: int = 4, graph: bool = True, kernels: bool = True) -> dict:
    """Build the configuration, run `warmup` + `steps` decode steps of its sparse path, -> one result dict.
    StreamingLLM is timed over at least one whole window cycle (rows walk 576 -> 1151 -> 576: 576 steps), so that the
    figure is the cycle mean and not wherever in the cycle a short window happens to sit."""
    t0 = time.perf_counter()
    if name.startswith("h2o_b"):
        drv, info = build_h2o(int(name[5:]))
    else:
        drv, info = build(name)
    if name.startswith("streamingllm"):
        steps = max(int(steps), 576)
    q, k, v = drv.random_step_inputs(seed=1)
    if graph:
        drv.enable_decode_graph()
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    for _ in range(max(warmup, 3 if graph else 0)):          # the graph is captured in the first steps
        drv.step(q, k, v)
    torch.cuda.synchronize()
    # two timed windows, the faster one reported: the GPU boxes are shared hosts, and a step of a small configuration is a
    # few hundred microseconds of GPU time behind a Python loop - a stalled host core shows up as a 3-4x slower window now
    # and then (all steps replayed, same kernels, same kernel times).  StreamingLLM's window is one whole cycle, so both
    # windows see the same row lengths.
    windows = []
    for _ in range(2):
        lens = []
        t1 = time.perf_counter()
        for _ in range(steps):
            drv.step(q, k, v)
            lens.append(float(drv.row_len()[0]))
        torch.cuda.synchronize()
        windows.append(((time.perf_counter() - t1) * 1e3 / steps, sum(lens) / len(lens)))
    ms, mean_len = min(windows)
    # a third, diagnostic window with HIP events right around every replay: the GPU time of the steps without the host's
    # share (the events themselves cost ~10 us of host time per step, so this window is not the reported number)
    graph_ms = None
    real_graph = getattr(drv, "_graph", None) if graph else None
    if real_graph is not None and steps <= 64:
        timed = _TimedGraph(real_graph)
        drv._graph = timed
        for _ in range(steps):
            drv.step(q, k, v)
        torch.cuda.synchronize()
        drv._graph = real_graph
        if timed.pairs:
            graph_ms = sum(a.elapsed_time(b) for a, b in timed.pairs) / len(timed.pairs)
        timed.graph, timed.pairs = None, []
    real_graph = None
    nbytes = algorithmic_bytes_per_step(name, info, mean_row_len=mean_len)
    stats = dict(getattr(drv, "graph_stats", {}) or {})
    res = dict(config=name, ms_per_step=round(ms, 4), tokens_per_s=round(info["batch"] / ms * 1e3, 1),
               algorithmic_mb_per_step=round(nbytes / 1e6, 1), roofline_frac=round(nbytes / (ms * 1e-3) / 8.0e12, 4),
               mean_row_len=round(mean_len, 1), graph=bool(graph), steps=steps, setup_s=round(setup, 1), **info)
    res["window_ms_per_step"] = [round(w[0], 4) for w in windows]
    if graph_ms is not None:
        res["graph_ms_per_step"] = round(graph_ms, 4)          # GPU time of the replayed step alone (diagnostic window)
    if graph and stats:
        res["graph_steps"] = stats               # how the warm-up + timed steps ran: eager / captured / replayed
        gen = getattr(drv.cache_manager, "device_step_generation", None)
        if gen is not None:
            res["device_step_generation"] = int(gen)
    if kernels:
        try:
            ks = kernel_timings(drv, q, k, v, name)
            res["kernels"] = ks
            if ks:
                res["dominant_kernel"], res["kernel_us"], res["kernel_frac"] = ks[0]["kernel"], ks[0]["kernel_us"], ks[0]["frac"]
        except Exception as e:      # a failing side leg must not lose the step timing
            res["kernels_error"] = f"{type(e).__name__}: {e}"
    drv._graph = None                    # the configuration's graph dies here, not whenever the collector finds the driver
    del drv, q, k, v
    gc.collect()
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="streamingllm,quest,deltakv_raw,deltakv,vanilla")
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--recon-sub-batches", default="", help="developer: DeltaKV look-ahead layers per launch group, e.g. 1,2,3")
    ap.add_argument("--no-recon-into-view", action="store_true", help="developer: DeltaKV reconstruction into scratch slots + full view copy")
    args = ap.parse_args()
    if args.no_recon_into_view:
        from sparse_vllm_amd.engine.cache_manager.deltakv import DeltaKVCacheManager
        DeltaKVCacheManager._RECON_INTO_VIEW_DEFAULT = False
    if args.recon_sub_batches:
        from sparse_vllm_amd.engine.cache_manager.deltakv import DeltaKVCacheManager
        DeltaKVCacheManager._RECON_SUB_BATCHES = [int(x) for x in args.recon_sub_batches.split(",")]
        DeltaKVCacheManager._RECON_SUB_BATCHES_WIDE = list(DeltaKVCacheManager._RECON_SUB_BATCHES)
    for name in args.configs.split(","):
        try:
            res = measure(name, steps=args.steps, warmup=args.warmup, graph=args.graph)
        except Exception as e:          # one failing configuration must not take the others with it
            res = {"config": name, "error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
