"""End-to-end decode leg: a random-weight Qwen2.5-7B-SHAPED decoder whose attention is this build's hot path.

MEASUREMENT SCAFFOLDING, not product code ("random bf16 Qwen2.5-7B-shaped weights when measuring
end-to-end tok/s"; the reference's model is models/qwen2.py:111-148,262-297 with the tensor-parallel all-reduce of
layers/linear.py:588).  Dense layers are plain torch (`F.linear` -> hipBLASLt, `F.rms_norm`, rotate-half RoPE); there is
no tokenizer, weight loader, scheduler or server.  Per layer the model does what the reference's does: qkv projection
(with bias), RoPE, `cache_manager.save_rope_kv_if_needed`, `Attention.forward` (the sparse path: H2O scored decode),
o_proj, [all-reduce], MLP, [all-reduce]; then final norm, vocab-sharded lm_head, greedy token.  The whole step - slot
allocation, 28 layers, score epilogue, predicated eviction burst, sampling - replays as ONE hipGraph.

Tensor parallel (`--tp t`, t in {1, 2, 4}: 28 q / 4 kv heads): rank r of a TP group holds Hq/t query heads, Hkv/t KV
heads, inter/t MLP columns, vocab/t logits rows; `dist.all_reduce` after o_proj and down_proj (RCCL over xGMI on GPUs,
gloo in the CPU test).  WORLD_SIZE / t replica groups run independently; tokens/s is the sum over groups.

    python tools/e2e_decoder.py --batches 1,64,256 --steps 32            # one GPU
    torchrun --nproc-per-node 4 tools/e2e_decoder.py --tp 2 --batches 64  # two TP-2 replicas

Prints one JSON line per (batch, launch mode).  `bound_ms` is BASELINE.md section 2's end-to-end lower bound per step,
(weights 14.1 GB / t + B x 0.243 GB / t) / 8 TB/s, on one GPU of the group.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# The dense layers are library GEMMs at M <= 256.  PyTorch's TunableOp picks the fastest hipBLASLt / rocBLAS solution per
# GEMM shape on first use (< 10 s for the five shapes of a batch size; results cached in a file under /tmp) instead of the
# library heuristic's default: 10.3 -> 8.9 ms per step at B=64.  No kernel of this build is involved; opt out with
# PYTORCH_TUNABLEOP_ENABLED=0.  (Read by torch at its first GEMM: set before the import.)
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(os.environ.get("TMPDIR", "/tmp"), "svk_e2e_tunableop.csv"))

import torch
import torch.distributed as dist
import torch.nn.functional as F

QWEN25_7B = dict(hidden=3584, layers=28, q_heads=28, kv_heads=4, head_dim=128, inter=18944, vocab=152064, rope_theta=1e6,
                 rms_eps=1e-6)
HBM_PEAK = 8.0e12


class QwenShapedDecoder:
    """Weights of one tensor-parallel rank + the decode step's dense math.  `attention(layer, q, k, v) -> o` is the hook
    the caller fills (the sparse path on a GPU; a stand-in in the CPU test)."""

    def __init__(self, shape: dict, *, tp_rank: int = 0, tp_size: int = 1, group=None, device="cuda:0", dtype=torch.bfloat16,
                 seed: int = 0, max_positions: int = 1 << 18, force_collectives: bool = False):
        s = dict(shape)
        for k in ("q_heads", "kv_heads", "inter", "vocab"):
            if s[k] % tp_size:
                raise ValueError(f"{k}={s[k]} is not divisible by tp={tp_size}")
        self.s, self.tp_rank, self.tp_size, self.group, self.device, self.dtype = s, tp_rank, tp_size, group, torch.device(device), dtype
        self.collectives = tp_size > 1 or force_collectives
        self.hq, self.hkv, self.D = s["q_heads"] // tp_size, s["kv_heads"] // tp_size, s["head_dim"]
        H, I, V = s["hidden"], s["inter"] // tp_size, s["vocab"] // tp_size
        self.vocab_shard = V
        # every rank draws the FULL tensors from the same seed and keeps its shard: a TP group then computes exactly what
        # one rank with tp=1 computes (the CPU test relies on it); on a GPU the full tensor lives only for the draw
        g = torch.Generator(device=self.device).manual_seed(seed)

        def draw(rows, cols, std):
            return (torch.randn((rows, cols), generator=g, device=self.device, dtype=torch.float32) * std).to(dtype)

        def rows_of(t, n_blocks, r):            # row-shard `t` = [n_blocks * k, cols] -> block r
            return t.view(n_blocks, -1, t.shape[-1])[r].contiguous()

        r, t = tp_rank, tp_size
        self.embed = draw(s["vocab"], H, 0.02)
        self.layers = []
        for _ in range(s["layers"]):
            wq, wk, wv = draw(s["q_heads"] * self.D, H, H ** -0.5), draw(s["kv_heads"] * self.D, H, H ** -0.5), draw(s["kv_heads"] * self.D, H, H ** -0.5)
            bq, bk, bv = draw(1, s["q_heads"] * self.D, 0.02), draw(1, s["kv_heads"] * self.D, 0.02), draw(1, s["kv_heads"] * self.D, 0.02)
            wo = draw(H, s["q_heads"] * self.D, (s["q_heads"] * self.D) ** -0.5)
            wg, wu = draw(s["inter"], H, H ** -0.5), draw(s["inter"], H, H ** -0.5)
            wd = draw(H, s["inter"], s["inter"] ** -0.5)
            self.layers.append(dict(
                qkv_w=torch.cat([rows_of(wq, t, r), rows_of(wk, t, r), rows_of(wv, t, r)]),
                qkv_b=torch.cat([rows_of(bq.t().contiguous(), t, r), rows_of(bk.t().contiguous(), t, r),
                                 rows_of(bv.t().contiguous(), t, r)]).view(-1),
                o_w=wo.view(H, t, -1)[:, r].contiguous(),
                gate_up_w=torch.cat([rows_of(wg, t, r), rows_of(wu, t, r)]),
                down_w=wd.view(H, t, -1)[:, r].contiguous(),
                ln1=torch.ones(H, device=self.device, dtype=dtype), ln2=torch.ones(H, device=self.device, dtype=dtype)))
            del wq, wk, wv, wo, wg, wu, wd
        self.final_ln = torch.ones(H, device=self.device, dtype=dtype)
        self.lm_head = rows_of(draw(s["vocab"], H, H ** -0.5), t, r)
        self.inter_shard = I
        inv = 1.0 / (s["rope_theta"] ** (torch.arange(0, self.D, 2, device=self.device, dtype=torch.float32) / self.D))
        ang = torch.arange(max_positions, device=self.device, dtype=torch.float32)[:, None] * inv[None, :]
        self.cos, self.sin = ang.cos().to(dtype), ang.sin().to(dtype)

    def weight_bytes(self) -> int:
        """What one decode step reads of this rank's weights (the embedding is a B-row gather: excluded)."""
        n = sum(t.numel() for L in self.layers for t in L.values()) + self.final_ln.numel() + self.lm_head.numel()
        return int(n * self.embed.element_size())

    def _rope(self, x, cos, sin):
        x1, x2 = x[..., : self.D // 2], x[..., self.D // 2:]
        return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)

    def _all_reduce(self, t):
        if self.collectives:
            dist.all_reduce(t, group=self.group)
        return t

    def step(self, tokens: torch.Tensor, positions: torch.Tensor, attention, before_layer=None) -> torch.Tensor:
        """tokens [B] int64, positions [B] int64 -> next tokens [B] int64 (greedy)."""
        s, eps = self.s, self.s["rms_eps"]
        h = self.embed[tokens]
        cos, sin = self.cos[positions][:, None, :], self.sin[positions][:, None, :]
        B = h.shape[0]
        for l, L in enumerate(self.layers):
            if before_layer is not None:
                before_layer(l)
            x = F.rms_norm(h, (s["hidden"],), L["ln1"], eps)
            qkv = F.linear(x, L["qkv_w"], L["qkv_b"])
            q, k, v = qkv.split([self.hq * self.D, self.hkv * self.D, self.hkv * self.D], dim=-1)
            q = self._rope(q.view(B, self.hq, self.D), cos, sin)
            k = self._rope(k.view(B, self.hkv, self.D), cos, sin)
            o = attention(l, q, k, v.reshape(B, self.hkv, self.D).contiguous())     # K and V rows share one layout (store contract)
            h = h + self._all_reduce(F.linear(o.reshape(B, -1), L["o_w"]))
            x = F.rms_norm(h, (s["hidden"],), L["ln2"], eps)
            gate, up = F.linear(x, L["gate_up_w"]).split(self.inter_shard, dim=-1)
            h = h + self._all_reduce(F.linear(F.silu(gate) * up, L["down_w"]))
        logits = F.linear(F.rms_norm(h, (s["hidden"],), self.final_ln, eps), self.lm_head).float()      # [B, vocab / t]
        best, idx = logits.max(dim=-1)
        idx = idx + self.tp_rank * self.vocab_shard
        if self.collectives and self.tp_size > 1:
            both = torch.stack((best, idx.to(best.dtype)), dim=-1).contiguous()                        # [B, 2]
            flat = torch.empty((self.tp_size * both.shape[0], 2), dtype=both.dtype, device=both.device)
            dist.all_gather_into_tensor(flat, both, group=self.group)                                  # rank-major concatenation
            gathered = flat.view(self.tp_size, both.shape[0], 2)
            win = gathered[..., 0].argmax(dim=0)                                                       # first max = lowest rank
            idx = gathered[..., 1].gather(0, win[None, :])[0].long()
        return idx


def _build_sparse_driver(B: int, model: QwenShapedDecoder, device: str, rank: int, *, layers: int):
    """The headline workload's resident state (bench.py): B rows of 4096 tokens (131 072 logical), permuted slot pool."""
    from sparse_vllm_amd.config import Config
    from tools.synthetic import SyntheticDecodeDriver
    budget, interval = 4096, 128

    class E2EDriver(SyntheticDecodeDriver):
        """`_forward_layers` runs the model instead of replaying given q / k / v: everything else of the step (allocation,
        hipGraph capture / replay, device-resident burst, post_forward) is the driver's."""
        model = None
        tokens = positions = None

        def _forward_layers(self, q, k, v, outputs):
            from sparse_vllm_amd.utils.context import set_context
            cm, sc = self.cache_manager, self.sparse_controller
            ctx = set_context(False, cache_manager=cm, sparse_controller=sc)
            sc.prepare_forward(self.seqs, False)

            def before_layer(l):
                ctx.now_layer_idx = l

            def attention(l, q_, k_, v_):
                cm.save_rope_kv_if_needed(l, k_, v_)            # models/qwen2.py:126-131
                return self.attn(q_, k_, v_)

            nxt = self.model.step(self.tokens, self.positions, attention, before_layer)
            cm.flush_deferred_decode_store()
            self.tokens.copy_(nxt)
            self.positions.add_(1)

    conf = Config.from_kwargs(sparse_method="h2o", num_hidden_layers=layers, num_attention_heads=model.hq,
                              num_key_value_heads=model.hkv, head_dim=model.D, max_model_len=131072 + 4096, max_num_seqs_in_gpu=B,
                              num_kvcache_slots=B * (budget + interval) + 4096, h2o_decode_budget=budget,
                              h2o_decode_eviction_interval=interval, h2o_prefill_budget=8192, engine_prefill_chunk_size=8192,
                              device=device)
    drv = E2EDriver(conf)
    drv.model = model
    drv.cache_manager.permute_free_slots(20260625 + rank)
    drv.admit_resident_rows(B, budget, logical_len=131072, seed=20260625 + rank, device_rng=True)
    g = torch.Generator(device=device).manual_seed(1 + rank)
    drv.tokens = torch.randint(0, model.s["vocab"], (B,), generator=g, device=device)
    drv.positions = torch.full((B,), 131072, dtype=torch.long, device=device)
    return drv


def rccl_sanity(world: int, rank: int, device: str, group, tp: int, batches) -> dict:
    """One record about the collective the model's tensor parallelism uses (layers/linear.py:588): how many ranks the job
    really has (all-reduce SUM of ones over the world) and what one all-reduce of a decode step's activations - [B, 3584]
    bf16 = 7 KB x B - costs inside the TP group, eager and as nodes of a replayed hipGraph (56 per step in the model)."""
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    rec = {"rccl_sanity": True, "ranks_seen": int(ones.item()), "world_size": world, "tp": tp, "backend": dist.get_backend(),
           "allreduce": []}
    for B in batches:
        x = torch.randn(B, QWEN25_7B["hidden"], device=device).to(torch.bfloat16)
        n = 56
        for _ in range(5):
            dist.all_reduce(x, group=group)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            dist.all_reduce(x, group=group)
        torch.cuda.synchronize()
        eager_us = (time.perf_counter() - t0) / n * 1e6
        graph_us = None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n):
                    dist.all_reduce(x, group=group)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                g.replay()
            torch.cuda.synchronize()
            graph_us = (time.perf_counter() - t0) / (4 * n) * 1e6
        except Exception as e:                   # capture support is a property of the RCCL build: report, do not fail the leg
            graph_us = f"{type(e).__name__}: {e}"[:160]
        rec["allreduce"].append({"batch": B, "bytes": int(x.numel() * 2), "eager_us": eager_us, "graph_replay_us": graph_us})
    worst = torch.tensor([max(a["eager_us"] for a in rec["allreduce"])], dtype=torch.float64, device=device)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    rec["slowest_rank_eager_us"] = float(worst.item())
    return rec


def bound_ms(batch: int, tp: int, weight_bytes: int | None = None) -> float:
    w = 14.1e9 / tp if weight_bytes is None else float(weight_bytes)
    return (w + batch * 0.243e9 / tp) / HBM_PEAK * 1e3


def measure(model, B: int, *, steps: int, warmup: int, graph: bool, device: str, rank: int, sync) -> dict:
    drv = _build_sparse_driver(B, model, device, rank, layers=model.s["layers"])
    if graph:
        drv.enable_decode_graph()
    dummy = torch.zeros(1, device=device)
    for _ in range(max(warmup, 3)):
        drv.step(dummy, dummy, dummy)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        drv.step(dummy, dummy, dummy)
    sync()
    dt = time.perf_counter() - t0
    toks = drv.tokens.tolist()[:4]
    stats = dict(getattr(drv, "graph_stats", {}) or {})
    del drv
    torch.cuda.empty_cache()
    return {"ms_per_step": dt / steps * 1e3, "seconds": dt, "graph_steps": stats, "sample_tokens": toks}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,64,256")
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--tp", type=int, default=1)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--modes", default="graph", help="comma list of graph,eager (each prints its own line, in this order)")
    ap.add_argument("--force-collectives", action="store_true", help="issue the TP all-reduces even at tp=1 (1-GPU check of "
                    "the RCCL calls, eager and under hipGraph capture)")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    tp = int(args.tp)
    if world % tp:
        raise SystemExit(f"WORLD_SIZE {world} is not a multiple of --tp {tp}")
    use_dist = world > 1 or args.force_collectives
    group = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=torch.device(device), rank=rank, world_size=world)
        if tp > 1 and world > tp:
            for g0 in range(0, world, tp):             # every rank creates every group (torch.distributed's rule)
                grp = dist.new_group(list(range(g0, g0 + tp)))
                if g0 <= rank < g0 + tp:
                    group = grp
    shape = dict(QWEN25_7B, layers=int(args.layers))
    model = QwenShapedDecoder(shape, tp_rank=rank % tp, tp_size=tp, group=group, device=device, seed=7 + rank // tp,
                              force_collectives=args.force_collectives)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    n_groups = world // tp
    if use_dist:
        try:
            rec = rccl_sanity(world, rank, device, group, tp, [int(x) for x in args.batches.split(",") if x])
        except Exception as e:
            rec = {"rccl_sanity": True, "error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            print(json.dumps(rec), flush=True)
    for B in [int(x) for x in args.batches.split(",") if x]:
        for mode in [m for m in args.modes.split(",") if m]:
            try:
                r = measure(model, B, steps=args.steps, warmup=args.warmup, graph=(mode == "graph"), device=device, rank=rank,
                            sync=sync)
            except Exception as e:           # a failed mode must not take the other lines with it
                if rank == 0:
                    print(json.dumps({"e2e": True, "batch_per_group": B, "tp": tp, "launch": mode, "error": f"{type(e).__name__}: {e}"[:300]}),
                          flush=True)
                continue
            secs = torch.tensor([r["seconds"]], dtype=torch.float64, device=device)
            if use_dist:
                dist.all_reduce(secs, op=dist.ReduceOp.MAX)
            step_ms = float(secs.item()) / args.steps * 1e3
            if rank == 0:
                lb = bound_ms(B, tp, model.weight_bytes() if args.layers != 28 else None)
                print(json.dumps({
                    "e2e": True, "metric": "end-to-end decode tokens/s at 128k ctx, H2O budget=4k, Qwen2.5-7B-shaped random weights",
                    "value": n_groups * B / (step_ms * 1e-3), "unit": "tokens/s", "ms_per_step": step_ms, "batch_per_group": B,
                    "tp": tp, "replica_groups": n_groups, "n_gpus": world, "launch": "hipGraph replay" if mode == "graph" else "eager",
                    "bound_ms": lb, "frac_of_bound": lb / step_ms, "layers": int(args.layers),
                    "weight_bytes_per_rank": model.weight_bytes(), "collectives": bool(model.collectives),
                    "gemm_selection": "TunableOp" if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1" else "library default",
                    "graph_steps": r["graph_steps"], "sample_tokens": r["sample_tokens"], "steps": args.steps}), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
