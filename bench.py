#!/usr/bin/env python3
"""bench.py — decode tokens/s of the MI355X engine on BASELINE.json's metric, with the roofline of the dominant
kernel and the CPU baseline measured in the same run.

    python bench.py --gpus 1 --steps 128 --warmup 8          (default: Llama-3-8B Q4_K_M, 128-token prompt)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N   (layer pipeline)
    python bench.py --gpus N                                  (the same: the GPU-free parent starts that launcher itself)

Protocol = the reference's `llama-gguf bench` (src/main.rs:1750-1871): prompt tokens i % 32000, prefill, then
greedy decode feeding back the arg-max (last-max tie rule).  A "step" is one decoded token.  Weights are
synthetic random-init blocks of the named architecture and quantization mix (no model files exist here);
they are resident in HBM before the timed region and the token feedback stays on device
(lgh_decode_greedy), so `value` contains no PCIe traffic; the PCIe-inclusive rate (full logits D2H per token,
as GpuInference::forward returns them) is reported beside it as `pcie_inclusive_tokens_per_s`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

# HBM bytes per launch of the dominant kernel come from rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE,
# guides/MI355X_MICROARCH.md "HBM traffic"), which cannot be collected from inside this process: tools/gpu_measure.sh
# runs them on the same command and tools/summarize_profiles.py commits the per-kernel table under profiles/.  The
# newest committed table is what `roofline.traffic` reports (with its file name), null when there is none.
def committed_traffic(kernel):
    import csv
    tables = sorted((Path(__file__).resolve().parent / "profiles").glob("*_pmc_traffic.csv"))
    for t in reversed(tables):
        with open(t, newline="") as f:
            for row in csv.DictReader(f):
                if row["kernel"] == kernel and int(row["launches_fetch_pass"]) > 0:
                    return int(float(row["hbm_bytes_per_launch_corrected"])), t.name
    return None, None


def traffic_age(table_name):
    """Which profile the PMC table came from and at which commit it was collected (profiles/<tag>_meta.json, written by
    tools/summarize_profiles.py), so that a stale table is visible in the bench line."""
    if not table_name:
        return None
    tag = table_name.split("_pmc_traffic")[0]
    meta = Path(__file__).resolve().parent / "profiles" / f"{tag}_meta.json"
    out = {"profile": tag}
    if meta.exists():
        try:
            out.update(json.loads(meta.read_text()))
        except ValueError:
            pass
    return out


MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16/bf16 (MI355X_MICROARCH.md)
# activations, accumulation and every epilogue are f32 as in the reference; the quantized mat-vec feeds the int8 matrix cores
# with x split exactly into four int8 limbs per element (csrc/xq.h) — a documented deviation from "wavefront reductions"
DTYPE_LABEL = "f32 (x as 4xint8 limbs on the matrix cores, exact int32 accumulation)"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6290 GB/s is the measured copy rate


def roofline_from_stats(st, P, step_us):
    """`roofline` object for the kernel with the largest share of device time, from an eager profiled pass of P steps.

    A hipEvent pair around ONE launch also times the bracket itself (event packets on the queue).  That fixed cost is
    measured live with empty brackets in mid-stream and removed: `kernel_only_us`.  rocprofv3 --kernel-trace reports
    back-to-back graph nodes with start(i+1) == end(i), i.e. each kernel's duration INCLUDES its dispatch gap; the same
    figure is derived here from the timed graph replay: gap = (replayed step time - sum of kernel-only times) / graph
    nodes.  `avg_launch_us` = kernel_only + gap is the number that must agree with the rocprofv3 average committed under
    profiles/ and the one the roofline fraction is computed from.  (step_us None: no gap — pipeline stages, where the
    step also contains the hops.)"""
    syms = st["symbols"]
    dom = max(syms, key=lambda s: syms[s]["time_us"])
    d = syms[dom]
    bracket_us = st["event_bracket_us"]
    n_launch = sum(v["launches"] for v in syms.values()) / P
    kernel_sum_us = sum(max(v["time_us"] - bracket_us * v["launches"], 0.0) for v in syms.values()) / P
    gap_us = 0.0 if step_us is None else max(step_us - kernel_sum_us, 0.0) / max(n_launch, 1.0)
    raw_us = d["time_us"] / d["launches"]
    kernel_only_us = max(raw_us - bracket_us, 1e-3)
    avg_us = kernel_only_us + gap_us
    bytes_per_launch = d["alg_bytes"] / d["launches"]
    achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
    traffic, traffic_src = committed_traffic(dom)
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "traffic_age": traffic_age(traffic_src),
            "launches_per_step": d["launches"] / P, "avg_launch_us": round(avg_us, 3),
            "kernel_only_us": round(kernel_only_us, 3), "dispatch_gap_us": round(gap_us, 3),
            "avg_launch_us_with_event_bracket": round(raw_us, 3), "event_bracket_us": round(bracket_us, 3),
            "alg_bytes_per_launch": int(bytes_per_launch)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--mix", default="Q4_K_M")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget (0 disables)")
    ap.add_argument("--profile-steps", type=int, default=16, help="steps of the eager hipEvent pass (0 disables)")
    ap.add_argument("--attn-splits", type=int, default=0)
    ap.add_argument("--attn-direct", type=int, default=0, help="single-launch decode attention up to 64*n rows (0 = default, 255 = never)")
    ap.add_argument("--batch", type=int, default=0, help="multi-sequence decode with this many sequences per step (1..16); not the headline metric")
    ap.add_argument("--reps", type=int, default=3, help="timed repetitions of the K-step region (SURVEY.md §8d: 1 warm-up + 3, mean and min)")
    ap.add_argument("--flags", type=int, default=0, help="extra LGH_FLAG_* bits for the engine context")
    ap.add_argument("--kv-cache-type", default="f32",
                    help="KV cache format, the reference's --kv-cache-type strings (f32, tq2, tq3, tq2-qjl, tq3-qjl; src/config.rs:808-817) plus "
                         "int8 / fp8e4m3 / fp8e5m2 (QuantizedKVCache's formats); not the headline metric")
    ap.add_argument("--inlib", action="store_true",
                    help="N > 1 in ONE process: the in-library pipeline (lgh_pipeline_*: peer copies + events between the stages' streams) "
                         "instead of one rank per GPU over RCCL")
    ap.add_argument("--stages", type=int, default=0, help="with --inlib: stage count when it differs from --gpus (several stages on one GPU)")
    ap.add_argument("--fake-stage", action="store_true",
                    help="CPU rehearsal of the N>1 launcher and hop protocol: gloo + pipeline.FakeStage, no GPU, no engine (not a measurement)")
    return ap.parse_args()


def prompt_tokens(n: int, vocab: int):
    return [(i % 32000) % vocab for i in range(n)]  # main.rs:1787


def cpu_baseline(pkg, model, cfg, budget_s: float):
    """The oracle (a C++ port of the reference CPU backend's arithmetic) timed on this host's cores, on a
    bounded sample of the same workload: same weights, short prefill, a few decode tokens."""
    orc = graft.load_oracle()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)                     # the GPU box gives one GPU's share of the host: 16 cores
    orc.set_threads(cores)
    ref = orc.Model(cfg.as_dict())
    for name, t, ne, data in model.tensors(keep=True):
        ref.add_tensor(name, t, ne, data)
    ref.finalize()
    pre = prompt_tokens(4, cfg.vocab_size)
    t0 = time.perf_counter()
    logits = ref.forward(pre)                      # also dequantizes the embedding table once
    prefill_s = time.perf_counter() - t0
    tok, n, t0 = orc.argmax_last(logits), 0, time.perf_counter()
    while True:
        logits = ref.forward([tok])
        tok = orc.argmax_last(logits)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 32:
            break
    isa = {1: "scalar", 2: "avx2", 3: "avx512"}[orc.get_isa()]
    # the reference's CPU forward as it stands re-dequantizes the WHOLE embedding table on every call (llama.rs:181-193, 288):
    # two tokens in that faithful mode, reported next to the kernel-only rate (SURVEY §8d asks for both)
    t0 = time.perf_counter()
    for _ in range(2):
        tok = orc.argmax_last(ref.forward([tok], faithful_embedding=True))
    faithful = 2 / (time.perf_counter() - t0)
    ref.close()
    return {"value": round(n / el, 4), "unit": "tokens/s", "cores": cores, "kind": "port",
            "faithful_embedding_value": round(faithful, 4),
            "sample": f"{n} greedy decode tokens after a 4-token prefill ({prefill_s:.1f}s, untimed), same weights, "
                      f"oracle C++ port of the reference CPU backend, kernel-only mode (embedding table dequantized "
                      f"once, not per call), dot_f32 ISA path {isa}; faithful_embedding_value = 2 more tokens with the reference's "
                      f"per-call dequantization of the whole embedding table (llama.rs:288)"}


KV_TYPES = {"f32": 0, "int8": 1, "fp8e4m3": 2, "fp8e5m2": 3, "tq2": 4, "turboquant2": 4, "tq3": 5, "turboquant3": 5,
            "tq2-qjl": 6, "turboquant2-qjl": 6, "tq3-qjl": 7, "turboquant3-qjl": 7}


def kv_cache_type_id(args) -> int:
    """--kv-cache-type -> LGH_KV_* (the reference's strings, src/config.rs:808-817, plus QuantizedKVCache's formats)."""
    if args.kv_cache_type.lower() not in KV_TYPES:
        raise SystemExit(f"bench.py: unknown --kv-cache-type {args.kv_cache_type}")
    return KV_TYPES[args.kv_cache_type.lower()]


def run_single(args, pkg):
    import torch
    hb = pkg.hip_backend
    if hb.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible (the engine has no CPU fallback)")
    W, K, P = args.warmup, args.steps, args.profile_steps
    max_seq = max(512, args.prompt + W + K + P + 16)
    cfg = pkg.make_config(args.model, max_seq_len=max_seq)
    want_cpu = args.cpu_seconds > 0
    model = pkg.SynthModel(cfg, mix=args.mix)
    t0 = time.perf_counter()
    eng = pkg.HipGpuInference.from_model(_Keep(model) if want_cpu else model, max_seq, attn_splits=args.attn_splits, attn_direct=args.attn_direct,
                                         flags=args.flags, kv_cache_type=kv_cache_type_id(args))
    load_s = time.perf_counter() - t0

    prompt = prompt_tokens(args.prompt, cfg.vocab_size)
    for t in prompt[:-1]:                                              # the exact (f32) prompt path: the decode that is timed
        eng.prefill_token(t)                                           # starts from the cache the CPU reference would have
    eng.forward(prompt[-1])                                           # prefill = Model::forward(prompt) (main.rs:1804-1807)
    warm = eng.decode_greedy(prompt[-1], W) if W > 0 else np.array([prompt[-1]], np.uint32)  # main.rs:1812-1822
    tok = int(warm[-1])
    kv0 = eng.position()
    # ---- timed region (SURVEY.md §8d protocol): W warm-up steps above, then `reps` repetitions of EXACTLY K decode steps,
    # every repetition from the same cache state (the rows past kv0 are forgotten in between, so each one decodes the same K
    # tokens at kv_len kv0+1..kv0+K); device-resident token feedback; `value` is the mean, the best repetition is reported too
    rep_s, toks = [], None
    for _ in range(max(args.reps, 1)):
        eng.kv_truncate(kv0)
        torch.cuda.synchronize()
        eng.synchronize()
        t0 = time.perf_counter()
        toks_r = eng.decode_greedy(tok, K)
        eng.synchronize()
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t0)
        if toks is not None and not np.array_equal(toks, toks_r):
            raise SystemExit("bench.py: repetitions decoded different tokens (the engine must be deterministic)")
        toks = toks_r
    elapsed = sum(rep_s) / len(rep_s)
    kv1 = eng.position()
    tok_s = K / elapsed
    mean_kv = (kv0 + 1 + kv1) / 2.0
    step_bytes = model.step_alg_bytes(int(round(mean_kv)))
    # ---- PCIe-inclusive variant (GpuInference::forward returns all logits to the host)
    orc_argmax = lambda v: int(len(v) - 1 - np.argmax(v[::-1]))      # last maximal index (main.rs:1815-1821)
    n_pcie = min(K, 32)
    # ---- prompt processing (untimed for the metric): the batched f16-GEMM path against n exact prefill_tokens
    prefill = None
    if len(prompt) > 2:
        eng.reset()
        eng.synchronize()
        t0 = time.perf_counter()
        for t in prompt[:-1]:
            eng.prefill_token(t)
        eng.synchronize()
        seq_s = time.perf_counter() - t0
        eng.reset()
        eng.forward_batch(prompt[:-1])                                 # first use allocates the scratch
        bat_all = []
        for _ in range(3):                                             # three timed passes: mean and min, like the decode region
            eng.reset()
            eng.synchronize()
            t0 = time.perf_counter()
            eng.forward_batch(prompt[:-1])
            eng.synchronize()
            bat_all.append(time.perf_counter() - t0)
        bat_s = sum(bat_all) / len(bat_all)
        # multiply-adds the prompt pass performs: every layer's seven matrices, except that the last layer stops after
        # its K/V projections (a prefill returns nothing; only the caches survive)
        H, QD, KD, F, NL = cfg.hidden_size, cfg.num_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim, cfg.intermediate_size, cfg.num_layers
        macs = (NL - 1) * (H * (QD + 2 * KD) + QD * H + 3 * H * F) + H * (QD + 2 * KD)
        tflops = 2.0 * macs * (len(prompt) - 1) / bat_s / 1e12
        prefill = {"tokens": len(prompt) - 1, "batched": bool(eng.prefill_is_batched()),
                   "forward_batch_tokens_per_s": round((len(prompt) - 1) / bat_s, 1), "forward_batch_ms": round(1e3 * bat_s, 3),
                   "forward_batch_ms_min": round(1e3 * min(bat_all), 3),
                   "token_by_token_tokens_per_s": round((len(prompt) - 1) / seq_s, 1),
                   "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tflops / MFMA_F16_PEAK_TFLOPS, 4), "dtype": "f16 operands, f32 accumulation",
                                "note": "whole prompt pass (GEMMs + attention + row kernels); flops counted for dense layers"}
                   if eng.prefill_is_batched() and not cfg.num_experts else None}
    eng.reset()
    for t in prompt[:-1]:
        eng.prefill_token(t)
    eng.forward(prompt[-1])
    t0 = time.perf_counter()
    t = prompt[-1]
    for _ in range(n_pcie):
        t = orc_argmax(eng.forward(t))
    pcie_tok_s = n_pcie / (time.perf_counter() - t0)
    # ---- roofline of the dominant kernel: eager pass with a hipEvent pair around every launch, on the launch stream
    roofline = None
    kernels = {}
    if P > 0:
        eng.set_profiling(True)
        t = int(toks[-1])
        for _ in range(P):
            t = eng.forward_argmax(t)
        eng.set_profiling(False)
        st = eng.stats()
        syms = st["symbols"]
        roofline = roofline_from_stats(st, P, 1e6 * elapsed / K)
        tot = sum(v["time_us"] for v in syms.values())
        kernels = {s: {"launches_per_step": v["launches"] / P, "avg_us": round(v["time_us"] / v["launches"], 3),
                       "share": round(v["time_us"] / tot, 4),
                       "GBps": round(v["alg_bytes"] / max(v["time_us"], 1e-9) / 1e3, 1)} for s, v in syms.items()}
        kernels["_classes"] = {k: {"launches_per_step": v["launches"] / P, "avg_us": round(v["time_us"] / v["launches"], 3)}
                               for k, v in st["kernels"].items()}
    stats = eng.stats()
    eng.close()
    try:
        read_peak = pkg.hip_backend.bench_hbm_read(1 << 31, 6)
    except Exception:   # diagnostic only
        read_peak = 0.0
    cpu = cpu_baseline(pkg, model, cfg, args.cpu_seconds) if want_cpu else None
    out = {
        "metric": "decode tokens/sec Llama-3-8B Q4_K_M, 1 GPU; % of HBM roofline" if (args.model, args.mix) == ("llama-3-8b", "Q4_K_M")
                  else f"decode tokens/sec {args.model} {args.mix}, 1 GPU; % of HBM roofline",
        "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_LABEL, "data": "synthetic",
        "repetitions": {"n": len(rep_s), "ms_per_step": [round(1e3 * r / K, 4) for r in rep_s],
                        "min_ms_per_step": round(1e3 * min(rep_s) / K, 4), "best_value": round(K / min(rep_s), 2),
                        "protocol": "W warm-up steps, then n repetitions of exactly K steps from the same cache state; value = K / mean"},
        "config": {"workload": f"{args.model} {args.mix} single-stream greedy decode, seq_len=1, {args.prompt}-token prompt "
                               f"prefilled, kv_len {kv0 + 1}..{kv1}" + ("" if args.kv_cache_type == "f32" else f", KV cache {args.kv_cache_type}"),
                   "quant_mix": args.mix, "prompt_tokens": args.prompt,
                   "parallelism": "single GPU", "weights": "random-init synthetic blocks (SURVEY.md §8d)"},
        "hbm_roofline": {"alg_bytes_per_token": int(step_bytes), "achieved_GBps": round(step_bytes * tok_s / 1e9, 1),
                         "peak_GBps": HBM_PEAK_GBPS, "frac": round(step_bytes * tok_s / 1e9 / HBM_PEAK_GBPS, 4),
                         # the practical ceiling: a streaming read of 1 GiB with the weight stream's access pattern
                         "measured_read_peak_GBps": round(read_peak, 1),
                         "frac_of_measured_peak": round(step_bytes * tok_s / 1e9 / read_peak, 4) if read_peak else None},
        "roofline": roofline, "cpu_baseline": cpu,
        "pcie_inclusive_tokens_per_s": round(pcie_tok_s, 2), "prefill": prefill,
        "graph_nodes_per_token": stats["graph_nodes"], "weight_bytes_resident": stats["weight_bytes"],
        "load_seconds": round(load_s, 1), "kernels": kernels,
    }
    print(json.dumps(out))


class _Keep:
    """Makes from_model keep the generated payloads so the CPU baseline reuses the same bytes."""

    def __init__(self, model):
        self._m, self.config = model, model.config

    def tensors(self, layers=None):
        return self._m.tensors(layers, keep=True)


def run_batch(args, pkg):
    """`--batch B`: multi-sequence decode on one GPU (the reference's BatchedEngine, src/engine_batched.rs) — B
    sequences, each with its own 128-token prompt and KV cache, one token per sequence per step, every weight tile read once per
    step.  NOT the headline metric (that is single-stream decode): value = aggregate tokens/s over the B sequences; the roofline
    object prices the step by its algorithmic bytes (weights once + every sequence's KV rows and vectors)."""
    import torch
    hb = pkg.hip_backend
    if hb.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible (the engine has no CPU fallback)")
    B, W, K = args.batch, args.warmup, args.steps
    reps = max(args.reps, 1)
    max_seq = max(512, args.prompt + W + reps * K + 16)
    cfg = pkg.make_config(args.model, max_seq_len=max_seq)
    model = pkg.SynthModel(cfg, mix=args.mix)
    t0 = time.perf_counter()
    eng = pkg.HipGpuInference.from_model(model, max_seq, attn_splits=args.attn_splits, flags=args.flags, kv_cache_type=kv_cache_type_id(args))
    eng.batch_create(B)
    load_s = time.perf_counter() - t0
    slots = list(range(B))
    prompts = [[(t + 977 * s) % cfg.vocab_size for t in prompt_tokens(args.prompt, cfg.vocab_size)] for s in range(B)]   # sequence s: the bench prompt, shifted
    for s in slots:
        eng.batch_prefill(s, prompts[s][:-1])
    first = [p[-1] for p in prompts]
    warm = eng.decode_greedy_multi(slots, first, 1 + W)
    toks = [int(t) for t in warm[-1]]
    kv0 = eng.batch_position(0)
    rep_s = []
    for _ in range(reps):                                   # every repetition: exactly K steps (the caches keep growing: kv0 + r*K ...)
        torch.cuda.synchronize()
        eng.synchronize()
        t0 = time.perf_counter()
        out = eng.decode_greedy_multi(slots, toks, K)
        eng.synchronize()
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t0)
        toks = [int(t) for t in out[-1]]
    elapsed = sum(rep_s) / len(rep_s)
    kv1 = eng.batch_position(0)
    mean_kv = (kv0 + 1 + kv1) / 2.0
    one = model.step_alg_bytes(int(round(mean_kv)))        # a single sequence's step: weights + its KV rows + vectors
    per_seq_extra = one - model.step_alg_bytes(0)            # the KV rows a sequence reads at the mean position
    kv_write = cfg.num_layers * 2 * cfg.num_kv_heads * cfg.head_dim * 4
    step_bytes = model.step_alg_bytes(0) + B * per_seq_extra + (B - 1) * (cfg.vocab_size * 4 + kv_write)   # weights ONCE + per-sequence rows
    tok_s = B * K / elapsed
    # the single-sequence engine on the same box, same protocol, for the ratio
    single = pkg.HipGpuInference.from_model(model, max_seq, attn_splits=args.attn_splits, flags=args.flags, kv_cache_type=kv_cache_type_id(args))
    single.forward_batch(prompts[0][:-1])
    w1 = single.decode_greedy(prompts[0][-1], 1 + W)
    torch.cuda.synchronize(); single.synchronize()
    t0 = time.perf_counter()
    single.decode_greedy(int(w1[-1]), K)
    single.synchronize()
    single_tok_s = K / (time.perf_counter() - t0)
    print(json.dumps({
        "metric": f"aggregate decode tokens/sec {args.model} {args.mix}, 1 GPU, {B} sequences per step (multi-sequence decode; NOT the headline metric)",
        "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL, "data": "synthetic",
        "repetitions": {"n": reps, "ms_per_step": [round(1e3 * r / K, 4) for r in rep_s]},
        "config": {"workload": f"{args.model} {args.mix}: {B} sequences, {args.prompt}-token prompts (batched prompt path), greedy decode, "
                               f"kv_len {kv0 + 1}..{kv1}, every weight tile read once per step" + ("" if args.kv_cache_type == "f32" else f", KV cache {args.kv_cache_type}"),
                   "batch": B, "quant_mix": args.mix,
                   "parallelism": "single GPU"},
        "per_sequence_tokens_per_s": round(tok_s / B, 2),
        "single_sequence_tokens_per_s_same_box": round(single_tok_s, 2), "speedup_vs_single_sequence": round(tok_s / single_tok_s, 3),
        "roofline": {"bound": "hbm", "achieved": round(step_bytes * (K / elapsed) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(step_bytes * (K / elapsed) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                     "alg_bytes_per_step": int(step_bytes), "alg_bytes_per_token": int(step_bytes / B),
                     "note": "whole step: weights once + B x (KV rows + logits + vectors); from ~4 sequences on the step is bound by the "
                             "matrix-core + vector-ALU work per (tile, sequence) — 4 MFMAs + ~42 vector instructions, the price of "
                             "bit-identical f32-exact activations — not by HBM (profiles/r03e_batched_decode.md)"},
        "load_seconds": round(load_s, 1)}))
    eng.close()
    single.close()


def run_pipeline(args, pkg):
    """N > 1: layers pipeline-sharded over N GPUs, one process per GPU.  Per token and stage boundary one f32[hidden] vector
    goes device to device with RCCL send/recv over xGMI, and the last stage's arg-max word goes straight into the first
    stage's token word the same way (PipelineDecoder.decode_device): no host value crosses a stage boundary per token, the
    host only enqueues.  `--fake-stage`: the same launcher, process group and protocol on CPU (gloo, pipeline.FakeStage)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))   # unset: the 1-rank rehearsal
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    local = int(os.environ.get("LOCAL_RANK", rank))
    assert world == args.gpus or os.environ.get("LGH_BENCH_FORCE_PIPELINE"), f"WORLD_SIZE {world} != --gpus {args.gpus}"
    fake = args.fake_stage
    W, K = args.warmup, args.steps
    reps = max(args.reps, 1)
    if fake:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sync = lambda: None
        n_layers = 8
        lo, hi = pkg.pipeline.split_layers(n_layers, world)[rank]
        stage = pkg.pipeline.FakeStage(lo, hi, rank == 0)
        eng = model = cfg = None
        vocab = stage.VOCAB
        red = lambda v, op: (lambda t: (dist.all_reduce(t, op=op), float(t.item()))[1])(torch.tensor([v], dtype=torch.float64))
    else:
        # LGH_PP_ONE_GPU=1: correctness rehearsal of the multi-process path on a box with ONE GPU — every rank on device 0, hops
        # staged through host memory over gloo (pipeline.HostStagedComm).  The line it prints says so; it is not a measurement.
        one_gpu = os.environ.get("LGH_PP_ONE_GPU", "") not in ("", "0")
        if one_gpu:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev)
        sync = torch.cuda.synchronize
        max_seq = max(512, args.prompt + W + reps * K + min(args.profile_steps, 8) + 16)
        cfg = pkg.make_config(args.model, max_seq_len=max_seq)
        model = pkg.SynthModel(cfg, mix=args.mix)
        lo, hi = pkg.pipeline.split_layers(cfg.num_layers, world)[rank]
        eng = pkg.HipGpuInference.from_model(model, max_seq, device=local, layer_range=(lo, hi), attn_splits=args.attn_splits,
                                             attn_direct=args.attn_direct, flags=args.flags)
        stage = pkg.pipeline.HipStage(eng, torch, dev)
        vocab = cfg.vocab_size
        red = lambda v, op: (lambda t: (dist.all_reduce(t, op=op), float(t.item()))[1])(
            torch.tensor([v], device="cpu" if one_gpu else dev, dtype=torch.float64))
    staged = (not fake) and one_gpu
    dec = pkg.pipeline.PipelineDecoder(stage, rank, world, pkg.pipeline.HostStagedComm(dist) if staged else pkg.pipeline.TorchComm(dist))
    prompt = prompt_tokens(args.prompt, vocab)
    prefill = None
    if not fake:
        # blocks of hidden vectors per hop only when EVERY stage has the batched prompt path (the hop protocol must agree)
        agree = red(1.0 if getattr(stage, "block_tokens", 0) else 0.0, dist.ReduceOp.MIN)
        if agree > 0:
            dec.prefill(prompt[:-1])                       # first use allocates the scratch on every stage
            eng.reset()
            dist.barrier()
            sync()
            t0 = time.perf_counter()
            dec.prefill(prompt[:-1])                       # the batched prompt pass: one [n][hidden] block per hop
            sync()
            dist.barrier()
            bat_s = time.perf_counter() - t0
            prefill = {"tokens": len(prompt) - 1, "batched": True, "forward_batch_ms": round(1e3 * bat_s, 3),
                       "forward_batch_tokens_per_s": round((len(prompt) - 1) / bat_s, 1)}
            eng.reset()
        # the decode that is timed starts from the exact (f32, token by token) cache, as on one GPU
        stage.block_tokens = 0
    dec.prefill(prompt[:-1])
    dec.decode_device(prompt[-1], 1 + W, collect=False)   # the prompt's last token + W warm-up steps; the next token stays on device
    kv0 = stage.position()
    rep_s = []
    for r in range(reps):                                  # each repetition: exactly K steps, barrier + device sync on both sides
        if r > 0 and not fake:
            eng.kv_truncate(kv0)                           # ... from the same cache state on every stage (the fed-back token stays on device)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        dec.decode_device(None, K, collect=False)
        sync()
        dist.barrier()
        rep_s.append(red(time.perf_counter() - t0, dist.ReduceOp.MAX))   # the slowest rank's clock
    kv1 = stage.position()
    elapsed = sum(rep_s) / len(rep_s)
    roofline = None
    P = 0 if fake else min(args.profile_steps, 8)
    if P > 0:   # eager pass with a hipEvent pair around every launch; every rank takes part, rank 0 reports its stage
        eng.set_profiling(True)
        dec.decode_device(None, P, collect=False)
        eng.set_profiling(False)
        sync()
        dist.barrier()
        if rank == 0:
            roofline = roofline_from_stats(eng.stats(), P, None)
            roofline["note"] = f"stage 0 of {world} (layers {lo}..{hi - 1}); kernel-only time, no dispatch gap added"
    if rank == 0:
        tok_s = K / elapsed
        name = "fake-stage" if fake else f"{args.model} {args.mix}"
        step_bytes = 0 if fake else model.step_alg_bytes(int(round((kv0 + 1 + kv0 + K) / 2)))
        print(json.dumps({
            "metric": f"decode tokens/sec {name}, {world} GPUs (layer pipeline); % of HBM roofline",
            "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "none (protocol rehearsal)" if fake else DTYPE_LABEL,
            "data": "fake stage on CPU: launcher + hop protocol rehearsal, NOT a measurement" if fake
                    else "synthetic; every rank on ONE device, hops staged through the host: protocol rehearsal, NOT a measurement" if staged else "synthetic",
            "repetitions": {"n": len(rep_s), "ms_per_step": [round(1e3 * r / K, 4) for r in rep_s],
                            "min_ms_per_step": round(1e3 * min(rep_s) / K, 4), "best_value": round(K / min(rep_s), 2)},
            "config": {"workload": f"{name} single-stream greedy decode, kv_len {kv0 + 1}..{kv1}",
                       "parallelism": f"pp{world} (contiguous layer ranges; per token one f32[hidden] hop per stage boundary and the 4-byte "
                                      f"token fed back device to device, RCCL send/recv over xGMI; no collective)",
                       "prompt_tokens": args.prompt},
            "hbm_roofline": None if fake else {"alg_bytes_per_token": int(step_bytes), "achieved_GBps": round(step_bytes * tok_s / 1e9, 1),
                             "peak_GBps_one_gpu": HBM_PEAK_GBPS, "frac_of_one_gpu": round(step_bytes * tok_s / 1e9 / HBM_PEAK_GBPS, 4)},
            "roofline": roofline, "cpu_baseline": None, "prefill": prefill,
        }))
    if eng is not None:
        eng.close()
    dist.destroy_process_group()


def run_inlib(args, pkg):
    """The layer pipeline inside the library, one process: `--gpus` devices, `--stages` stage contexts (default one per device)."""
    import torch
    hb = pkg.hip_backend
    n_dev = hb.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no HIP device visible (the engine has no CPU fallback)")
    stages = args.stages or args.gpus
    devices = [min(s * args.gpus // stages, n_dev - 1) for s in range(stages)]
    W, K = args.warmup, args.steps
    reps = max(args.reps, 1)
    max_seq = max(512, args.prompt + W + K + 16)
    cfg = pkg.make_config(args.model, max_seq_len=max_seq)
    model = pkg.SynthModel(cfg, mix=args.mix)
    eng = pkg.HipPipeline.from_model(model, max_seq, stages, devices=devices, flags=args.flags)
    prompt = prompt_tokens(args.prompt, cfg.vocab_size)
    for t in prompt[:-1]:
        eng.prefill_token(t)
    tok0 = int(eng.decode_greedy(prompt[-1], 1 + W)[-1])   # the prompt's last token + W warm-up steps (as the other modes)
    kv0 = eng.position()
    rep_s = []
    for _ in range(reps):
        eng.kv_truncate(kv0)                               # every repetition: the same K steps from the same cache state
        tok = tok0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok = int(eng.decode_greedy(tok, K)[-1])       # returns after every stage's stream has drained
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t0)
    kv1 = eng.position()
    elapsed = sum(rep_s) / len(rep_s)
    tok_s = K / elapsed
    step_bytes = model.step_alg_bytes(int(round((kv0 + 1 + kv0 + K) / 2)))
    eng.close()
    print(json.dumps({
        "metric": f"decode tokens/sec {args.model} {args.mix}, {len(set(devices))} GPUs (in-library layer pipeline, {stages} stages); % of HBM roofline",
        "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": len(set(devices)), "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": DTYPE_LABEL, "data": "synthetic",
        "repetitions": {"n": len(rep_s), "ms_per_step": [round(1e3 * r / K, 4) for r in rep_s], "min_ms_per_step": round(1e3 * min(rep_s) / K, 4),
                        "best_value": round(K / min(rep_s), 2)},
        "config": {"workload": f"{args.model} {args.mix} single-stream greedy decode, kv_len {kv0 + 1}..{kv1}",
                   "parallelism": f"pp{stages} in one process on devices {devices}: per token one f32[hidden] peer copy per stage boundary and "
                                  f"the 4-byte token fed back device to device (hipMemcpyPeerAsync + events); no collective",
                   "prompt_tokens": args.prompt},
        "hbm_roofline": {"alg_bytes_per_token": int(step_bytes), "achieved_GBps": round(step_bytes * tok_s / 1e9, 1),
                         "peak_GBps_one_gpu": HBM_PEAK_GBPS, "frac_of_one_gpu": round(step_bytes * tok_s / 1e9 / HBM_PEAK_GBPS, 4)},
        "roofline": None, "cpu_baseline": None,
    }))


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: this parent — which has touched no GPU, loaded no HIP
    library and imported no torch — starts `python -m torch.distributed.run` with N ranks of this same script as a child
    process (never an exec), relays rank 0's single JSON line and returns the child's exit status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: the {args.gpus}-rank child run failed with status {proc.returncode}", file=sys.stderr)
        return proc.returncode
    if line is None:
        print("bench.py: the child run printed no result line", file=sys.stderr)
        return 1
    print(line)
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and not os.environ.get("LGH_BENCH_FORCE_PIPELINE") and not args.inlib:
        sys.exit(self_launch(args))
    # ONE JSON line on stdout, nothing else: native libraries write banners to fd 1 (RCCL prints its version block there
    # at communicator creation), so fd 1 is pointed at stderr for the whole run and the result goes to the saved descriptor.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    out = sys.stdout
    sys.stdout = os.fdopen(saved_stdout, "w", buffering=1)
    del out
    pkg = graft.load_package()
    from importlib import import_module
    pkg.pipeline = import_module("llama_gguf_amd.pipeline")
    # LGH_BENCH_FORCE_PIPELINE=1: run the N>1 code path with however many ranks there are (a 1-rank rehearsal on a 1-GPU box)
    if args.batch:
        run_batch(args, pkg)
    elif args.inlib:
        run_inlib(args, pkg)
    elif args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("LGH_BENCH_FORCE_PIPELINE"):
        run_pipeline(args, pkg)
    else:
        run_single(args, pkg)


if __name__ == "__main__":
    main()
