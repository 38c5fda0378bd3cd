"""Synthetic Llama-family models (random-init, valid GGUF payloads) for parity tests and benchmarks.

No model file exists in this environment and none may be fetched, so every measured or tested model
is generated: shapes follow the reference's ModelConfig presets (src/model/config.rs:239-281 for
Llama-3-8B) and GGUF conventions (a linear weight is [in_features, out_features], dim 0
fastest; MoE expert stacks are [in, out, n_expert]); payload bytes come from the C generator behind
include/llama_gguf_synth.h.  The quantization "mixes" model llama.cpp's `_M` recipes as defined in
SURVEY.md §8(d): all 2-D weights in the base type, except attn_v / ffn_down on layers where
``i < L/8 or i >= 7L/8 or (i - L/8) % 3 == 2`` and output.weight, which are Q6_K.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Iterator, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SYNTH_LIB = os.path.join(_HERE, "lib", "libllama_gguf_synth.so")

F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0 = 0, 1, 2, 3, 6, 7, 8
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K = 10, 11, 12, 13, 14
TYPE_IDS = {"F32": F32, "F16": F16, "Q4_0": Q4_0, "Q4_1": Q4_1, "Q5_0": Q5_0, "Q5_1": Q5_1, "Q8_0": Q8_0,
            "Q2_K": Q2_K, "Q3_K": Q3_K, "Q4_K": Q4_K, "Q5_K": Q5_K, "Q6_K": Q6_K}

_lib = None


def synth_lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SYNTH_LIB):
            raise RuntimeError(f"{_SYNTH_LIB} is missing: run `make -C {_HERE}` (or __graft_entry__.build())")
        L = C.CDLL(_SYNTH_LIB)
        L.lgs_tensor_nbytes.restype = C.c_size_t
        L.lgs_tensor_nbytes.argtypes = [C.c_uint32, C.c_uint64]
        L.lgs_fill_tensor.restype = C.c_int
        L.lgs_fill_tensor.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p,
                                      C.c_size_t, C.c_int]
        _lib = L
    return _lib


def tensor_nbytes(ggml_type: int, n_elems: int) -> int:
    return synth_lib().lgs_tensor_nbytes(ggml_type, n_elems)


def fill_tensor(name: str, ggml_type: int, n_elems: int, in_features: int, kind: int = 0, seed: int = 0,
                threads: int = 0) -> np.ndarray:
    nb = tensor_nbytes(ggml_type, n_elems)
    if nb == 0:
        raise ValueError(f"cannot size tensor {name}: type {ggml_type}, {n_elems} elements")
    out = np.empty(nb, dtype=np.uint8)
    if synth_lib().lgs_fill_tensor(name.encode(), ggml_type, n_elems, in_features, kind, seed, out.ctypes.data, nb, threads):
        raise ValueError(f"lgs_fill_tensor failed for {name}")
    return out


@dataclass
class ModelConfig:
    """The fields of the reference's ModelConfig that the decode path reads (src/model/config.rs)."""
    name: str
    hidden_size: int
    intermediate_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    vocab_size: int
    max_seq_len: int = 512
    norm_eps: float = 1e-5
    rope_freq_base: float = 10000.0
    rope_freq_scale: float = 1.0
    use_neox_rope: bool = False
    num_experts: int = 0
    num_experts_per_token: int = 0
    expert_intermediate_size: int = 0
    tie_embeddings: bool = False

    def as_dict(self) -> dict:
        d = dict(self.__dict__)
        d["use_neox_rope"] = int(self.use_neox_rope)
        return d


# Shapes of the BASELINE.json configs (SURVEY.md §8 header) and small test models.
CONFIGS = {
    "tinyllama-1.1b": dict(hidden_size=2048, intermediate_size=5632, num_layers=22, num_heads=32, num_kv_heads=4,
                           head_dim=64, vocab_size=32000, rope_freq_base=10000.0),
    "llama-3-8b": dict(hidden_size=4096, intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8,
                       head_dim=128, vocab_size=128256, rope_freq_base=500000.0),
    "mixtral-8x7b": dict(hidden_size=4096, intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8,
                         head_dim=128, vocab_size=32000, rope_freq_base=1000000.0, num_experts=8,
                         num_experts_per_token=2, expert_intermediate_size=14336),
    "llama-3-70b": dict(hidden_size=8192, intermediate_size=28672, num_layers=80, num_heads=64, num_kv_heads=8,
                        head_dim=128, vocab_size=128256, rope_freq_base=500000.0),
    # small models for parity tests (the oracle finishes in well under a second)
    "test-dense": dict(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=2,
                       head_dim=64, vocab_size=1024, rope_freq_base=10000.0),
    "test-dense-d128": dict(hidden_size=1024, intermediate_size=2816, num_layers=3, num_heads=8, num_kv_heads=2,
                            head_dim=128, vocab_size=2048, rope_freq_base=500000.0),
    "test-moe": dict(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=4,
                     head_dim=64, vocab_size=1024, rope_freq_base=10000.0, num_experts=4, num_experts_per_token=2,
                     expert_intermediate_size=768),
}


def make_config(name: str, max_seq_len: int = 512, **overrides) -> ModelConfig:
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    return ModelConfig(name=name, max_seq_len=max_seq_len, **kw)


def _is_more_bits_layer(i: int, n_layers: int) -> bool:
    """layers whose attn_v / ffn_down are stored in Q6_K in the `_M` mixes."""
    e = n_layers // 8
    return i < e or i >= 7 * n_layers // 8 or (i - e) % 3 == 2


def mix_type(mix: str, role: str, layer: int, n_layers: int) -> int:
    """GGUF type of a 2-D weight under quantization mix `mix` ("Q4_K_M", "Q5_K_M", or a plain type name)."""
    if mix.endswith("_M"):
        base = TYPE_IDS[mix[:-2]]
        if role == "output":
            return Q6_K
        if role in ("attn_v", "ffn_down") and _is_more_bits_layer(layer, n_layers):
            return Q6_K
        return base
    return TYPE_IDS[mix]


@dataclass
class TensorSpec:
    name: str
    ggml_type: int
    ne: Tuple[int, ...]
    kind: int = 0  # 0 weight, 1 norm, 2 bias

    @property
    def n_elems(self) -> int:
        n = 1
        for v in self.ne:
            n *= v
        return n

    @property
    def nbytes(self) -> int:
        return tensor_nbytes(self.ggml_type, self.n_elems)


@dataclass
class SynthModel:
    """Stands in for the reference's loaded `LlamaModel` (what `from_model` consumes, llama.rs:138-160):
    a config plus named tensors with host byte payloads, generated on demand."""
    config: ModelConfig
    mix: str = "Q4_K_M"
    seed: int = 0x9E3779B97F4A7C15
    with_bias: bool = False
    threads: int = 0
    _cache: dict = field(default_factory=dict, repr=False)

    def specs(self, layers: Optional[range] = None) -> Iterator[TensorSpec]:
        c = self.config
        L = c.num_layers
        H, QD, KD = c.hidden_size, c.num_heads * c.head_dim, c.num_kv_heads * c.head_dim
        t = lambda role, i=0: mix_type(self.mix, role, i, L)
        yield TensorSpec("token_embd.weight", t("embd"), (H, c.vocab_size))
        for i in (layers if layers is not None else range(L)):
            p = f"blk.{i}."
            yield TensorSpec(p + "attn_norm.weight", F32, (H,), 1)
            yield TensorSpec(p + "attn_q.weight", t("attn_q", i), (H, QD))
            yield TensorSpec(p + "attn_k.weight", t("attn_k", i), (H, KD))
            yield TensorSpec(p + "attn_v.weight", t("attn_v", i), (H, KD))
            yield TensorSpec(p + "attn_output.weight", t("attn_output", i), (QD, H))
            if self.with_bias:
                yield TensorSpec(p + "attn_q.bias", F32, (QD,), 2)
                yield TensorSpec(p + "attn_k.bias", F32, (KD,), 2)
                yield TensorSpec(p + "attn_v.bias", F32, (KD,), 2)
            yield TensorSpec(p + "ffn_norm.weight", F32, (H,), 1)
            if c.num_experts:
                EI = c.expert_intermediate_size or c.intermediate_size
                yield TensorSpec(p + "ffn_gate_inp.weight", F32, (H, c.num_experts))
                yield TensorSpec(p + "ffn_gate_exps.weight", t("ffn_gate", i), (H, EI, c.num_experts))
                yield TensorSpec(p + "ffn_up_exps.weight", t("ffn_up", i), (H, EI, c.num_experts))
                yield TensorSpec(p + "ffn_down_exps.weight", t("ffn_down", i), (EI, H, c.num_experts))
            else:
                yield TensorSpec(p + "ffn_gate.weight", t("ffn_gate", i), (H, c.intermediate_size))
                yield TensorSpec(p + "ffn_up.weight", t("ffn_up", i), (H, c.intermediate_size))
                yield TensorSpec(p + "ffn_down.weight", t("ffn_down", i), (c.intermediate_size, H))
        yield TensorSpec("output_norm.weight", F32, (H,), 1)
        if not c.tie_embeddings:
            yield TensorSpec("output.weight", t("output"), (H, c.vocab_size))

    def payload(self, spec: TensorSpec, keep: bool = False) -> np.ndarray:
        if spec.name in self._cache:
            return self._cache[spec.name]
        data = fill_tensor(spec.name, spec.ggml_type, spec.n_elems, spec.ne[0], spec.kind, self.seed, self.threads)
        if keep:
            self._cache[spec.name] = data
        return data

    def tensors(self, layers: Optional[range] = None, keep: bool = False):
        """Yields (name, ggml_type, ne, bytes) — the hand-over of `upload_model_weights`
        (src/backend/cuda/dequant_weights.rs:244-505)."""
        for s in self.specs(layers):
            yield s.name, s.ggml_type, s.ne, self.payload(s, keep)

    def weight_bytes(self) -> int:
        return sum(s.nbytes for s in self.specs())

    def step_alg_bytes(self, kv_len: int) -> int:
        """Algorithmic HBM bytes of one decode step: every weight touched once (top-k experts
        only), one embedding row, KV read at `kv_len` + write, norm vectors, router, logits write."""
        c = self.config
        total = 0
        for s in self.specs():
            if s.name == "token_embd.weight":
                total += s.nbytes // c.vocab_size
                if c.tie_embeddings:
                    total += s.nbytes
            elif "_exps." in s.name:
                total += s.nbytes // c.num_experts * c.num_experts_per_token
            else:
                total += s.nbytes
        total += c.num_layers * (2 * c.num_kv_heads * kv_len * c.head_dim * 4 + 2 * c.num_kv_heads * c.head_dim * 4)
        total += c.vocab_size * 4
        return total
