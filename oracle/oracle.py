"""ctypes binding of the CPU parity oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg; the product package never imports this module (see oracle/oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

# ggml type ids (reference: src/gguf/constants.rs:92-126)
F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K, BF16 = 10, 11, 12, 13, 14, 15, 30
ISA_AUTO, ISA_SCALAR, ISA_AVX2, ISA_AVX512 = 0, 1, 2, 3

QUANT_TYPES = (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K)
TYPE_NAMES = {F32: "F32", F16: "F16", Q4_0: "Q4_0", Q4_1: "Q4_1", Q5_0: "Q5_0", Q5_1: "Q5_1", Q8_0: "Q8_0",
              Q8_1: "Q8_1", Q2_K: "Q2_K", Q3_K: "Q3_K", Q4_K: "Q4_K", Q5_K: "Q5_K", Q6_K: "Q6_K", Q8_K: "Q8_K",
              BF16: "BF16"}


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with the committed Makefile (g++, seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("quant.cpp", "ops.cpp", "model.cpp", "oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


class Config(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "hidden_size", "intermediate_size", "num_layers", "num_heads", "num_kv_heads", "head_dim",
        "vocab_size", "max_seq_len", "num_experts", "num_experts_per_token", "expert_intermediate_size",
        "use_neox_rope")] + [(n, C.c_float) for n in ("norm_eps", "rope_freq_base", "rope_freq_scale")]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    f32p, u32p, vp, sz = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t
    sig = {
        "orc_set_isa": (None, [C.c_int]), "orc_get_isa": (C.c_int, []),
        "orc_set_threads": (None, [C.c_int]), "orc_get_threads": (C.c_int, []),
        "orc_block_size": (sz, [C.c_int]), "orc_block_bytes": (sz, [C.c_int]),
        "orc_f32_to_f16": (C.c_uint16, [C.c_float]), "orc_f16_to_f32": (C.c_float, [C.c_uint16]),
        "orc_quantize": (C.c_int, [C.c_int, vp, sz, vp]), "orc_dequantize": (C.c_int, [C.c_int, vp, sz, vp]),
        "orc_dot_q": (C.c_float, [C.c_int, vp, vp, sz]), "orc_has_fused_dot": (C.c_int, [C.c_int]),
        "orc_dot_f32": (C.c_float, [vp, vp, sz]),
        "orc_vec_mat_q": (C.c_int, [C.c_int, vp, vp, vp, sz, sz]),
        "orc_vec_mat_f32": (None, [vp, vp, vp, sz, sz]),
        "orc_rms_norm": (None, [vp, vp, C.c_float, vp, sz]),
        "orc_rope": (None, [vp, vp, sz, sz, sz, sz, sz, C.c_float, C.c_float, C.c_int]),
        "orc_attention_cached": (None, [vp, vp, vp, vp, sz, sz, sz, sz, C.c_float, sz]),
        "orc_softmax_inplace": (None, [vp, sz]), "orc_silu": (None, [vp, vp, sz]),
        "orc_add": (None, [vp, vp, vp, sz]), "orc_mul": (None, [vp, vp, vp, sz]), "orc_scale": (None, [vp, C.c_float, vp, sz]),
        "orc_gelu": (None, [vp, vp, sz]), "orc_softmax_rows": (None, [vp, vp, sz, sz]),
        "orc_matmul": (None, [vp, vp, vp, sz, sz, sz]), "orc_matvec": (None, [vp, vp, vp, sz, sz]),
        "orc_attention": (None, [vp, vp, vp, vp, sz, sz, sz, sz, sz, C.c_float]),
        "orc_silu_mul_inplace": (None, [vp, vp, sz]), "orc_max_f32": (C.c_float, [vp, sz]), "orc_sum_f32": (C.c_float, [vp, sz]),
        "orc_kv_quantize_int8": (None, [vp, sz, vp, vp]), "orc_kv_dequantize_int8": (None, [vp, C.c_float, sz, vp]),
        "orc_model_set_kv_int8": (None, [vp, C.c_int]), "orc_model_set_kv_fp8": (None, [vp, C.c_int]),
        "orc_kv_quantize_fp8": (C.c_uint8, [C.c_int, C.c_float]), "orc_kv_dequantize_fp8": (C.c_float, [C.c_int, C.c_uint8]),
        "orc_axpy_f32": (None, [C.c_float, vp, vp, sz]),
        "orc_tq_padded_dim": (sz, [sz]), "orc_tq_codebook": (C.c_int, [sz, C.c_int, vp, vp]),
        "orc_tq_quantize": (C.c_uint8, [vp, C.c_int, C.c_float]), "orc_tq_packed_bytes": (sz, [C.c_int, sz]),
        "orc_tq_quantize_vector": (None, [sz, C.c_int, vp, sz, vp]), "orc_tq_dequantize_vector": (None, [sz, C.c_int, vp, sz, vp]),
        "orc_tq_dot_with_packed": (C.c_float, [sz, C.c_int, vp, vp, sz]),
        "orc_tq_rotate": (None, [vp, sz, vp, vp]), "orc_tq_rotate_inverse": (None, [vp, sz, vp, vp]),
        "orc_tq_compress": (None, [vp, sz, C.c_int, vp, vp]),
        "orc_tq_attention_head": (None, [vp, vp, vp, sz, sz, C.c_int, vp, vp, C.c_float, vp]),
        "orc_model_set_kv_turboquant": (C.c_int, [vp, C.c_int, vp, sz]),
        "orc_tq_dot_with_sign_bits": (C.c_float, [vp, vp, sz]), "orc_tq_qjl_project": (None, [vp, sz, vp, vp]),
        "orc_tq_qjl_compress": (None, [vp, sz, vp, vp, vp]), "orc_tq_qjl_inner_product_fast": (C.c_float, [sz, vp, vp, C.c_float]),
        "orc_tq_bytes_per_entry": (sz, [sz, C.c_int, C.c_int]),
        "orc_tq_compress_qjl": (None, [vp, sz, C.c_int, vp, vp, vp, vp, vp]),
        "orc_tq_attention_head_qjl": (None, [vp, vp, vp, vp, vp, sz, sz, C.c_int, vp, vp, vp, C.c_float, vp]),
        "orc_model_set_kv_turboquant_qjl": (C.c_int, [vp, vp, sz]),
        "orc_argmax_last": (C.c_uint32, [vp, sz]), "orc_greedy_sample": (C.c_uint32, [vp, sz]),
        "orc_moe_route": (None, [vp, vp, sz, sz, sz, C.c_int, vp, vp]),
        "orc_model_create": (vp, [C.POINTER(Config)]), "orc_model_destroy": (None, [vp]),
        "orc_model_add_tensor": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), vp, sz, C.c_int]),
        "orc_model_finalize": (C.c_int, [vp]), "orc_model_last_error": (C.c_char_p, [vp]),
        "orc_model_forward": (C.c_int, [vp, vp, sz, vp, C.c_int]),
        "orc_model_reset": (None, [vp]), "orc_model_position": (sz, [vp]),
        "orc_model_kv_truncate": (None, [vp, sz]), "orc_model_kv_shift_left": (None, [vp, sz]),
        "orc_model_last_hidden": (C.c_int, [vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def block_size(t: int) -> int:
    return lib().orc_block_size(t)


def block_bytes(t: int) -> int:
    return lib().orc_block_bytes(t)


def nbytes_for(t: int, n_elems: int) -> int:
    return n_elems // block_size(t) * block_bytes(t)


def set_isa(isa: int) -> None:
    lib().orc_set_isa(isa)


def get_isa() -> int:
    return lib().orc_get_isa()


def set_threads(n: int) -> None:
    lib().orc_set_threads(n)


def quantize(t: int, x) -> np.ndarray:
    x = _f32(x).ravel()
    out = np.zeros(nbytes_for(t, x.size), dtype=np.uint8)
    if lib().orc_quantize(t, _p(x), x.size, _p(out)):
        raise ValueError("orc_quantize failed")
    return out


def dequantize(t: int, raw: np.ndarray, n_elems: int) -> np.ndarray:
    raw = np.ascontiguousarray(raw)
    out = np.empty(n_elems, dtype=np.float32)
    if lib().orc_dequantize(t, _p(raw), n_elems, _p(out)):
        raise ValueError("orc_dequantize failed")
    return out


def dot_q(t: int, raw: np.ndarray, x) -> float:
    x = _f32(x)
    raw = np.ascontiguousarray(raw)
    return float(lib().orc_dot_q(t, _p(raw), _p(x), x.size))


def dot_f32(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot_f32(_p(a), _p(b), a.size))


def vec_mat_q(t: int, raw: np.ndarray, x, n: int) -> np.ndarray:
    x = _f32(x)
    raw = np.ascontiguousarray(raw)
    out = np.empty(n, dtype=np.float32)
    if lib().orc_vec_mat_q(t, _p(raw), _p(x), _p(out), x.size, n):
        raise ValueError("orc_vec_mat_q failed")
    return out


def vec_mat_f32(w, x, n: int) -> np.ndarray:
    w, x = _f32(w).ravel(), _f32(x)
    out = np.empty(n, dtype=np.float32)
    lib().orc_vec_mat_f32(_p(w), _p(x), _p(out), x.size, n)
    return out


def rms_norm(x, w, eps: float) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    out = np.empty_like(x)
    lib().orc_rms_norm(_p(x), _p(w), eps, _p(out), x.size)
    return out


def rope(q, k, pos: int, freq_base: float, freq_scale: float, neox: bool):
    """q: [n_heads, seq, d], k: [n_kv, seq, d]; returns rotated copies."""
    q, k = _f32(q).copy(), _f32(k).copy()
    lib().orc_rope(_p(q), _p(k), q.shape[0], k.shape[0], q.shape[1], q.shape[2], pos, freq_base, freq_scale,
                   int(neox))
    return q, k


def attention_cached(q, k_cache, v_cache, scale: float, kv_len: int) -> np.ndarray:
    """q: [n_heads, d]; caches: [n_kv, max_seq, d]."""
    q, k_cache, v_cache = _f32(q), _f32(k_cache), _f32(v_cache)
    out = np.empty_like(q)
    lib().orc_attention_cached(_p(q), _p(k_cache), _p(v_cache), _p(out), q.shape[0], k_cache.shape[0], q.shape[1],
                               k_cache.shape[1], scale, kv_len)
    return out


def softmax(x) -> np.ndarray:
    x = _f32(x).copy()
    lib().orc_softmax_inplace(_p(x), x.size)
    return x


def add(a, b) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_add(_p(a), _p(b), _p(out), a.size)
    return out


def mul(a, b) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_mul(_p(a), _p(b), _p(out), a.size)
    return out


def scale(a, s: float) -> np.ndarray:
    a = _f32(a)
    out = np.empty_like(a)
    lib().orc_scale(_p(a), s, _p(out), a.size)
    return out


def gelu(x) -> np.ndarray:
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_gelu(_p(x), _p(out), x.size)
    return out


def softmax_rows(x) -> np.ndarray:
    """Backend::softmax: along the last dimension (ops.rs:350-385)."""
    x = _f32(x)
    out = np.empty_like(x)
    last = x.shape[-1] if x.ndim else 1
    lib().orc_softmax_rows(_p(x), _p(out), x.size // max(last, 1), last)
    return out


def matmul(a, b) -> np.ndarray:
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[1]), dtype=np.float32)
    lib().orc_matmul(_p(a), _p(b), _p(out), a.shape[0], a.shape[1], b.shape[1])
    return out


def matvec(a, x) -> np.ndarray:
    a, x = _f32(a), _f32(x)
    out = np.empty(a.shape[0], dtype=np.float32)
    lib().orc_matvec(_p(a), _p(x), _p(out), a.shape[0], a.shape[1])
    return out


def attention(q, k, v, scale_: float) -> np.ndarray:
    """Backend::attention: q [heads, seq, d], k / v [kv_heads, kv_len, d], causal (ops.rs:1353-1472)."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    out = np.empty_like(q)
    lib().orc_attention(_p(q), _p(k), _p(v), _p(out), q.shape[0], k.shape[0], q.shape[1], k.shape[1], q.shape[2], scale_)
    return out


def silu(x) -> np.ndarray:
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_silu(_p(x), _p(out), x.size)
    return out


def silu_mul(gate, up) -> np.ndarray:
    g, u = _f32(gate).copy(), _f32(up)
    lib().orc_silu_mul_inplace(_p(g), _p(u), g.size)
    return g


def kv_quantize_int8(x):
    """quantize_int8 (kv_quantized.rs:385-405): (int8 values, scale)."""
    x = _f32(x)
    q = np.zeros(x.size, dtype=np.int8)
    sc = C.c_float(0.0)
    lib().orc_kv_quantize_int8(_p(x), x.size, _p(q), C.byref(sc))
    return q, float(sc.value)


def kv_dequantize_int8(q, scale: float) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.int8)
    out = np.zeros(q.size, dtype=np.float32)
    lib().orc_kv_dequantize_int8(_p(q), C.c_float(scale), q.size, _p(out))
    return out


FP8_E4M3, FP8_E5M2 = 1, 2


# ---- TurboQuant (oracle/turboquant.cpp; src/model/turboquant/*.rs, src/model/kv_turboquant.rs)
def tq_padded_dim(dim: int) -> int:
    return int(lib().orc_tq_padded_dim(dim))


def tq_codebook(dim: int, bits: int):
    """Codebook::new (codebook.rs:55-77): (centroids, boundaries), scaled by 1 / sqrt(dim)."""
    c, b = np.zeros(1 << bits, np.float32), np.zeros((1 << bits) - 1, np.float32)
    assert lib().orc_tq_codebook(dim, bits, _p(c), _p(b)) == 0
    return c, b


def tq_quantize(dim: int, bits: int, val: float) -> int:
    _, b = tq_codebook(dim, bits)
    return int(lib().orc_tq_quantize(_p(b), bits, C.c_float(val)))


def tq_packed_bytes(bits: int, count: int) -> int:
    return int(lib().orc_tq_packed_bytes(bits, count))


def tq_quantize_vector(dim: int, bits: int, data) -> np.ndarray:
    data = _f32(data)
    out = np.zeros(tq_packed_bytes(bits, data.size), np.uint8)
    lib().orc_tq_quantize_vector(dim, bits, _p(data), data.size, _p(out))
    return out


def tq_dequantize_vector(dim: int, bits: int, packed, count: int) -> np.ndarray:
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.zeros(count, np.float32)
    lib().orc_tq_dequantize_vector(dim, bits, _p(packed), count, _p(out))
    return out


def tq_dot_with_packed(dim: int, bits: int, query, packed, count: int) -> float:
    query, packed = _f32(query), np.ascontiguousarray(packed, dtype=np.uint8)
    return float(lib().orc_tq_dot_with_packed(dim, bits, _p(query), _p(packed), count))


def tq_rotate(x, signs) -> np.ndarray:
    x, signs = _f32(x), _f32(signs)
    out = np.zeros(tq_padded_dim(x.size), np.float32)
    assert signs.size == out.size
    lib().orc_tq_rotate(_p(x), x.size, _p(signs), _p(out))
    return out


def tq_rotate_inverse(x, dim: int, signs) -> np.ndarray:
    x, signs = _f32(x), _f32(signs)
    out = np.zeros(dim, np.float32)
    lib().orc_tq_rotate_inverse(_p(x), dim, _p(signs), _p(out))
    return out


def tq_compress(x, bits: int, signs) -> np.ndarray:
    x, signs = _f32(x), _f32(signs)
    out = np.zeros(tq_packed_bytes(bits, tq_padded_dim(x.size)), np.uint8)
    lib().orc_tq_compress(_p(x), x.size, bits, _p(signs), _p(out))
    return out


def tq_dot_with_sign_bits(values, bits, count: int) -> float:
    values, bits = _f32(values), np.ascontiguousarray(bits, dtype=np.uint64)
    return float(lib().orc_tq_dot_with_sign_bits(_p(values), _p(bits), count))


def tq_qjl_project(S, q) -> np.ndarray:
    S, q = _f32(S), _f32(q)
    out = np.zeros(q.size, np.float32)
    lib().orc_tq_qjl_project(_p(S), q.size, _p(q), _p(out))
    return out


def tq_qjl_compress(S, x):
    """-> (bits uint64[(dim + 63) // 64], norm)"""
    S, x = _f32(S), _f32(x)
    bits = np.zeros((x.size + 63) // 64, np.uint64)
    norm = C.c_float(0.0)
    lib().orc_tq_qjl_compress(_p(S), x.size, _p(x), _p(bits), C.byref(norm))
    return bits, float(norm.value)


def tq_qjl_inner_product_fast(projected_query, bits, norm: float) -> float:
    pq, bits = _f32(projected_query), np.ascontiguousarray(bits, dtype=np.uint64)
    return float(lib().orc_tq_qjl_inner_product_fast(pq.size, _p(pq), _p(bits), C.c_float(norm)))


def tq_bytes_per_entry(dim: int, bits: int, use_qjl: bool) -> int:
    return int(lib().orc_tq_bytes_per_entry(dim, bits, int(use_qjl)))


def tq_compress_qjl(x, bits: int, signs, S):
    """TurboQuantEngine::compress with QJL -> (codes uint8[], qjl_bits uint64[], residual_norm)"""
    x, signs, S = _f32(x), _f32(signs), _f32(S)
    pd = tq_padded_dim(x.size)
    codes = np.zeros(tq_packed_bytes(bits, pd), np.uint8)
    qb = np.zeros((pd + 63) // 64, np.uint64)
    norm = C.c_float(0.0)
    lib().orc_tq_compress_qjl(_p(x), x.size, bits, _p(signs), _p(S), _p(codes), _p(qb), C.byref(norm))
    return codes, qb, float(norm.value)


def tq_attention_head_qjl(query, k_codes, k_qjl, k_norm, v_codes, kv_len: int, bits: int, signs_k, signs_v, S_k, scale: float) -> np.ndarray:
    query = _f32(query)
    k_codes, v_codes = np.ascontiguousarray(k_codes, dtype=np.uint8), np.ascontiguousarray(v_codes, dtype=np.uint8)
    k_qjl, k_norm = np.ascontiguousarray(k_qjl, dtype=np.uint64), _f32(k_norm)
    signs_k, signs_v, S_k = _f32(signs_k), _f32(signs_v), _f32(S_k)
    out = np.zeros(query.size, np.float32)
    lib().orc_tq_attention_head_qjl(_p(query), _p(k_codes), _p(k_qjl), _p(k_norm), _p(v_codes), kv_len, query.size, bits, _p(signs_k), _p(signs_v),
                                    _p(S_k), C.c_float(scale), _p(out))
    return out


def tq_attention_head(query, k_codes, v_codes, kv_len: int, bits: int, signs_k, signs_v, scale: float) -> np.ndarray:
    query = _f32(query)
    k_codes, v_codes = np.ascontiguousarray(k_codes, dtype=np.uint8), np.ascontiguousarray(v_codes, dtype=np.uint8)
    signs_k, signs_v = _f32(signs_k), _f32(signs_v)
    out = np.zeros(query.size, np.float32)
    lib().orc_tq_attention_head(_p(query), _p(k_codes), _p(v_codes), kv_len, query.size, bits, _p(signs_k), _p(signs_v), C.c_float(scale), _p(out))
    return out


def kv_quantize_fp8(fmt: int, x: float) -> int:
    """quantize_fp8_e4m3 / _e5m2 (kv_quantized.rs:413-449, 492-528): the byte."""
    return int(lib().orc_kv_quantize_fp8(fmt, C.c_float(x)))


def kv_dequantize_fp8(fmt: int, b: int) -> float:
    return float(lib().orc_kv_dequantize_fp8(fmt, C.c_uint8(b)))


def argmax_last(v) -> int:
    v = _f32(v)
    return int(lib().orc_argmax_last(_p(v), v.size))


def greedy_sample(v) -> int:
    v = _f32(v)
    return int(lib().orc_greedy_sample(_p(v), v.size))


def moe_route(h, w, n_experts: int, top_k: int, normalize: bool = False):
    h, w = _f32(h), _f32(w).ravel()
    idx = np.zeros(top_k, dtype=np.uint32)
    wt = np.zeros(top_k, dtype=np.float32)
    lib().orc_moe_route(_p(h), _p(w), h.size, n_experts, top_k, int(normalize), _p(idx), _p(wt))
    return idx, wt


class Model:
    """The reference's ``LlamaModel`` + ``InferenceContext`` as one CPU object (llama.rs:275-362)."""

    def __init__(self, cfg: dict):
        self._cfg = Config(**{k: cfg[k] for k, _ in Config._fields_ if k in cfg})
        self._h = lib().orc_model_create(C.byref(self._cfg))
        self._keep = []
        self.vocab_size = cfg["vocab_size"]
        self.hidden_size = cfg["hidden_size"]

    def set_kv_int8(self, on: bool = True) -> None:
        """K/V rows go through the reference's int8 KV format (kv_quantized.rs) on their way into the cache."""
        lib().orc_model_set_kv_int8(self._h, int(on))

    def set_kv_turboquant(self, bits: int, signs) -> None:
        """KVCacheType::TurboQuantMSE { bits }: K/V rows are stored as TurboQuant codes and attention runs over the codes
        (kv_turboquant.rs); `signs` = [layers][kv heads][2][padded head_dim] of +-1."""
        signs = _f32(signs)
        self._keep.append(signs)
        if lib().orc_model_set_kv_turboquant(self._h, int(bits), _p(signs), signs.size):
            raise ValueError("orc_model_set_kv_turboquant: bad bits / sign vector")

    def set_kv_turboquant_qjl(self, qjl) -> None:
        """KVCacheType::TurboQuantProd { bits } on top of set_kv_turboquant: `qjl` = [layers][kv heads][pd][pd], the K engines'
        QJL projection matrices (qjl.rs)."""
        qjl = _f32(qjl)
        self._keep.append(qjl)
        if lib().orc_model_set_kv_turboquant_qjl(self._h, _p(qjl), qjl.size):
            raise ValueError("orc_model_set_kv_turboquant_qjl: call set_kv_turboquant first / wrong matrix size")

    def set_kv_fp8(self, fmt: int) -> None:
        """K/V rows go through one of the reference's FP8 KV formats (FP8_E4M3 / FP8_E5M2; 0 = off)."""
        lib().orc_model_set_kv_fp8(self._h, int(fmt))

    def add_tensor(self, name: str, t: int, ne, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data)
        self._keep.append(data)  # borrowed by the oracle
        ne4 = (C.c_uint64 * 4)(*(list(ne) + [0] * (4 - len(ne))))
        if lib().orc_model_add_tensor(self._h, name.encode(), t, ne4, _p(data), data.nbytes, 1):
            raise ValueError(self.last_error())

    def finalize(self) -> None:
        if lib().orc_model_finalize(self._h):
            raise ValueError(self.last_error())

    def last_error(self) -> str:
        return lib().orc_model_last_error(self._h).decode()

    def forward(self, tokens, faithful_embedding: bool = False) -> np.ndarray:
        toks = np.ascontiguousarray(tokens, dtype=np.uint32)
        logits = np.empty(self.vocab_size, dtype=np.float32)
        if lib().orc_model_forward(self._h, _p(toks), toks.size, _p(logits), int(faithful_embedding)):
            raise ValueError(self.last_error())
        return logits

    def last_hidden(self) -> np.ndarray:
        out = np.empty(self.hidden_size, dtype=np.float32)
        lib().orc_model_last_hidden(self._h, _p(out))
        return out

    def reset(self) -> None:
        lib().orc_model_reset(self._h)

    def kv_truncate(self, new_len: int) -> None:
        lib().orc_model_kv_truncate(self._h, new_len)

    def kv_shift_left(self, amount: int) -> None:
        lib().orc_model_kv_shift_left(self._h, amount)

    @property
    def position(self) -> int:
        return lib().orc_model_position(self._h)

    def close(self) -> None:
        if self._h:
            lib().orc_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
