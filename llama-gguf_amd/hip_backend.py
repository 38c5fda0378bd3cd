"""Host-side mirror of the reference's GPU-engine interface over the C ABI (include/llama_gguf_hip.h).

The reference drives every GPU engine through ``trait GpuInference`` (src/backend/mod.rs:283-296:
``forward / prefill_token / reset / position``), builds it with ``from_model(model, max_seq_len)``
(src/backend/cuda/gpu_only.rs:426) and adapts it to ``trait Model`` with ``GpuModelWrapper``
(src/backend/mod.rs:302-364).  The reference is Rust and this image has no Rust toolchain, so the
Rust shim is shipped as source in INTEGRATION.md; this module is the same surface — same names,
argument meaning and error behaviour — for the tests and the benchmark.  Nothing here computes: every
call goes through ``lib/libllama_gguf_hip.so`` and FAILS LOUDLY if that library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LGH_LIB_VARIANT=stamps selects the diagnostic build (in-kernel phase stamps; tools/microbench.py only)
LIB_PATH = os.path.join(_HERE, "lib", "libllama_gguf_hip%s.so" % ("_" + os.environ["LGH_LIB_VARIANT"] if os.environ.get("LGH_LIB_VARIANT") else ""))

# every symbol include/llama_gguf_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "lgh_device_count", "lgh_create", "lgh_upload_tensor", "lgh_finalize", "lgh_destroy", "lgh_forward",
    "lgh_prefill_token", "lgh_prefill_batch", "lgh_prefill_is_batched", "lgh_op_mat_mat",
    "lgh_op_kv_roundtrip", "lgh_op_add", "lgh_op_mul", "lgh_op_scale", "lgh_op_silu", "lgh_op_gelu", "lgh_op_softmax", "lgh_op_matmul", "lgh_op_matvec",
    "lgh_op_matvec_q", "lgh_op_attention", "lgh_backend_create", "lgh_backend_destroy", "lgh_backend_load_weight",
    "lgh_backend_has_weight", "lgh_backend_vec_mat_q", "lgh_backend_last_error", "lgh_reset", "lgh_position", "lgh_kv_truncate", "lgh_kv_shift_left", "lgh_stage_prefill_batch", "lgh_stage_hidden_block_buffer", "lgh_forward_argmax", "lgh_decode_greedy",
    "lgh_last_error", "lgh_get_stats", "lgh_set_profiling", "lgh_set_stream", "lgh_get_stream", "lgh_synchronize",
    "lgh_read_hidden", "lgh_stage_hidden_buffer", "lgh_stage_forward", "lgh_op_dequantize", "lgh_op_vec_mat",
    "lgh_op_rms_norm", "lgh_op_rope", "lgh_op_attention_cached", "lgh_op_silu_mul", "lgh_op_norm_vec_mat",
    "lgh_op_swiglu_vec_mat", "lgh_bench_vec_mat", "lgh_bench_hbm_read", "lgh_gguf_inspect", "lgh_load_gguf",
    "lgh_stage_io_buffers", "lgh_stage_set_forward_targets", "lgh_stage_step", "lgh_stage_read_tokens", "lgh_gguf_get", "lgh_stage_read_logits",
    "lgh_pipeline_create", "lgh_pipeline_upload_tensor", "lgh_pipeline_finalize", "lgh_pipeline_destroy", "lgh_pipeline_forward",
    "lgh_pipeline_prefill_token", "lgh_pipeline_decode_greedy", "lgh_pipeline_reset", "lgh_pipeline_position", "lgh_pipeline_kv_truncate", "lgh_pipeline_stages",
    "lgh_pipeline_last_error",
    "lgh_set_kv_rotation_signs", "lgh_op_tq_compress", "lgh_set_kv_qjl_matrices", "lgh_op_tq_compress_qjl",
    "lgh_batch_create", "lgh_batch_reset", "lgh_batch_position", "lgh_batch_prefill", "lgh_forward_multi", "lgh_decode_greedy_multi",
)

K_NAMES = ("embed", "qkv", "attn", "attn_combine", "wo", "gate_up", "down", "router", "output", "argmax", "misc", "token")
K_COUNT = 16
SYM_COUNT = 24
# kernel symbols as rocprofv3 prints them (LGH_SYM_* order)
SYM_NAMES = ("lgh::mv_kernel<1u, 1024>", "lgh::mv_kernel<8u, 1024>", "lgh::mv_kernel<16u, 1024>",
             "lgh::mv_kernel<2u, 768>", "lgh::mv_kernel<4u, 512>", "lgh::mv_kernel<5u, 512>", "lgh::mv_kernel<6u, 512>",
             "lgh::mv_kernel<31u, 512>", "lgh::f32_matvec_kernel", "lgh::attn_partial_kernel", "lgh::attn_combine_kernel",
             "lgh::embed_kernel", "lgh::argmax_stage1+2", "lgh::moe_router_kernel", "other", "lgh::mvq_kernel<1u>", "lgh::mvq_kernel<2u>", "lgh::mvq_kernel<3u>", "lgh::mvq_kernel<4u>",
             "lgh::mvq_kernel<8u>", "unused")
FLAG_NO_GRAPH = 1
FLAG_EXACT_PREFILL = 4   # forward_batch feeds tokens one by one (f32 throughout) instead of the batched f16 GEMM path
KV_F32, KV_INT8, KV_FP8_E4M3, KV_FP8_E5M2 = 0, 1, 2, 3   # lgh_model_desc.kv_cache_type: f32, or QuantizedKVCache's formats (kv_quantized.rs; no CLI flag reaches them)
KV_TQ2, KV_TQ3 = 4, 5   # KVCacheType::TurboQuantMSE { bits: 2 | 3 }: what `--kv-cache-type tq2 | tq3` selects (src/config.rs:808-817)
KV_TQ2_QJL, KV_TQ3_QJL = 6, 7   # KVCacheType::TurboQuantProd { bits: 2 | 3 }: `tq2-qjl | tq3-qjl`
FLAG_KV_INT8 = 16   # KV cache in the reference's int8 format (kv_quantized.rs: int8 rows + one scale per head and position)
FLAG_REMOVED_MASK = 2 | 8 | 32 | 64 | 128 | (0xFF << 24)   # round-2 decode experiments, removed in round 3: lgh_create answers Unsupported


class BackendError(RuntimeError):
    """The reference's BackendError (src/backend/error.rs:3-37); `.variant` names the Rust variant."""
    VARIANTS = {1: "NotAvailable", 2: "ShapeMismatch", 3: "DTypeMismatch", 4: "UnsupportedDType", 5: "Unsupported",
                6: "InvalidArgument", 7: "Tensor", 8: "InitializationFailed", 9: "AllocationFailed",
                10: "OperationFailed"}

    def __init__(self, status: int, message: str = ""):
        self.status = status
        self.variant = self.VARIANTS.get(status, f"Status{status}")
        super().__init__(f"{self.variant}: {message}" if message else self.variant)


class ModelDesc(C.Structure):
    _fields_ = ([(n, C.c_uint32) for n in (
        "struct_size", "hidden_size", "intermediate_size", "num_layers", "num_heads", "num_kv_heads", "head_dim",
        "vocab_size", "max_seq_len", "num_experts", "num_experts_per_token", "expert_intermediate_size",
        "use_neox_rope")]
                + [(n, C.c_float) for n in ("norm_eps", "rope_freq_base", "rope_freq_scale")]
                + [("device_id", C.c_int32), ("layer_begin", C.c_uint32), ("layer_end", C.c_uint32),
                   ("flags", C.c_uint32), ("kv_cache_type", C.c_uint32)])


class GgufInfo(C.Structure):
    _fields_ = [("version", C.c_uint32), ("alignment", C.c_uint32), ("n_tensors", C.c_uint64), ("n_kv", C.c_uint64),
                ("data_offset", C.c_uint64), ("file_bytes", C.c_uint64), ("architecture", C.c_char * 64), ("desc", ModelDesc)]


class GgufValue(C.Structure):
    _fields_ = [("type", C.c_uint32), ("reserved", C.c_uint32), ("u", C.c_uint64), ("f", C.c_double), ("arr_len", C.c_uint64),
                ("s", C.c_char * 256)]


class Stats(C.Structure):
    _fields_ = [("weight_bytes", C.c_uint64), ("kv_bytes", C.c_uint64), ("scratch_bytes", C.c_uint64),
                ("tokens_processed", C.c_uint64), ("graph_nodes", C.c_uint64),
                ("k_launches", C.c_uint64 * K_COUNT), ("k_time_us", C.c_double * K_COUNT),
                ("k_alg_bytes", C.c_uint64 * K_COUNT),
                ("sym_launches", C.c_uint64 * SYM_COUNT), ("sym_time_us", C.c_double * SYM_COUNT),
                ("sym_alg_bytes", C.c_uint64 * SYM_COUNT), ("step_alg_bytes", C.c_uint64),
                ("event_bracket_us", C.c_double), ("event_bracket_samples", C.c_uint64), ("overlapped_edges", C.c_uint64)]


_lib = None


def load_library() -> C.CDLL:
    """dlopen the engine.  Raises (never falls back) when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(8, f"{LIB_PATH} is missing: build it with `make -C {_HERE}` or __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_float
    sig = {
        "lgh_device_count": (C.c_int, []),
        "lgh_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(vp)]),
        "lgh_upload_tensor": (C.c_int, [vp, C.c_char_p, u32, C.POINTER(C.c_uint64), vp, sz]),
        "lgh_finalize": (C.c_int, [vp]), "lgh_destroy": (None, [vp]),
        "lgh_forward": (C.c_int, [vp, u32, vp]), "lgh_prefill_token": (C.c_int, [vp, u32]),
        "lgh_prefill_batch": (C.c_int, [vp, vp, sz]), "lgh_prefill_is_batched": (C.c_int, [vp]),
        "lgh_op_mat_mat": (C.c_int, [C.c_int, u32, vp, vp, vp, sz, sz, sz]), "lgh_reset": (None, [vp]), "lgh_position": (sz, [vp]),
        "lgh_kv_truncate": (C.c_int, [vp, sz]), "lgh_kv_shift_left": (C.c_int, [vp, sz]),
        "lgh_stage_prefill_batch": (C.c_int, [vp, vp, sz]), "lgh_stage_hidden_block_buffer": (C.c_int, [vp, C.POINTER(vp)]),
        "lgh_forward_argmax": (C.c_int, [vp, u32, C.POINTER(u32)]),
        "lgh_decode_greedy": (C.c_int, [vp, u32, sz, vp]),
        "lgh_last_error": (C.c_char_p, [vp]), "lgh_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
        "lgh_set_profiling": (C.c_int, [vp, C.c_int]), "lgh_set_stream": (C.c_int, [vp, vp]),
        "lgh_get_stream": (vp, [vp]), "lgh_synchronize": (C.c_int, [vp]), "lgh_read_hidden": (C.c_int, [vp, vp]),
        "lgh_stage_hidden_buffer": (C.c_int, [vp, C.POINTER(vp)]),
        "lgh_stage_forward": (C.c_int, [vp, u32, C.c_int, vp, C.POINTER(u32)]),
        "lgh_op_dequantize": (C.c_int, [C.c_int, u32, vp, sz, vp]),
        "lgh_op_vec_mat": (C.c_int, [C.c_int, u32, vp, vp, vp, sz, sz]),
        "lgh_op_rms_norm": (C.c_int, [C.c_int, vp, vp, f32, vp, sz]),
        "lgh_op_rope": (C.c_int, [C.c_int, vp, vp, sz, sz, sz, sz, f32, f32, C.c_int]),
        "lgh_op_attention_cached": (C.c_int, [C.c_int, vp, vp, vp, vp, sz, sz, sz, sz, f32, sz, C.c_int]),
        "lgh_op_silu_mul": (C.c_int, [C.c_int, vp, vp, vp, sz]),
        "lgh_op_norm_vec_mat": (C.c_int, [C.c_int, u32, vp, vp, vp, f32, vp, sz, sz]),
        "lgh_op_swiglu_vec_mat": (C.c_int, [C.c_int, u32, vp, vp, vp, vp, f32, vp, sz, sz]),
        "lgh_op_add": (C.c_int, [C.c_int, vp, vp, vp, sz]), "lgh_op_mul": (C.c_int, [C.c_int, vp, vp, vp, sz]),
        "lgh_op_kv_roundtrip": (C.c_int, [C.c_int, C.c_uint32, vp, sz, vp, vp, vp]),
        "lgh_op_scale": (C.c_int, [C.c_int, vp, f32, vp, sz]), "lgh_op_silu": (C.c_int, [C.c_int, vp, vp, sz]),
        "lgh_op_gelu": (C.c_int, [C.c_int, vp, vp, sz]), "lgh_op_softmax": (C.c_int, [C.c_int, vp, vp, sz, sz]),
        "lgh_op_matmul": (C.c_int, [C.c_int, vp, vp, vp, sz, sz, sz]), "lgh_op_matvec": (C.c_int, [C.c_int, vp, vp, vp, sz, sz]),
        "lgh_op_matvec_q": (C.c_int, [C.c_int, u32, vp, vp, vp, sz, sz]),
        "lgh_op_attention": (C.c_int, [C.c_int, vp, vp, vp, vp, sz, sz, sz, sz, sz, f32]),
        "lgh_backend_create": (C.c_int, [C.c_int, C.POINTER(vp)]), "lgh_backend_destroy": (None, [vp]),
        "lgh_backend_load_weight": (C.c_int, [vp, C.c_char_p, u32, vp, sz, sz]), "lgh_backend_has_weight": (C.c_int, [vp, C.c_char_p]),
        "lgh_backend_vec_mat_q": (C.c_int, [vp, C.c_char_p, vp, vp, sz, sz]), "lgh_backend_last_error": (C.c_char_p, [vp]),
        "lgh_bench_vec_mat": (C.c_int, [C.c_int, u32, vp, vp, sz, sz, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
        "lgh_bench_hbm_read": (C.c_int, [C.c_int, sz, C.c_int, C.POINTER(C.c_double)]),
        "lgh_gguf_inspect": (C.c_int, [C.c_char_p, C.POINTER(GgufInfo), C.c_char_p, sz]),
        "lgh_load_gguf": (C.c_int, [C.c_char_p, u32, C.c_int, u32, u32, u32, C.POINTER(vp), C.c_char_p, sz]),
        "lgh_stage_io_buffers": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]), "lgh_stage_set_forward_targets": (C.c_int, [vp, vp, vp]), "lgh_stage_step": (C.c_int, [vp, C.c_int]),
        "lgh_stage_read_tokens": (C.c_int, [vp, sz, sz, vp]),
        "lgh_gguf_get": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(GgufValue), C.c_char_p, sz]),
        "lgh_stage_read_logits": (C.c_int, [vp, vp]),
        "lgh_pipeline_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
        "lgh_pipeline_upload_tensor": (C.c_int, [vp, C.c_char_p, u32, C.POINTER(C.c_uint64), vp, sz]),
        "lgh_pipeline_finalize": (C.c_int, [vp]), "lgh_pipeline_destroy": (None, [vp]),
        "lgh_pipeline_forward": (C.c_int, [vp, u32, vp]), "lgh_pipeline_prefill_token": (C.c_int, [vp, u32]),
        "lgh_pipeline_decode_greedy": (C.c_int, [vp, u32, sz, vp]), "lgh_pipeline_reset": (None, [vp]),
        "lgh_pipeline_position": (sz, [vp]), "lgh_pipeline_kv_truncate": (C.c_int, [vp, sz]), "lgh_pipeline_stages": (C.c_int, [vp]), "lgh_pipeline_last_error": (C.c_char_p, [vp]),
        "lgh_set_kv_rotation_signs": (C.c_int, [vp, vp, sz]), "lgh_op_tq_compress": (C.c_int, [C.c_int, C.c_int, vp, sz, vp, vp]),
        "lgh_set_kv_qjl_matrices": (C.c_int, [vp, vp, sz]), "lgh_op_tq_compress_qjl": (C.c_int, [C.c_int, C.c_int, vp, sz, vp, vp, vp, vp, vp]),
        "lgh_batch_create": (C.c_int, [vp, u32]), "lgh_batch_reset": (C.c_int, [vp, u32]), "lgh_batch_position": (sz, [vp, u32]),
        "lgh_batch_prefill": (C.c_int, [vp, u32, vp, sz]), "lgh_forward_multi": (C.c_int, [vp, vp, vp, u32, vp, vp]),
        "lgh_decode_greedy_multi": (C.c_int, [vp, vp, vp, u32, sz, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def device_count() -> int:
    return load_library().lgh_device_count()


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _chk(status: int, what: str = "") -> None:
    if status != 0:
        raise BackendError(status, what)


class HipGpuInference:
    """`impl GpuInference for HipGpuInference` (src/backend/mod.rs:283-296), one context per GPU."""

    def __init__(self):
        self._h = C.c_void_p()
        self.config = None
        self.vocab_size = 0

    # -- pub fn from_model(model: LlamaModel, max_seq_len: usize) -> BackendResult<Self>  (gpu_only.rs:426)
    @classmethod
    def from_model(cls, model, max_seq_len: int, device: int = 0, layer_range: Optional[Sequence[int]] = None,
                   flags: int = 0, attn_splits: int = 0, attn_direct: int = 0, kv_cache_type: int = 0,
                   kv_rotation_signs=None, kv_qjl_matrices=None) -> "HipGpuInference":
        """`model` hands over what LlamaModel::into_parts does (llama.rs:138-160): `.config` and
        `.tensors(layers)` yielding (gguf_name, ggml_type, ne, host bytes)."""
        L = load_library()
        self = cls()
        cfg = model.config
        d = ModelDesc()
        d.struct_size = C.sizeof(ModelDesc)
        for k in ("hidden_size", "intermediate_size", "num_layers", "num_heads", "num_kv_heads", "head_dim",
                  "vocab_size", "num_experts", "num_experts_per_token", "expert_intermediate_size"):
            setattr(d, k, int(getattr(cfg, k)))
        d.max_seq_len = int(max_seq_len)
        d.use_neox_rope = int(cfg.use_neox_rope)
        d.norm_eps, d.rope_freq_base, d.rope_freq_scale = cfg.norm_eps, cfg.rope_freq_base, cfg.rope_freq_scale
        d.device_id = device
        lb, le = (0, cfg.num_layers) if layer_range is None else (layer_range[0], layer_range[1])
        d.layer_begin, d.layer_end = lb, le
        # attn_direct: 64-row units, 0 = tuned default, 255 = never
        d.flags = flags | ((attn_splits & 0xFF) << 8) | ((attn_direct & 0xFF) << 16)
        d.kv_cache_type = int(kv_cache_type)
        _chk(L.lgh_create(C.byref(d), C.byref(self._h)), "lgh_create (is a HIP device visible?)")
        self.config, self.vocab_size, self.hidden_size = cfg, cfg.vocab_size, cfg.hidden_size
        try:
            if kv_rotation_signs is not None:      # TurboQuant: HadamardRotation::signs() of every (layer, kv head, k / v) engine
                sg = np.ascontiguousarray(kv_rotation_signs, dtype=np.float32)
                self._call(L.lgh_set_kv_rotation_signs(self._h, sg.ctypes.data, sg.size))
            if kv_qjl_matrices is not None:        # TurboQuantProd: the K engines' QjlProjector matrices [layer][kv head][d][d]
                qm = np.ascontiguousarray(kv_qjl_matrices, dtype=np.float32)
                self._call(L.lgh_set_kv_qjl_matrices(self._h, qm.ctypes.data, qm.size))
            for name, t, ne, data in model.tensors(range(lb, le)):
                self.upload_tensor(name, t, ne, data)
            self._call(L.lgh_finalize(self._h))
        except Exception:
            self.close()
            raise
        return self

    @classmethod
    def from_gguf(cls, path: str, max_seq_len: int = 0, device: int = 0, layer_range: Optional[Sequence[int]] = None,
                  flags: int = 0) -> "HipGpuInference":
        """GGUF -> HBM direct load (lgh_load_gguf): header parsed and tensors uploaded by the library itself."""
        L = load_library()
        info = gguf_inspect(path)
        self = cls()
        err = C.create_string_buffer(512)
        lb, le = (0, 0) if layer_range is None else (layer_range[0], layer_range[1])
        rc = L.lgh_load_gguf(path.encode(), int(max_seq_len), device, flags, lb, le, C.byref(self._h), err, len(err))
        if rc:
            raise BackendError(rc, "lgh_load_gguf: " + err.value.decode(errors="replace"))
        d = info["desc"]
        self.vocab_size, self.hidden_size = d["vocab_size"], d["hidden_size"]
        self.config = None
        return self

    def upload_tensor(self, name: str, ggml_type: int, ne: Iterable[int], data: np.ndarray) -> None:
        data = np.ascontiguousarray(data)
        ne = list(ne)
        ne4 = (C.c_uint64 * 4)(*(ne + [0] * (4 - len(ne))))
        self._call(load_library().lgh_upload_tensor(self._h, name.encode(), ggml_type, ne4, data.ctypes.data, data.nbytes))

    def _call(self, status: int) -> None:
        if status != 0:
            raise BackendError(status, load_library().lgh_last_error(self._h).decode())

    # -- GpuInference
    def forward(self, token_id: int) -> np.ndarray:
        logits = np.empty(self.vocab_size, dtype=np.float32)
        self._call(load_library().lgh_forward(self._h, token_id, logits.ctypes.data))
        return logits

    def prefill_token(self, token_id: int) -> None:
        self._call(load_library().lgh_prefill_token(self._h, token_id))

    def forward_batch(self, tokens: Sequence[int]) -> None:  # gpu_only.rs:776-790 minus the final forward
        toks = np.ascontiguousarray(tokens, dtype=np.uint32)
        self._call(load_library().lgh_prefill_batch(self._h, toks.ctypes.data, toks.size))

    def prefill_is_batched(self) -> bool:
        return bool(load_library().lgh_prefill_is_batched(self._h))

    def reset(self) -> None:
        load_library().lgh_reset(self._h)

    def position(self) -> int:
        return load_library().lgh_position(self._h)

    def kv_truncate(self, new_len: int) -> None:       # KVCache::truncate (model/mod.rs:130-134)
        self._call(load_library().lgh_kv_truncate(self._h, new_len))

    def kv_shift_left(self, amount: int) -> None:      # KVCache::shift_left (model/mod.rs:142-172) on the device cache
        self._call(load_library().lgh_kv_shift_left(self._h, amount))

    # -- bench fast paths
    def forward_argmax(self, token_id: int) -> int:
        nxt = C.c_uint32()
        self._call(load_library().lgh_forward_argmax(self._h, token_id, C.byref(nxt)))
        return nxt.value

    def decode_greedy(self, first_token: int, n_steps: int) -> np.ndarray:
        out = np.zeros(n_steps, dtype=np.uint32)
        self._call(load_library().lgh_decode_greedy(self._h, first_token, n_steps, out.ctypes.data))
        return out

    # -- multi-sequence decode: the device side of BatchedEngine (src/engine_batched.rs; include/llama_gguf_hip.h lgh_batch_*)
    def batch_create(self, max_batch: int) -> None:
        """`max_batch` slots (BatchedEngineConfig::max_batch_size), each one sequence's KV caches + position."""
        self._call(load_library().lgh_batch_create(self._h, max_batch))

    def batch_reset(self, slot: int) -> None:          # create_active_sequence: a fresh context for the slot
        self._call(load_library().lgh_batch_reset(self._h, slot))

    def batch_position(self, slot: int) -> int:
        return load_library().lgh_batch_position(self._h, slot)

    def batch_prefill(self, slot: int, tokens: Sequence[int]) -> None:
        toks = np.ascontiguousarray(tokens, dtype=np.uint32)
        self._call(load_library().lgh_batch_prefill(self._h, slot, toks.ctypes.data, toks.size))

    def forward_multi(self, slots: Sequence[int], tokens: Sequence[int], want_logits: bool = True, greedy: bool = False):
        """One iteration of the batched loop (engine_batched.rs:236-290): tokens[i] to slot slots[i], every weight read once.
        Returns (logits [n_seq, vocab] or None, next tokens [n_seq] or None)."""
        sl = np.ascontiguousarray(slots, dtype=np.uint32)
        tk = np.ascontiguousarray(tokens, dtype=np.uint32)
        assert sl.size == tk.size
        logits = np.empty((sl.size, self.vocab_size), dtype=np.float32) if want_logits else None
        nxt = np.zeros(sl.size, dtype=np.uint32) if greedy else None
        self._call(load_library().lgh_forward_multi(self._h, sl.ctypes.data, tk.ctypes.data, sl.size,
                                                    logits.ctypes.data if logits is not None else None,
                                                    nxt.ctypes.data if nxt is not None else None))
        return logits, nxt

    def decode_greedy_multi(self, slots: Sequence[int], first_tokens: Sequence[int], n_steps: int) -> np.ndarray:
        """n_steps greedy iterations, tokens fed back on the device; returns [n_steps, n_seq]."""
        sl = np.ascontiguousarray(slots, dtype=np.uint32)
        tk = np.ascontiguousarray(first_tokens, dtype=np.uint32)
        out = np.zeros((n_steps, sl.size), dtype=np.uint32)
        self._call(load_library().lgh_decode_greedy_multi(self._h, sl.ctypes.data, tk.ctypes.data, sl.size, n_steps, out.ctypes.data))
        return out

    # -- pipeline stage
    def stage_hidden_ptr(self) -> int:
        p = C.c_void_p()
        self._call(load_library().lgh_stage_hidden_buffer(self._h, C.byref(p)))
        return p.value

    def stage_hidden_block_ptr(self) -> int:
        """Device address of the [128][hidden] f32 block the batched prompt path hands from stage to stage."""
        p = C.c_void_p()
        self._call(load_library().lgh_stage_hidden_block_buffer(self._h, C.byref(p)))
        return p.value

    def stage_prefill_batch(self, tokens: Optional[Sequence[int]], n: int) -> None:
        """Up to 128 prompt tokens through this stage's layers on the batched path (`tokens` is read on the first stage)."""
        toks = np.ascontiguousarray(tokens, dtype=np.uint32) if tokens is not None else None
        self._call(load_library().lgh_stage_prefill_batch(self._h, toks.ctypes.data if toks is not None else None, n))

    def stage_forward(self, token_id: int = 0, want_logits: bool = False, argmax: bool = False):
        logits = np.empty(self.vocab_size, dtype=np.float32) if (want_logits and not argmax) else None
        nxt = C.c_uint32()
        self._call(load_library().lgh_stage_forward(
            self._h, token_id, int(want_logits), logits.ctypes.data if logits is not None else None,
            C.byref(nxt) if (want_logits and argmax) else None))
        return nxt.value if (want_logits and argmax) else logits

    def stage_io_ptrs(self):
        """(token_in, argmax_out): device addresses of the int32 words the first stage embeds from / the last stage's
        arg-max lands in — the token is fed back between them without the host."""
        a, b = C.c_void_p(), C.c_void_p()
        self._call(load_library().lgh_stage_io_buffers(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stage_step(self, mode: int = 0) -> None:
        """One token through this stage with no host value in or out (0 layers only, 1 + logits, 2 + device arg-max)."""
        self._call(load_library().lgh_stage_step(self._h, mode))

    def stage_read_tokens(self, pos0: int, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint32)
        self._call(load_library().lgh_stage_read_tokens(self._h, pos0, n, out.ctypes.data))
        return out

    def set_stream(self, stream_handle: int) -> None:
        self._call(load_library().lgh_set_stream(self._h, stream_handle))

    def synchronize(self) -> None:
        self._call(load_library().lgh_synchronize(self._h))

    def read_hidden(self) -> np.ndarray:
        out = np.empty(self.hidden_size, dtype=np.float32)
        self._call(load_library().lgh_read_hidden(self._h, out.ctypes.data))
        return out

    def set_profiling(self, on: bool) -> None:
        self._call(load_library().lgh_set_profiling(self._h, int(on)))

    def stats(self) -> dict:
        s = Stats()
        self._call(load_library().lgh_get_stats(self._h, C.byref(s)))
        out = {k: getattr(s, k) for k in ("weight_bytes", "kv_bytes", "scratch_bytes", "tokens_processed",
                                          "graph_nodes", "step_alg_bytes", "event_bracket_us", "overlapped_edges")}
        out["kernels"] = {K_NAMES[i]: {"launches": s.k_launches[i], "time_us": s.k_time_us[i],
                                       "alg_bytes": s.k_alg_bytes[i]}
                          for i in range(len(K_NAMES)) if s.k_launches[i]}
        out["symbols"] = {SYM_NAMES[i]: {"launches": s.sym_launches[i], "time_us": s.sym_time_us[i],
                                         "alg_bytes": s.sym_alg_bytes[i]}
                          for i in range(len(SYM_NAMES)) if s.sym_launches[i]}
        return out

    def close(self) -> None:
        if self._h:
            load_library().lgh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipPipeline:
    """The in-library layer pipeline behind the GpuInference surface (lgh_pipeline_*): one process, `n_stages` stage contexts
    on `devices` (default: all on device 0), hidden vector and greedy token hopped device to device.  `GpuModelWrapper`
    drives it like a single HipGpuInference."""

    def __init__(self):
        self._h = C.c_void_p()
        self.config = None
        self.vocab_size = 0

    @classmethod
    def from_model(cls, model, max_seq_len: int, n_stages: int, devices: Optional[Sequence[int]] = None, flags: int = 0) -> "HipPipeline":
        L = load_library()
        self = cls()
        cfg = model.config
        d = ModelDesc()
        d.struct_size = C.sizeof(ModelDesc)
        for k in ("hidden_size", "intermediate_size", "num_layers", "num_heads", "num_kv_heads", "head_dim",
                  "vocab_size", "num_experts", "num_experts_per_token", "expert_intermediate_size"):
            setattr(d, k, int(getattr(cfg, k)))
        d.max_seq_len = int(max_seq_len)
        d.use_neox_rope = int(cfg.use_neox_rope)
        d.norm_eps, d.rope_freq_base, d.rope_freq_scale = cfg.norm_eps, cfg.rope_freq_base, cfg.rope_freq_scale
        d.device_id, d.flags = 0, flags
        devs = (C.c_int * n_stages)(*devices) if devices is not None else None
        _chk(L.lgh_pipeline_create(C.byref(d), devs, n_stages, C.byref(self._h)), "lgh_pipeline_create (is a HIP device visible?)")
        self.config, self.vocab_size, self.hidden_size = cfg, cfg.vocab_size, cfg.hidden_size
        try:
            for name, t, ne, data in model.tensors():
                data = np.ascontiguousarray(data)
                ne = list(ne)
                ne4 = (C.c_uint64 * 4)(*(ne + [0] * (4 - len(ne))))
                self._call(L.lgh_pipeline_upload_tensor(self._h, name.encode(), t, ne4, data.ctypes.data, data.nbytes))
            self._call(L.lgh_pipeline_finalize(self._h))
        except Exception:
            self.close()
            raise
        return self

    def _call(self, status: int) -> None:
        if status != 0:
            raise BackendError(status, load_library().lgh_pipeline_last_error(self._h).decode())

    def forward(self, token_id: int) -> np.ndarray:
        logits = np.empty(self.vocab_size, dtype=np.float32)
        self._call(load_library().lgh_pipeline_forward(self._h, token_id, logits.ctypes.data))
        return logits

    def prefill_token(self, token_id: int) -> None:
        self._call(load_library().lgh_pipeline_prefill_token(self._h, token_id))

    def decode_greedy(self, first_token: int, n_steps: int) -> np.ndarray:
        out = np.zeros(n_steps, dtype=np.uint32)
        self._call(load_library().lgh_pipeline_decode_greedy(self._h, first_token, n_steps, out.ctypes.data))
        return out

    def reset(self) -> None:
        load_library().lgh_pipeline_reset(self._h)

    def position(self) -> int:
        return load_library().lgh_pipeline_position(self._h)

    def kv_truncate(self, new_len: int) -> None:
        self._call(load_library().lgh_pipeline_kv_truncate(self._h, int(new_len)))

    def stages(self) -> int:
        return load_library().lgh_pipeline_stages(self._h)

    def close(self) -> None:
        if self._h:
            load_library().lgh_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class InferenceContext:
    """The fields of the reference's InferenceContext that GpuModelWrapper touches (model/mod.rs:222-326)."""

    def __init__(self):
        self.position = 0
        self.kv_seq_len = 0

    def reset(self) -> None:
        self.position = 0
        self.kv_seq_len = 0


class GpuModelWrapper:
    """GpuModelWrapper<T: GpuInference> (src/backend/mod.rs:302-364): adapts the engine to `Model::forward`."""

    def __init__(self, gpu: HipGpuInference, config=None):
        self.gpu = gpu
        self.config = config if config is not None else gpu.config

    def forward(self, tokens: Sequence[int], ctx: InferenceContext) -> np.ndarray:
        if ctx.position == 0 and self.gpu.position() > 0:  # mod.rs:334-336
            self.gpu.reset()
        if len(tokens) == 0:
            raise ValueError("No tokens to process")           # mod.rs:338-342
        for t in tokens[:-1]:                                   # mod.rs:344-347
            self.gpu.prefill_token(int(t))
        logits = self.gpu.forward(int(tokens[-1]))              # mod.rs:349
        ctx.position += len(tokens)
        ctx.kv_seq_len = ctx.position
        return logits


# ---- per-op surface (the `Backend` trait ops on the path, src/backend/mod.rs:29-265) ----
def op_dequantize(ggml_type: int, raw: np.ndarray, n_elems: int, device: int = 0) -> np.ndarray:
    raw = np.ascontiguousarray(raw)
    out = np.empty(n_elems, dtype=np.float32)
    _chk(load_library().lgh_op_dequantize(device, ggml_type, raw.ctypes.data, n_elems, out.ctypes.data), "dequantize")
    return out


def op_vec_mat(ggml_type: int, w: np.ndarray, x, n: int, device: int = 0) -> np.ndarray:
    w, x = np.ascontiguousarray(w), _f32(x)
    out = np.empty(n, dtype=np.float32)
    _chk(load_library().lgh_op_vec_mat(device, ggml_type, w.ctypes.data, x.ctypes.data, out.ctypes.data, x.size, n), "vec_mat")
    return out


def op_mat_mat(ggml_type: int, w: np.ndarray, x, n: int, device: int = 0) -> np.ndarray:
    """x [m][k] . W^T -> [m][n] on the batched-prefill GEMM path (f16 matrix cores, f32 accumulation)."""
    w, x = np.ascontiguousarray(w), _f32(x)
    m, k = x.shape
    out = np.empty((m, n), dtype=np.float32)
    _chk(load_library().lgh_op_mat_mat(device, ggml_type, w.ctypes.data, x.ctypes.data, out.ctypes.data, k, n, m), "mat_mat")
    return out


def op_norm_vec_mat(ggml_type: int, w: np.ndarray, x, norm_w, eps: float, n: int, device: int = 0) -> np.ndarray:
    w, x, nw = np.ascontiguousarray(w), _f32(x), _f32(norm_w)
    out = np.empty(n, dtype=np.float32)
    _chk(load_library().lgh_op_norm_vec_mat(device, ggml_type, w.ctypes.data, x.ctypes.data, nw.ctypes.data, eps,
                                            out.ctypes.data, x.size, n), "norm_vec_mat")
    return out


def op_swiglu_vec_mat(ggml_type: int, w_gate: np.ndarray, w_up: np.ndarray, x, norm_w, eps: float, n: int,
                      device: int = 0) -> np.ndarray:
    wg, wu, x = np.ascontiguousarray(w_gate), np.ascontiguousarray(w_up), _f32(x)
    nw = _f32(norm_w) if norm_w is not None else None
    out = np.empty(n, dtype=np.float32)
    _chk(load_library().lgh_op_swiglu_vec_mat(device, ggml_type, wg.ctypes.data, wu.ctypes.data, x.ctypes.data,
                                              nw.ctypes.data if nw is not None else None, eps, out.ctypes.data,
                                              x.size, n), "swiglu_vec_mat")
    return out


def op_rms_norm(x, w, eps: float, device: int = 0) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    out = np.empty_like(x)
    _chk(load_library().lgh_op_rms_norm(device, x.ctypes.data, w.ctypes.data, eps, out.ctypes.data, x.size), "rms_norm")
    return out


def op_rope(q, k, pos: int, freq_base: float, freq_scale: float, neox: bool, device: int = 0):
    q, k = _f32(q).copy(), _f32(k).copy()  # [n_heads, d], [n_kv, d]
    _chk(load_library().lgh_op_rope(device, q.ctypes.data, k.ctypes.data, q.shape[0], k.shape[0], q.shape[1], pos,
                                    freq_base, freq_scale, int(neox)), "rope")
    return q, k


def op_attention_cached(q, k_cache, v_cache, scale: float, kv_len: int, n_splits: int = 8, device: int = 0) -> np.ndarray:
    q, kc, vc = _f32(q), _f32(k_cache), _f32(v_cache)  # [nh, d], [nkv, max_seq, d]
    out = np.empty_like(q)
    _chk(load_library().lgh_op_attention_cached(device, q.ctypes.data, kc.ctypes.data, vc.ctypes.data, out.ctypes.data,
                                                q.shape[0], kc.shape[0], q.shape[1], kc.shape[1], scale, kv_len,
                                                n_splits), "attention_cached")
    return out


def op_kv_roundtrip(kv_cache_type: int, row, device: int = 0):
    """One row through a KV cache format (KV_INT8 / KV_FP8_E4M3 / KV_FP8_E5M2) and back: (bytes, scale, values read back)."""
    x = _f32(row)
    b = np.zeros(x.size, dtype=np.uint8)
    back = np.empty_like(x)
    sc = C.c_float(1.0)
    _chk(load_library().lgh_op_kv_roundtrip(device, kv_cache_type, x.ctypes.data, x.size, b.ctypes.data, C.addressof(sc),
                                            back.ctypes.data), "kv_roundtrip")
    return b, float(sc.value), back


def op_tq_compress(x, bits: int, signs, device: int = 0) -> np.ndarray:
    """TurboQuantEngine::compress without QJL (src/model/turboquant/quant.rs:71-103) on the device: the row's packed codes."""
    x, sg = _f32(x), _f32(signs)
    out = np.zeros(x.size // 4 if bits == 2 else x.size // 8 * 3, dtype=np.uint8)
    _chk(load_library().lgh_op_tq_compress(device, bits, x.ctypes.data, x.size, sg.ctypes.data, out.ctypes.data), "tq_compress")
    return out


def op_tq_compress_qjl(x, bits: int, signs, qjl_matrix, device: int = 0):
    """TurboQuantEngine::compress with use_qjl (quant.rs:71-103) on the device -> (codes, qjl_bits uint64[dim / 64], residual_norm)."""
    x, sg, S = _f32(x), _f32(signs), _f32(qjl_matrix)
    assert S.size == x.size * x.size
    codes = np.zeros(x.size // 4 if bits == 2 else x.size // 8 * 3, dtype=np.uint8)
    qb = np.zeros(x.size // 64, dtype=np.uint64)
    norm = C.c_float(0.0)
    _chk(load_library().lgh_op_tq_compress_qjl(device, bits, x.ctypes.data, x.size, sg.ctypes.data, S.ctypes.data, codes.ctypes.data,
                                               qb.ctypes.data, C.addressof(norm)), "tq_compress_qjl")
    return codes, qb, float(norm.value)


def op_silu_mul(gate, up, device: int = 0) -> np.ndarray:
    g, u = _f32(gate), _f32(up)
    out = np.empty_like(g)
    _chk(load_library().lgh_op_silu_mul(device, g.ctypes.data, u.ctypes.data, out.ctypes.data, g.size), "silu_mul")
    return out


class HipBackend:
    """Mirror of the reference's per-op `Backend` trait (src/backend/mod.rs:29-265) over the C ABI: host tensors (numpy
    arrays) in and out, the same method names and argument meaning; what `select_gpu_backend` (src/engine.rs:738-812) would
    hold as `Box<dyn Backend>`.  `load_weight` + a `name=` on vec_mat_q is the CUDA backend's device-resident weight store
    (src/backend/cuda/mod.rs:121-146, 511-575).  No CPU fallback: every method runs a HIP kernel or raises."""

    def __init__(self, device: int = 0):
        self.device = device
        self._h = C.c_void_p()
        _chk(load_library().lgh_backend_create(device, C.byref(self._h)), "lgh_backend_create (is a HIP device visible?)")

    def close(self) -> None:
        if self._h:
            load_library().lgh_backend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def name(self) -> str:
        return "hip"

    def is_available(self) -> bool:
        return device_count() > self.device

    # memory: host tensors are the currency of this surface (backend/mod.rs:40-49)
    def alloc(self, shape, dtype=np.float32) -> np.ndarray:
        return np.zeros(shape, dtype=dtype)

    def copy_to(self, tensor: np.ndarray) -> np.ndarray:
        return np.array(tensor, copy=True)

    def _ew(self, fn, what, *arrs, scalar=None):
        a = [_f32(x) for x in arrs]
        for b in a[1:]:
            if b.shape != a[0].shape:
                raise BackendError(2, f"{what}: shapes {a[0].shape} and {b.shape} differ")   # check_same_shape (ops.rs:1543)
        out = np.empty_like(a[0])
        args = [self.device] + [x.ctypes.data for x in a] + ([scalar] if scalar is not None else []) + [out.ctypes.data, out.size]
        _chk(fn(*args), what)
        return out

    def add(self, a, b) -> np.ndarray:
        return self._ew(load_library().lgh_op_add, "add", a, b)

    def mul(self, a, b) -> np.ndarray:
        return self._ew(load_library().lgh_op_mul, "mul", a, b)

    def scale(self, a, scalar: float) -> np.ndarray:
        return self._ew(load_library().lgh_op_scale, "scale", a, scalar=float(scalar))

    def silu(self, x) -> np.ndarray:
        return self._ew(load_library().lgh_op_silu, "silu", x)

    def gelu(self, x) -> np.ndarray:
        return self._ew(load_library().lgh_op_gelu, "gelu", x)

    def softmax(self, x) -> np.ndarray:
        x = _f32(x)
        out = np.empty_like(x)
        last = x.shape[-1] if x.ndim else 1
        _chk(load_library().lgh_op_softmax(self.device, x.ctypes.data, out.ctypes.data, x.size // max(last, 1), last), "softmax")
        return out

    def rms_norm(self, x, weight, eps: float) -> np.ndarray:
        return op_rms_norm(x, weight, eps, self.device)

    def matmul(self, a, b) -> np.ndarray:
        a, b = _f32(a), _f32(b)
        if a.ndim != 2 or b.ndim != 2:
            raise BackendError(6, "matmul requires 2D tensors")                                  # ops.rs:434-438
        if a.shape[1] != b.shape[0]:
            raise BackendError(2, f"matmul: [{a.shape}] @ [{b.shape}]")                            # ops.rs:443-448
        out = np.empty((a.shape[0], b.shape[1]), dtype=np.float32)
        _chk(load_library().lgh_op_matmul(self.device, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0], a.shape[1], b.shape[1]), "matmul")
        return out

    def matvec(self, a, b) -> np.ndarray:
        a, b = _f32(a), _f32(b)
        if a.ndim != 2 or b.ndim != 1:
            raise BackendError(6, "matvec requires 2D matrix and 1D vector")                     # ops.rs:536-540
        if b.shape[0] != a.shape[1]:
            raise BackendError(2, f"matvec: expected [{a.shape[1]}], got {b.shape}")
        out = np.empty(a.shape[0], dtype=np.float32)
        _chk(load_library().lgh_op_matvec(self.device, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0], a.shape[1]), "matvec")
        return out

    def vec_mat(self, a, b_f32, n: int) -> np.ndarray:
        """x [k] @ W [k, n] with W in GGUF order (element (i, j) at i + j*k, ops.rs:959-1002)."""
        return op_vec_mat(0, _f32(b_f32), a, n, self.device)

    def dequantize(self, ggml_type: int, raw: np.ndarray, n_elems: int) -> np.ndarray:
        return op_dequantize(ggml_type, raw, n_elems, self.device)

    def matvec_q(self, ggml_type: int, a_raw: np.ndarray, b, m: int) -> np.ndarray:
        a_raw, b = np.ascontiguousarray(a_raw), _f32(b)
        out = np.empty(m, dtype=np.float32)
        _chk(load_library().lgh_op_matvec_q(self.device, ggml_type, a_raw.ctypes.data, b.ctypes.data, out.ctypes.data, m, b.size), "matvec_q")
        return out

    def load_weight(self, name: str, ggml_type: int, raw: np.ndarray, k: int, n: int) -> None:
        raw = np.ascontiguousarray(raw)
        self._call(load_library().lgh_backend_load_weight(self._h, name.encode(), ggml_type, raw.ctypes.data, k, n))

    def has_weight(self, name: str) -> bool:
        return bool(load_library().lgh_backend_has_weight(self._h, name.encode()))

    def vec_mat_q(self, a, ggml_type: int = 0, b_raw: Optional[np.ndarray] = None, n: int = 0, name: Optional[str] = None) -> np.ndarray:
        """x [k] @ quantized W (n rows of k/bs blocks).  With `name` of a loaded weight nothing but x crosses PCIe."""
        a = _f32(a)
        if name is not None and self.has_weight(name):
            out = np.empty(n, dtype=np.float32)
            self._call(load_library().lgh_backend_vec_mat_q(self._h, name.encode(), a.ctypes.data, out.ctypes.data, a.size, n))
            return out
        if b_raw is None:
            raise BackendError(6, f"vec_mat_q: no device-resident weight named {name!r} and no host tensor given")
        return op_vec_mat(ggml_type, b_raw, a, n, self.device)

    def rope(self, q, k, pos: int, freq_base: float, freq_scale: float, use_neox: bool):
        return op_rope(q, k, pos, freq_base, freq_scale, use_neox, self.device)

    def attention(self, q, k, v, scale: float) -> np.ndarray:
        q, k, v = _f32(q), _f32(k), _f32(v)
        if q.ndim != 3 or k.ndim != 3 or v.ndim != 3:
            raise BackendError(6, "Attention requires 3D tensors")                               # ops.rs:1365-1369
        if k.shape[2] != q.shape[2] or v.shape != k.shape or q.shape[0] % k.shape[0]:
            raise BackendError(6, "Attention tensor dimension mismatch")                         # ops.rs:1381-1389
        out = np.empty_like(q)
        _chk(load_library().lgh_op_attention(self.device, q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data, q.shape[0],
                                             k.shape[0], q.shape[1], k.shape[1], q.shape[2], scale), "attention")
        return out

    def flash_attention(self, q, k, v, scale: float, causal: bool = True) -> np.ndarray:   # backend/mod.rs:159-171: defaults to attention
        return self.attention(q, k, v, scale)

    def attention_cached(self, q, k_cache, v_cache, scale: float, kv_len: int) -> np.ndarray:
        return op_attention_cached(q, k_cache, v_cache, scale, kv_len, device=self.device)

    def _call(self, status: int) -> None:
        if status != 0:
            raise BackendError(status, load_library().lgh_backend_last_error(self._h).decode())


def bench_vec_mat(ggml_type: int, w: np.ndarray, k: int, n: int, mode: int = 0, iters: int = 50,
                  w2: Optional[np.ndarray] = None, copies: int = 1, device: int = 0) -> float:
    """Mean device time (µs) of one fused mat-vec launch (hipEvents on the launch stream)."""
    w = np.ascontiguousarray(w)
    us = C.c_double()
    _chk(load_library().lgh_bench_vec_mat(device, ggml_type, w.ctypes.data, w2.ctypes.data if w2 is not None else None,
                                          k, n, mode, iters, copies, C.byref(us)), "bench_vec_mat")
    return us.value


def bench_hbm_read(nbytes: int = 1 << 30, iters: int = 10, device: int = 0) -> float:
    """Measured streaming-read rate of the device in GB/s (the practical ceiling next to the 8 TB/s spec peak)."""
    out = C.c_double(0.0)
    _chk(load_library().lgh_bench_hbm_read(device, nbytes, iters, C.byref(out)), "bench_hbm_read")
    return float(out.value)


def gguf_inspect(path: str) -> dict:
    """Header of a GGUF file as the loader sees it (no GPU needed): version, counts, architecture, model description."""
    info = GgufInfo()
    err = C.create_string_buffer(512)
    rc = load_library().lgh_gguf_inspect(path.encode(), C.byref(info), err, len(err))
    if rc:
        raise BackendError(rc, "lgh_gguf_inspect: " + err.value.decode(errors="replace"))
    return {"version": info.version, "alignment": info.alignment, "n_tensors": info.n_tensors, "n_kv": info.n_kv,
            "data_offset": info.data_offset, "file_bytes": info.file_bytes, "architecture": info.architecture.decode(),
            "has_model_config": info.desc.struct_size != 0, "note": err.value.decode(errors="replace"),
            "desc": {n: getattr(info.desc, n) for n, _ in ModelDesc._fields_}}


GGUF_VALUE_TYPES = ("u8", "i8", "u16", "i16", "u32", "i32", "f32", "bool", "string", "array", "u64", "i64", "f64")


def gguf_get(path: str, key: str):
    """One metadata value: (type name, python value) — GgufData::get_* (src/gguf/types.rs:71-104); None when absent."""
    v = GgufValue()
    err = C.create_string_buffer(512)
    rc = load_library().lgh_gguf_get(path.encode(), key.encode(), C.byref(v), err, len(err))
    if rc:
        msg = err.value.decode(errors="replace")
        if msg.startswith("no metadata key"):
            return None
        raise BackendError(rc, "lgh_gguf_get: " + msg)
    t = GGUF_VALUE_TYPES[v.type]
    if t in ("f32", "f64"):
        return t, float(v.f)
    if t == "string":
        return t, v.s.decode(errors="replace")
    if t == "array":
        return t, int(v.arr_len)
    if t == "bool":
        return t, bool(v.u)
    if t in ("i8", "i16", "i32", "i64"):
        return t, int(C.c_int64(v.u).value)
    return t, int(v.u)
