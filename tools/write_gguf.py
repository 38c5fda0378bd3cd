#!/usr/bin/env python3
"""Minimal GGUF v3 writer for tests and experiments: the synthetic models of llama-gguf_amd/synth.py as real files.

Layout per the GGUF specification as the reference reads it (src/gguf/reader.rs:49-104): magic, version, tensor count,
metadata count, metadata (key, type, value), tensor infos (name, n_dims, dims, ggml type, offset), padding to
`general.alignment`, tensor data (each tensor aligned).  Only what a decoder needs is written: `general.architecture`,
the `{arch}.*` shape keys, and the tensors.
"""
import struct

U32, F32, STR, U64 = 4, 6, 8, 10


def _s(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


def _kv(key: str, vtype: int, value) -> bytes:
    out = _s(key.encode()) + struct.pack("<I", vtype)
    if vtype == U32:
        out += struct.pack("<I", value)
    elif vtype == U64:
        out += struct.pack("<Q", value)
    elif vtype == F32:
        out += struct.pack("<f", value)
    elif vtype == STR:
        out += _s(value.encode())
    else:
        raise ValueError(vtype)
    return out


def write_gguf(path: str, cfg, tensors, arch: str = "llama", alignment: int = 32, extra_kv=(), version: int = 3) -> int:
    """cfg: llama-gguf_amd synth.ModelConfig; tensors: iterable of (name, ggml_type, ne, bytes ndarray).  Returns the file size."""
    tensors = list(tensors)
    kv = [("general.architecture", STR, arch), ("general.alignment", U32, alignment),
          (f"{arch}.embedding_length", U32, cfg.hidden_size), (f"{arch}.feed_forward_length", U32, cfg.intermediate_size),
          (f"{arch}.block_count", U32, cfg.num_layers), (f"{arch}.attention.head_count", U32, cfg.num_heads),
          (f"{arch}.attention.head_count_kv", U32, cfg.num_kv_heads), (f"{arch}.attention.key_length", U32, cfg.head_dim),
          (f"{arch}.attention.layer_norm_rms_epsilon", F32, cfg.norm_eps), (f"{arch}.rope.freq_base", F32, cfg.rope_freq_base),
          (f"{arch}.context_length", U32, cfg.max_seq_len), (f"{arch}.vocab_size", U32, cfg.vocab_size)]
    if cfg.num_experts:
        kv += [(f"{arch}.expert_count", U32, cfg.num_experts), (f"{arch}.expert_used_count", U32, cfg.num_experts_per_token),
               (f"{arch}.expert_feed_forward_length", U32, cfg.expert_intermediate_size)]
    kv += list(extra_kv)
    head = struct.pack("<IIQQ", 0x46554747, version, len(tensors), len(kv))
    for k, t, v in kv:
        head += _kv(k, t, v)
    infos, off = b"", 0
    offsets = []
    for name, ggml_type, ne, data in tensors:
        ne = [int(x) for x in ne if int(x) > 0]
        while len(ne) > 1 and ne[-1] == 1:
            ne.pop()
        infos += _s(name.encode()) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne)
        infos += struct.pack("<IQ", int(ggml_type), off)
        offsets.append(off)
        off = (off + data.nbytes + alignment - 1) // alignment * alignment
    head += infos
    pad = (-len(head)) % alignment
    with open(path, "wb") as f:
        f.write(head + b"\0" * pad)
        base = f.tell()
        for (name, ggml_type, ne, data), o in zip(tensors, offsets):
            f.seek(base + o)
            f.write(data.tobytes())
        end = f.tell()
        f.write(b"\0" * ((-end) % alignment))
        return f.tell()
