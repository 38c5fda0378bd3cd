"""GGUF -> HBM direct loader (lgh_gguf_inspect / lgh_load_gguf).

CPU: the header parser against files written by tools/write_gguf.py (the reference's reading rules, src/gguf/reader.rs:
49-104; ModelConfig keys, src/model/loader.rs:62-170) and against damaged files.  GPU: a model loaded from the file gives
bit-identical logits to the same tensors handed over one by one through lgh_upload_tensor."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from write_gguf import STR, U32, write_gguf  # noqa: E402


def _write(pkg, tmp_path, name="test-dense", mix="Q4_K_M", **kw):
    cfg = pkg.make_config(name, max_seq_len=64)
    model = pkg.SynthModel(cfg, mix=mix)
    path = str(tmp_path / f"{name}-{mix}.gguf")
    size = write_gguf(path, cfg, model.tensors(), **kw)
    return cfg, model, path, size


@pytest.mark.parametrize("name,mix", [("test-dense", "Q4_K_M"), ("test-moe", "Q5_K_M")])
def test_inspect_reads_what_the_writer_wrote(pkg, tmp_path, name, mix):
    cfg, model, path, size = _write(pkg, tmp_path, name, mix, extra_kv=[("tokenizer.ggml.model", STR, "llama"), ("general.file_type", U32, 15)])
    info = pkg.hip_backend.gguf_inspect(path)
    assert info["version"] == 3 and info["architecture"] == "llama" and info["alignment"] == 32
    assert info["file_bytes"] == size == os.path.getsize(path) and info["data_offset"] % 32 == 0
    assert info["n_tensors"] == len(list(model.specs()))
    d = info["desc"]
    for k in ("hidden_size", "intermediate_size", "num_layers", "num_heads", "num_kv_heads", "head_dim", "vocab_size",
              "num_experts", "num_experts_per_token", "expert_intermediate_size"):
        assert d[k] == getattr(cfg, k), k
    assert d["max_seq_len"] == 64 and d["use_neox_rope"] == 0
    assert abs(d["norm_eps"] - cfg.norm_eps) < 1e-12 and d["rope_freq_base"] == np.float32(cfg.rope_freq_base)


def test_inspect_defaults_and_neox_archs(pkg, tmp_path):
    cfg = pkg.make_config("test-dense", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q8_0")
    path = str(tmp_path / "q.gguf")
    write_gguf(path, cfg, model.tensors(), arch="qwen2", alignment=64)
    info = pkg.hip_backend.gguf_inspect(path)
    assert info["architecture"] == "qwen2" and info["desc"]["use_neox_rope"] == 1 and info["alignment"] == 64   # loader.rs:145-162


def test_damaged_files_are_rejected(pkg, tmp_path):
    cfg, model, path, size = _write(pkg, tmp_path)
    raw = open(path, "rb").read()
    cases = {"magic": b"GGML" + raw[4:], "version": raw[:4] + struct.pack("<I", 9) + raw[8:], "truncated": raw[:200],
             "counts": raw[:8] + struct.pack("<QQ", 1 << 40, 3) + raw[24:], "tiny": raw[:10]}
    for name, blob in cases.items():
        p = str(tmp_path / f"bad_{name}.gguf")
        open(p, "wb").write(blob)
        with pytest.raises(pkg.BackendError) as ei:
            pkg.hip_backend.gguf_inspect(p)
        assert ei.value.variant in ("InvalidArgument", "Unsupported"), name
    with pytest.raises(pkg.BackendError):
        pkg.hip_backend.gguf_inspect(str(tmp_path / "does_not_exist.gguf"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,mix", [("test-dense", "Q4_K_M"), ("test-moe", "Q5_K_M"), ("test-dense-d128", "Q8_0")])
def test_load_gguf_equals_upload_tensor_path(gpu, pkg, tmp_path, name, mix):
    cfg, model, path, _ = _write(pkg, tmp_path, name, mix)
    a = pkg.HipGpuInference.from_model(model, 64)
    b = pkg.HipGpuInference.from_gguf(path, 64)
    try:
        toks = [3, 77, 500, 9]
        for t in toks[:-1]:
            a.prefill_token(t)
            b.prefill_token(t)
        assert np.array_equal(a.forward(toks[-1]), b.forward(toks[-1]))
        assert a.decode_greedy(5, 16).tolist() == b.decode_greedy(5, 16).tolist()
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_load_gguf_rejects_other_architectures_and_bad_tensors(gpu, pkg, tmp_path):
    cfg = pkg.make_config("test-dense", max_seq_len=64)
    model = pkg.SynthModel(cfg, mix="Q4_K")
    p1 = str(tmp_path / "mamba.gguf")
    write_gguf(p1, cfg, model.tensors(), arch="mamba")
    with pytest.raises(pkg.BackendError) as ei:
        pkg.HipGpuInference.from_gguf(p1, 64)
    assert ei.value.variant == "Unsupported"
    p2 = str(tmp_path / "missing.gguf")
    write_gguf(p2, cfg, [t for t in model.tensors() if "blk.1.ffn_down" not in t[0]])
    with pytest.raises(pkg.BackendError):
        pkg.HipGpuInference.from_gguf(p2, 64)     # finalize: a tensor the stage needs is absent
