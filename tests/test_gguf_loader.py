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


def test_mutated_headers_never_crash_the_parser(pkg, tmp_path):
    """400 mutations (bit flips, random bytes, truncations) of a valid file's header / metadata / tensor-info region: the
    parser either reads the file or reports a status — it never faults or hangs.  (The same fuzz at 3000 cases runs under
    AddressSanitizer + UBSan in tools/sanitize/run.sh.)"""
    import random
    cfg, model, path, size = _write(pkg, tmp_path)
    blob = bytearray(open(path, "rb").read())
    rnd = random.Random(11)
    hdr = min(len(blob), 6000)
    ok = bad = 0
    for it in range(400):
        b = bytearray(blob if it % 3 else blob[:rnd.randrange(16, hdr)])
        for _ in range(rnd.randrange(1, 4)):
            k = rnd.randrange(0, min(len(b), hdr))
            b[k] = rnd.randrange(256) if it % 2 else b[k] ^ (1 << rnd.randrange(8))
        p = str(tmp_path / "mut.gguf")
        open(p, "wb").write(b)
        try:
            pkg.hip_backend.gguf_inspect(p)
            ok += 1
        except pkg.BackendError as e:
            assert e.variant in ("InvalidArgument", "Unsupported", "OperationFailed", "ShapeMismatch", "DTypeMismatch", "UnsupportedDType"), e.variant
            bad += 1
    assert ok + bad == 400 and bad > 50


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


# ---------------------------------------------------------------------------------------------------------------
# The reference's own reader fixtures (tests/gguf_reader_test.rs:4-160), restated byte for byte and pushed through
# lgh_gguf_inspect / lgh_gguf_get: these pin the loader's header parser to the reference, not to tools/write_gguf.py.
# ---------------------------------------------------------------------------------------------------------------
GGUF_MAGIC = 0x46554747          # src/gguf/constants.rs:4


def _minimal_v3():               # gguf_reader_test.rs:4-24
    key, val = b"general.architecture", b"llama"
    return (struct.pack("<IIQQ", GGUF_MAGIC, 3, 0, 1) + struct.pack("<Q", len(key)) + key + struct.pack("<I", 8)
            + struct.pack("<Q", len(val)) + val)


def _minimal_v1():               # gguf_reader_test.rs:26-44: u32 counts and u32 string lengths
    key = b"test.key"
    return struct.pack("<IIII", GGUF_MAGIC, 1, 0, 1) + struct.pack("<I", len(key)) + key + struct.pack("<II", 4, 42)


def _put(tmp_path, name, blob):
    p = str(tmp_path / name)
    open(p, "wb").write(blob)
    return p


def test_reference_fixture_minimal_v3(pkg, tmp_path):
    """test_read_minimal_gguf (gguf_reader_test.rs:46-56) + test_data_offset_alignment (:153-160)."""
    p = _put(tmp_path, "v3.gguf", _minimal_v3())
    info = pkg.hip_backend.gguf_inspect(p)
    assert info["version"] == 3 and info["n_tensors"] == 0 and info["n_kv"] == 1
    assert pkg.hip_backend.gguf_get(p, "general.architecture") == ("string", "llama")
    assert info["architecture"] == "llama"
    assert info["data_offset"] % 32 == 0 and info["alignment"] == 32
    assert not info["has_model_config"] and "embedding_length" in info["note"]   # a valid GGUF, but not a model: lgh_load_gguf refuses it
    assert pkg.hip_backend.gguf_get(p, "no.such.key") is None


def test_reference_fixture_v1_u32_counts(pkg, tmp_path):
    """test_read_gguf_v1 (gguf_reader_test.rs:58-68)."""
    p = _put(tmp_path, "v1.gguf", _minimal_v1())
    info = pkg.hip_backend.gguf_inspect(p)
    assert info["version"] == 1 and info["n_tensors"] == 0 and info["n_kv"] == 1
    assert pkg.hip_backend.gguf_get(p, "test.key") == ("u32", 42)
    assert not info["has_model_config"]


def test_reference_fixture_invalid_magic_version_eof(pkg, tmp_path):
    """test_invalid_magic (:70-76), test_unsupported_version (:78-87), test_unexpected_eof (:142-150): the reference's
    GgufError variants, carried as status + the reference's message text (src/gguf/error.rs:3-14)."""
    hb = pkg.hip_backend
    with pytest.raises(pkg.BackendError) as ei:
        hb.gguf_inspect(_put(tmp_path, "magic.gguf", bytes([0, 0, 0, 0, 3, 0, 0, 0])))
    assert ei.value.variant == "InvalidArgument" and "Invalid magic number: expected 0x46554747, got 0x00000000" in str(ei.value)
    with pytest.raises(pkg.BackendError) as ei:
        hb.gguf_inspect(_put(tmp_path, "ver.gguf", struct.pack("<II", GGUF_MAGIC, 99)))
    assert ei.value.variant == "Unsupported" and "Unsupported GGUF version: 99" in str(ei.value)
    with pytest.raises(pkg.BackendError) as ei:
        hb.gguf_inspect(_put(tmp_path, "eof.gguf", struct.pack("<I", GGUF_MAGIC)))
    assert ei.value.variant == "InvalidArgument" and "Unexpected end of file" in str(ei.value)


def test_reference_fixture_every_metadata_type(pkg, tmp_path):
    """test_multiple_metadata_types (gguf_reader_test.rs:89-140): u8, i32, f32, bool, u64 — plus the remaining scalar types,
    a string array and an array of arrays, which the reader also has to step over (reader.rs:106-190)."""
    def kv(key, vtype, payload):
        return struct.pack("<Q", len(key)) + key + struct.pack("<I", vtype) + payload
    body = (kv(b"test.u8", 0, bytes([255])) + kv(b"test.i32", 5, struct.pack("<i", -42)) + kv(b"test.f32", 6, struct.pack("<f", 2.5))
            + kv(b"test.bool", 7, bytes([1])) + kv(b"test.u64", 10, struct.pack("<Q", 0xFFFFFFFFFFFFFFFF)))
    p = _put(tmp_path, "types5.gguf", struct.pack("<IIQQ", GGUF_MAGIC, 3, 0, 5) + body)
    hb = pkg.hip_backend
    info = hb.gguf_inspect(p)
    assert info["n_kv"] == 5
    assert hb.gguf_get(p, "test.u64") == ("u64", 0xFFFFFFFFFFFFFFFF)
    assert hb.gguf_get(p, "test.f32") == ("f32", 2.5)
    assert hb.gguf_get(p, "test.u8") == ("u8", 255) and hb.gguf_get(p, "test.i32") == ("i32", -42) and hb.gguf_get(p, "test.bool") == ("bool", True)
    more = (kv(b"t.i8", 1, struct.pack("<b", -7)) + kv(b"t.u16", 2, struct.pack("<H", 65535)) + kv(b"t.i16", 3, struct.pack("<h", -300))
            + kv(b"t.i64", 11, struct.pack("<q", -(1 << 40))) + kv(b"t.f64", 12, struct.pack("<d", 1.25))
            + kv(b"t.strs", 9, struct.pack("<IQ", 8, 2) + struct.pack("<Q", 2) + b"ab" + struct.pack("<Q", 0))
            + kv(b"t.nested", 9, struct.pack("<IQ", 9, 2) + struct.pack("<IQ", 4, 2) + struct.pack("<II", 1, 2) + struct.pack("<IQ", 0, 1) + b"\x09")
            + kv(b"general.alignment", 4, struct.pack("<I", 64)))
    p2 = _put(tmp_path, "types_more.gguf", struct.pack("<IIQQ", GGUF_MAGIC, 3, 0, 8) + more)
    info = hb.gguf_inspect(p2)
    assert info["n_kv"] == 8 and info["alignment"] == 64 and info["data_offset"] % 64 == 0      # reader.rs:84-96
    assert hb.gguf_get(p2, "t.i8") == ("i8", -7) and hb.gguf_get(p2, "t.u16") == ("u16", 65535) and hb.gguf_get(p2, "t.i16") == ("i16", -300)
    assert hb.gguf_get(p2, "t.i64") == ("i64", -(1 << 40)) and hb.gguf_get(p2, "t.f64") == ("f64", 1.25)
    assert hb.gguf_get(p2, "t.strs") == ("array", 2) and hb.gguf_get(p2, "t.nested") == ("array", 2)
    with pytest.raises(pkg.BackendError) as ei:   # an unknown value type: GgufError::InvalidMetadataType
        hb.gguf_inspect(_put(tmp_path, "badtype.gguf", struct.pack("<IIQQ", GGUF_MAGIC, 3, 0, 1) + kv(b"x", 77, b"\0\0\0\0")))
    assert ei.value.variant == "InvalidArgument"


@pytest.mark.gpu
def test_load_gguf_refuses_a_gguf_without_model_keys(gpu, pkg, tmp_path):
    with pytest.raises(pkg.BackendError) as ei:
        pkg.HipGpuInference.from_gguf(_put(tmp_path, "v3.gguf", _minimal_v3()), 64)
    assert ei.value.variant == "InvalidArgument" and "embedding_length" in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("name,mix", [("test-dense", "Q4_K_M"), ("test-moe", "Q5_K_M"), ("test-dense-d128", "Q8_0")])
def test_gguf_loaded_model_matches_the_oracle(gpu, pkg, orc, tmp_path, name, mix):
    """The loader against the ORACLE (not against the engine's own upload path): a model read from a GGUF file gives the CPU
    oracle's logits and greedy tokens."""
    cfg, model, path, _ = _write(pkg, tmp_path, name, mix)
    ref = orc.Model(cfg.as_dict())
    for nm, t, ne, data in model.tensors(keep=True):
        ref.add_tensor(nm, t, ne, data)
    ref.finalize()
    eng = pkg.HipGpuInference.from_gguf(path, 64)
    try:
        toks = [3, 77, 500, 9]
        for t in toks[:-1]:
            eng.prefill_token(t)
        got, want = eng.forward(toks[-1]), ref.forward(toks)
        for _ in range(6):
            err = float(np.abs(got - want).max())
            assert err <= 2e-3 * float(np.abs(want).max()) + 2e-3
            srt = np.sort(want)
            tok = orc.argmax_last(want)
            if float(srt[-1] - srt[-2]) > 4 * err:
                assert orc.argmax_last(got) == tok
            got, want = eng.forward(tok), ref.forward([tok])
    finally:
        ref.close()
        eng.close()
