"""CPU-side checks of the drop-in boundary: the C-ABI libraries load and export every symbol that
include/*.h declares, status codes follow BackendError, and the product fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_hip_library_exports_every_declared_symbol(pkg):
    lib = pkg.hip_backend.load_library()
    declared = _declared("llama_gguf_hip.h", "lgh_")
    assert len(declared) >= 30
    for sym in declared:
        assert getattr(lib, sym) is not None, sym
    assert sorted(pkg.hip_backend.ABI_SYMBOLS) == declared   # the Python binding covers the whole header


def test_synth_library_exports_every_declared_symbol(pkg):
    lib = pkg.synth.synth_lib()
    for sym in _declared("llama_gguf_synth.h", "lgs_"):
        assert getattr(lib, sym) is not None, sym


def test_status_codes_follow_backend_error(pkg):
    # src/backend/error.rs:3-37, in declaration order
    text = open(os.path.join(ROOT, "include", "llama_gguf_hip.h")).read()
    codes = dict(re.findall(r"(LGH_[A-Z_]+) = (\d+),?\s*/\* BackendError", text))
    assert codes == {"LGH_NOT_AVAILABLE": "1", "LGH_SHAPE_MISMATCH": "2", "LGH_DTYPE_MISMATCH": "3",
                     "LGH_UNSUPPORTED_DTYPE": "4", "LGH_UNSUPPORTED": "5", "LGH_INVALID_ARGUMENT": "6",
                     "LGH_TENSOR_ERROR": "7", "LGH_INITIALIZATION_FAILED": "8", "LGH_ALLOCATION_FAILED": "9",
                     "LGH_OPERATION_FAILED": "10"}
    assert pkg.BackendError(6, "x").variant == "InvalidArgument"


def test_model_desc_layout_matches_header(pkg):
    text = open(os.path.join(ROOT, "include", "llama_gguf_hip.h")).read()
    body = re.search(r"typedef struct lgh_model_desc \{(.*?)\} lgh_model_desc;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:uint32_t|int32_t|float)\s+(\w+);", body)
    assert fields == [n for n, _ in pkg.hip_backend.ModelDesc._fields_]
    assert C.sizeof(pkg.hip_backend.ModelDesc) == 4 * len(fields)


def test_no_silent_cpu_fallback_without_device(pkg):
    """Without a HIP device every product entry point reports NotAvailable instead of computing."""
    hb = pkg.hip_backend
    if hb.device_count() > 0:
        pytest.skip("a GPU is visible")
    model = pkg.SynthModel(pkg.make_config("test-dense", max_seq_len=8), mix="Q4_K")
    with pytest.raises(pkg.BackendError) as ei:
        pkg.HipGpuInference.from_model(model, 8)
    assert ei.value.variant == "NotAvailable"
    with pytest.raises(pkg.BackendError) as ei:
        hb.op_rms_norm(np.ones(8, np.float32), np.ones(8, np.float32), 1e-5)
    assert ei.value.variant == "NotAvailable"


def test_removed_decode_experiment_flags_are_refused(pkg):
    """The decode structures round 2 measured slower (chained FFN, persistent token kernel, flag-ordered / flow launches,
    self-merging attention, merge inside wo) were removed in round 3: their flag bits answer Unsupported, with or without a GPU."""
    hb = pkg.hip_backend
    model = pkg.SynthModel(pkg.make_config("test-dense", max_seq_len=8), mix="Q4_K")
    for bit in (2, 8, 32, 64, 128, 1 << 24):
        assert bit & hb.FLAG_REMOVED_MASK
        with pytest.raises(pkg.BackendError) as ei:
            pkg.HipGpuInference.from_model(model, 8, flags=bit)
        assert ei.value.variant == "Unsupported", bit


def test_product_does_not_import_the_oracle():
    """The package must never route through oracle/ (it is test infrastructure)."""
    pkg_dir = os.path.join(ROOT, "llama-gguf_amd")
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f
                assert 'oracle/' not in src.replace("oracle/_ref", ""), f
