"""llama-gguf_amd — MI355X-native GPU-resident decode backend for llama-gguf (hot path only).

The directory name is the project's (`llama-gguf_amd`); it is not a valid Python identifier, so the
package is registered as ``llama_gguf_amd`` by ``__graft_entry__.load_package()``.

  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/llama_gguf_hip.h)
  hip_backend.py   host-side mirror of the reference's GpuInference / GpuModelWrapper over that ABI
  synth.py         synthetic random-init models (no model files exist in this environment)
"""
from . import hip_backend, synth  # noqa: F401
from .hip_backend import BackendError, GpuModelWrapper, HipBackend, HipGpuInference, HipPipeline, InferenceContext  # noqa: F401
from .synth import ModelConfig, SynthModel, make_config  # noqa: F401
