"""Layer-pipeline decode across the GPUs of one node (one process per GPU).

Restates the reference's partitioning scheme — contiguous layer ranges per shard, embedding on the first
stage, final norm + output projection on the last (src/distributed/pipeline.rs:50-96, shard.rs:377-445,
model.rs:87-148; there the hop is a protobuf `TensorData` over gRPC/TCP) — with the ONE exchange the path
needs done device-to-device: per stage boundary and token, an f32[hidden_size] vector (16-32 KB) over xGMI
via RCCL send/recv, plus the 4-byte greedy token fed back from the last stage to the first.  No collective:
single-stream decode has nothing to reduce.

The host protocol below is transport-agnostic (`comm` needs send/recv of a tensor), so it is covered on CPU
with world_size-2 gloo tests and a fake stage.
"""
from __future__ import annotations

from typing import List, Tuple


def split_layers(num_layers: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal layer ranges (earlier stages take the remainder)."""
    if world < 1 or world > num_layers:
        raise ValueError(f"cannot split {num_layers} layers over {world} stages")
    base, rem = divmod(num_layers, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


class TorchComm:
    """send/recv over torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" on CPU)."""

    def __init__(self, dist):
        self.dist = dist

    def send(self, tensor, dst: int) -> None:
        self.dist.send(tensor, dst)

    def recv(self, tensor, src: int) -> None:
        self.dist.recv(tensor, src)


class PipelineDecoder:
    """Greedy single-stream decode over `world` stages.

    `stage` must provide:
      hidden            tensor the stage reads its input from / leaves its output in (f32[hidden_size])
      token_buf         int32[1] tensor on the transport's device
      run(token, last)  run the stage's layers for one token; on the last stage returns the arg-max token
    """

    def __init__(self, stage, rank: int, world: int, comm):
        self.stage, self.rank, self.world, self.comm = stage, rank, world, comm
        self.first, self.last = rank == 0, rank == world - 1

    def step(self, token: int) -> int:
        """One decode step.  `token` is only read on rank 0; the return value is the next token on rank 0 and
        on the last rank (-1 elsewhere)."""
        st = self.stage
        if not self.first:
            self.comm.recv(st.hidden, self.rank - 1)
        nxt = st.run(token if self.first else 0, self.last)
        if not self.last:
            self.comm.send(st.hidden, self.rank + 1)
        if self.world == 1:
            return nxt
        if self.last:
            st.token_buf[0] = int(nxt)
            self.comm.send(st.token_buf, 0)
            return int(nxt)
        if self.first:
            self.comm.recv(st.token_buf, self.world - 1)
            return int(st.token_buf[0].item())
        return -1

    def prefill(self, tokens) -> None:
        """The prompt (all ranks pass the same token list; only rank 0 reads the ids).  Stages with a batched prompt path
        (`stage.block` / `stage.run_block`) move blocks of up to `stage.block_tokens` hidden vectors per hop, otherwise the
        tokens go through one by one as in `step` (nothing is returned either way: a prefill only fills the caches)."""
        st = self.stage
        bt = getattr(st, "block_tokens", 0)
        if not bt:
            for t in tokens:
                if not self.first:
                    self.comm.recv(st.hidden, self.rank - 1)
                st.run(int(t) if self.first else 0, False)
                if not self.last:
                    self.comm.send(st.hidden, self.rank + 1)
            return
        hs = st.hidden.numel()
        for i in range(0, len(tokens), bt):
            chunk = [int(t) for t in tokens[i:i + bt]]
            view = st.block[:len(chunk) * hs]
            if not self.first:
                self.comm.recv(view, self.rank - 1)
            st.run_block(chunk if self.first else None, len(chunk))
            if not self.last:
                self.comm.send(view, self.rank + 1)

    def decode(self, first_token: int, n_steps: int) -> List[int]:
        out, tok = [], first_token
        for _ in range(n_steps):
            tok = self.step(tok)
            out.append(tok)
        return out


class DevicePtrTensor:
    """Exposes a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr: int, n_elems: int, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": (n_elems,), "typestr": typestr, "data": (ptr, False), "version": 2}


class HipStage:
    """A HipGpuInference stage context as a PipelineDecoder stage (tensors alias the engine's HBM buffers)."""

    def __init__(self, engine, torch, device):
        self.engine = engine
        self.hidden = torch.as_tensor(DevicePtrTensor(engine.stage_hidden_ptr(), engine.hidden_size), device=device)
        self.token_buf = torch.zeros(1, dtype=torch.int32, device=device)
        # kernels and the RCCL hop are ordered on torch's current stream
        engine.set_stream(torch.cuda.current_stream(device).cuda_stream)
        # batched prompt path (lgh_stage_prefill_batch): a [128][hidden] f32 block per hop instead of one vector per token
        self.block_tokens = 0
        if engine.prefill_is_batched():
            self.block_tokens = 128
            self.block = torch.as_tensor(DevicePtrTensor(engine.stage_hidden_block_ptr(), 128 * engine.hidden_size), device=device)

    def run_block(self, tokens, n: int) -> None:
        self.engine.stage_prefill_batch(tokens, n)

    def run(self, token: int, last: bool) -> int:
        if last:
            return int(self.engine.stage_forward(token, want_logits=True, argmax=True))
        self.engine.stage_forward(token)
        return -1
