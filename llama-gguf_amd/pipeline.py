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


class HostStagedComm:
    """Rehearsal transport: device tensors hop through host memory over a CPU backend (gloo has no device send/recv).  Used to
    run the multi-process protocol with REAL stage engines on a box with one GPU (every rank on device 0; RCCL refuses two ranks
    on one device): same HipStage, same PipelineDecoder calls, but a host round trip per hop — a correctness rehearsal, never a
    measurement."""

    def __init__(self, dist):
        self.dist = dist

    def send(self, tensor, dst: int) -> None:
        self.dist.send(tensor.cpu(), dst)             # (.cpu() waits for the stage's kernels on the current stream)

    def recv(self, tensor, src: int) -> None:
        host = tensor.cpu()
        self.dist.recv(host, src)
        tensor.copy_(host)                            # enqueued on the current stream, ahead of the stage's next kernels


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

    def decode_device(self, first_token, n_steps: int, collect: bool = True):
        """`n_steps` greedy tokens with the token fed back DEVICE TO DEVICE: the last stage's arg-max word goes straight into
        the first stage's token word (send/recv between the stages' own buffers), every stage only enqueues
        recv -> its layers -> send on its stream, and no host value crosses a stage boundary per token (`step` above does a
        host sync and a 4-byte device-to-host copy per boundary).  `first_token` None continues from the token the previous
        call left in the first stage.  With `collect`, the tokens are read from the last stage's log afterwards (one sync)
        and handed to rank 0; returns them on rank 0 and on the last rank, [] elsewhere.

        `stage` must additionally provide: token_in / argmax_out (int32[1] tensors aliasing the device words),
        set_token(t), step_async(last), position(), read_tokens(pos0, n), int_tensor(n)."""
        st, w = self.stage, self.world
        pos0 = st.position()
        if self.first and first_token is not None:
            st.set_token(int(first_token))
        for i in range(n_steps):
            if not self.first:
                self.comm.recv(st.hidden, self.rank - 1)
            elif i > 0 and w > 1:
                self.comm.recv(st.token_in, w - 1)
            st.step_async(self.last)
            if not self.last:
                self.comm.send(st.hidden, self.rank + 1)
            elif w > 1:
                self.comm.send(st.argmax_out, 0)
        if w > 1 and self.first and n_steps > 0:
            self.comm.recv(st.token_in, w - 1)       # the last token: where the next call continues from
        if not collect:
            return []
        if w == 1:
            return [int(t) for t in st.read_tokens(pos0, n_steps)]
        if self.last:
            toks = st.read_tokens(pos0, n_steps)
            buf = st.int_tensor(n_steps)
            buf.copy_(st.int_tensor(n_steps, toks))
            self.comm.send(buf, 0)
            return [int(t) for t in toks]
        if self.first:
            buf = st.int_tensor(n_steps)
            self.comm.recv(buf, w - 1)
            return [int(t) for t in buf.cpu().tolist()]
        return []


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
        # Kernels and the hop must be ordered on ONE stream.  torch's default stream has the handle 0, which lgh_set_stream reads
        # as "the engine's own stream" — a non-blocking stream nothing in torch orders against, so a recv could land while the
        # stage's kernels were still reading the vector (found by the one-GPU multi-process rehearsal, tests/test_gpu_model.py).
        # The stage therefore makes a stream of its own current for torch (RCCL orders send/recv against the current stream)
        # and hands that stream to the engine.
        self.stream = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(self.stream)
        engine.set_stream(self.stream.cuda_stream)
        # batched prompt path (lgh_stage_prefill_batch): a [128][hidden] f32 block per hop instead of one vector per token
        self.block_tokens = 0
        if engine.prefill_is_batched():
            self.block_tokens = 128
            self.block = torch.as_tensor(DevicePtrTensor(engine.stage_hidden_block_ptr(), 128 * engine.hidden_size), device=device)

        tin, aout = engine.stage_io_ptrs()
        self.token_in = torch.as_tensor(DevicePtrTensor(tin, 1, "<i4"), device=device)
        self.argmax_out = torch.as_tensor(DevicePtrTensor(aout, 1, "<i4"), device=device)
        self._torch, self._device = torch, device

    # device-side token feedback (PipelineDecoder.decode_device)
    def set_token(self, token: int) -> None:
        self.token_in.fill_(int(token))

    def step_async(self, last: bool) -> None:
        self.engine.stage_step(2 if last else 0)

    def position(self) -> int:
        return self.engine.position()

    def read_tokens(self, pos0: int, n: int):
        return self.engine.stage_read_tokens(pos0, n)

    def int_tensor(self, n: int, values=None):
        if values is None:
            return self._torch.zeros(n, dtype=self._torch.int32, device=self._device)
        return self._torch.tensor([int(v) for v in values], dtype=self._torch.int32, device=self._device)

    def run_block(self, tokens, n: int) -> None:
        self.engine.stage_prefill_batch(tokens, n)

    def run(self, token: int, last: bool) -> int:
        if last:
            return int(self.engine.stage_forward(token, want_logits=True, argmax=True))
        self.engine.stage_forward(token)
        return -1


class FakeStage:
    """Protocol test double (CPU tensors, no compute kernels): layer l maps h -> 1.5 h + l + pos / 4, the embedding is keyed by
    the token, the "arg-max" is a hash of the final vector.  Used by the world_size > 1 `gloo` tests and by
    `bench.py --fake-stage` (a CPU rehearsal of the launcher and of the hop protocol); never by a measured run."""
    HID, VOCAB = 16, 97

    def __init__(self, lo: int, hi: int, first: bool, block_tokens: int = 0):
        import torch
        self._torch = torch
        self.lo, self.hi, self.first = lo, hi, first
        self.hidden = torch.zeros(self.HID, dtype=torch.float32)
        self.token_buf = torch.zeros(1, dtype=torch.int32)
        self.token_in = torch.zeros(1, dtype=torch.int32)
        self.argmax_out = torch.zeros(1, dtype=torch.int32)
        self.pos = 0
        self.log = {}
        self.block_tokens = block_tokens                     # > 0: the stage has a batched prompt path
        self.block = torch.zeros(max(block_tokens, 1) * self.HID, dtype=torch.float32)

    def _embed(self, dst, token):
        dst.copy_(self._torch.arange(self.HID, dtype=self._torch.float32) * 0.01 + float(token))

    def run_block(self, tokens, n):
        rows = self.block[:n * self.HID].view(n, self.HID)
        for i in range(n):
            if self.first:
                self._embed(rows[i], tokens[i])
            for l in range(self.lo, self.hi):
                rows[i].mul_(1.5).add_(float(l) + 0.25 * (self.pos + i))
        self.pos += n

    def run(self, token, last):
        if self.first:
            self._embed(self.hidden, token)
        for l in range(self.lo, self.hi):
            self.hidden.mul_(1.5).add_(float(l) + 0.25 * self.pos)
        self.pos += 1
        if last:
            return int(self.hidden.abs().sum().item()) % self.VOCAB
        return -1

    # device-feedback interface (decode_device)
    def set_token(self, token):
        self.token_in.fill_(int(token))

    def step_async(self, last):
        p = self.pos
        t = self.run(int(self.token_in.item()), last)
        if last:
            self.argmax_out.fill_(t)
            self.log[p] = t
            if self.first:
                self.token_in.fill_(t)

    def position(self):
        return self.pos

    def read_tokens(self, pos0, n):
        return [self.log[pos0 + i] for i in range(n)]

    def int_tensor(self, n, values=None):
        if values is None:
            return self._torch.zeros(n, dtype=self._torch.int32)
        return self._torch.tensor([int(v) for v in values], dtype=self._torch.int32)
