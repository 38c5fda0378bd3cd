"""The N>1 path on CPU: layer partitioning and the pipeline host protocol (hidden hop + token feedback) with
world_size-2 and -3 `gloo` process groups and a fake stage (no GPU).  The protocol code under test is the one
bench.py --gpus N runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def _pipeline():
    graft.load_package()
    from importlib import import_module
    return import_module("llama_gguf_amd.pipeline")


def test_split_layers():
    pl = _pipeline()
    assert pl.split_layers(32, 1) == [(0, 32)]
    assert pl.split_layers(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    assert pl.split_layers(80, 8)[-1] == (70, 80)
    assert pl.split_layers(22, 4) == [(0, 6), (6, 12), (12, 17), (17, 22)]
    for n, w in ((22, 3), (32, 5), (80, 7)):
        parts = pl.split_layers(n, w)
        assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    with pytest.raises(ValueError):
        pl.split_layers(4, 5)


HID, VOCAB = 16, 97


class FakeStage:
    """Deterministic stand-in for a stage context: layer l maps h -> h * 1.5 + l (f32), embedding is token-keyed."""

    def __init__(self, lo, hi, first, block_tokens=0):
        self.lo, self.hi, self.first = lo, hi, first
        self.hidden = torch.zeros(HID, dtype=torch.float32)
        self.token_buf = torch.zeros(1, dtype=torch.int32)
        self.pos = 0
        self.block_tokens = block_tokens                     # > 0: the stage has a batched prompt path
        self.block = torch.zeros(max(block_tokens, 1) * HID, dtype=torch.float32)

    def run_block(self, tokens, n):
        rows = self.block[:n * HID].view(n, HID)
        for i in range(n):
            if self.first:
                rows[i].copy_(torch.arange(HID, dtype=torch.float32) * 0.01 + float(tokens[i]))
            for l in range(self.lo, self.hi):
                rows[i].mul_(1.5).add_(float(l) + 0.25 * (self.pos + i))
        self.pos += n

    def run(self, token, last):
        if self.first:
            self.hidden.copy_(torch.arange(HID, dtype=torch.float32) * 0.01 + float(token))
        for l in range(self.lo, self.hi):
            self.hidden.mul_(1.5).add_(float(l) + 0.25 * self.pos)
        self.pos += 1
        if last:
            return int(self.hidden.abs().sum().item()) % VOCAB
        return -1


def _reference(n_layers, first_token, n_steps, prompt=()):
    st = FakeStage(0, n_layers, True)
    for t in prompt:
        st.run(t, False)
    out, tok = [], first_token
    for _ in range(n_steps):
        tok = st.run(tok, True)
        out.append(tok)
    return out


def _worker(rank, world, port, n_layers, n_steps, q, prompt=(), block_tokens=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pl = _pipeline()
    lo, hi = pl.split_layers(n_layers, world)[rank]
    dec = pl.PipelineDecoder(FakeStage(lo, hi, rank == 0, block_tokens), rank, world, pl.TorchComm(dist))
    if prompt:
        dec.prefill(list(prompt))
    toks = dec.decode(7, n_steps)
    dist.barrier()
    q.put((rank, toks))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_protocol_gloo(world):
    n_layers, n_steps = 7, 12
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_layers, n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _reference(n_layers, 7, n_steps)
    assert results[0] == want                  # the first stage learns every fed-back token
    assert results[world - 1] == want          # the last stage produced them
    for r in range(1, world - 1):
        assert results[r] == [-1] * n_steps


@pytest.mark.parametrize("world,block_tokens", [(2, 4), (3, 4), (2, 0)])
def test_pipeline_prefill_blocks_gloo(world, block_tokens):
    """The prompt hop: blocks of hidden vectors per stage boundary (10 tokens as 4 + 4 + 2) or, for stages without a
    batched path, one vector per token; either way the decode that follows matches the single-stage run."""
    n_layers, n_steps, prompt = 7, 6, (3, 9, 27, 81, 46, 41, 26, 78, 40, 23)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_layers, n_steps, q, prompt, block_tokens)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _reference(n_layers, 7, n_steps, prompt)
    assert results[0] == want and results[world - 1] == want
