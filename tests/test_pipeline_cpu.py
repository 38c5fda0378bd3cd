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


def FakeStage(lo, hi, first, block_tokens=0):
    """The protocol test double shipped with the package (pipeline.FakeStage): deterministic, CPU tensors, no kernels."""
    return _pipeline().FakeStage(lo, hi, first, block_tokens)


def _reference(n_layers, first_token, n_steps, prompt=()):
    st = FakeStage(0, n_layers, True)
    for t in prompt:
        st.run(t, False)
    out, tok = [], first_token
    for _ in range(n_steps):
        tok = st.run(tok, True)
        out.append(tok)
    return out


def _worker(rank, world, port, n_layers, n_steps, q, prompt=(), block_tokens=0, device_feedback=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pl = _pipeline()
    lo, hi = pl.split_layers(n_layers, world)[rank]
    dec = pl.PipelineDecoder(FakeStage(lo, hi, rank == 0, block_tokens), rank, world, pl.TorchComm(dist))
    if prompt:
        dec.prefill(list(prompt))
    if device_feedback:     # two calls: the second continues from the token the first left in the first stage
        toks = dec.decode_device(7, n_steps // 2) + dec.decode_device(None, n_steps - n_steps // 2)
    else:
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


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_device_feedback_gloo(world):
    """decode_device: the arg-max word goes from the last stage's buffer into the first stage's token word by send/recv, no
    per-token host value; tokens equal the single-stage run; the middle ranks learn nothing."""
    n_layers, n_steps, prompt = 7, 11, (3, 9, 27)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_layers, n_steps, q, prompt, 0, True)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _reference(n_layers, 7, n_steps, prompt)
    assert results[0] == want and results[world - 1] == want
    for r in range(1, world - 1):
        assert results[r] == []


def test_single_stage_device_feedback_matches_step_loop():
    pl = _pipeline()

    class _NoComm:
        def send(self, *a):
            raise AssertionError("a single stage sends nothing")
        recv = send
    dec = pl.PipelineDecoder(FakeStage(0, 5, True), 0, 1, _NoComm())
    assert dec.decode_device(7, 9) == _reference(5, 7, 9)


def test_bench_self_launches_two_ranks_on_cpu():
    """`python bench.py --gpus 2` with no launcher around it (the driver's form): the GPU-free parent starts the
    torch.distributed.run child itself, the two ranks run the hop protocol over gloo with the fake stage, rank 0's single
    JSON line comes back through the parent, and a failing child makes the parent exit non-zero."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "5", "--warmup", "2", "--prompt", "6", "--fake-stage"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["warmup"] == 2 and rec["unit"] == "tokens/s" and rec["value"] > 0
    assert rec["repetitions"]["n"] == 3 and "NOT a measurement" in rec["data"]
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "5", "--fake-stage", "--model", "no-such-model", "--prompt", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and not bad.stdout.strip()
