#!/usr/bin/env python3
"""One rank of a multi-process layer-pipeline rehearsal on ONE GPU: real stage engines (every rank on device 0), the hop protocol
of pipeline.PipelineDecoder with hidden vectors staged through the host over gloo (pipeline.HostStagedComm).  Rank 0 prints the
greedy tokens as JSON.  Started by tests/test_gpu_model.py with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set:
    pp_rehearse.py <config> <num_layers> <mix> <prompt_len> <steps> [batched]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    name, layers, mix, n_prompt, steps = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    batched = len(sys.argv) > 6 and sys.argv[6] == "batched"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    pkg = graft.load_package()
    import importlib
    pipeline = importlib.import_module(pkg.__name__ + ".pipeline")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = pkg.make_config(name, max_seq_len=n_prompt + steps + 16, num_layers=layers)
    model = pkg.SynthModel(cfg, mix=mix)
    lo, hi = pipeline.split_layers(cfg.num_layers, world)[rank]
    eng = pkg.HipGpuInference.from_model(model, cfg.max_seq_len, device=0, layer_range=(lo, hi))
    stage = pipeline.HipStage(eng, torch, dev)
    if not batched:
        stage.block_tokens = 0
    dec = pipeline.PipelineDecoder(stage, rank, world, pipeline.HostStagedComm(dist))
    prompt = [(31 * i + 7) % cfg.vocab_size for i in range(n_prompt)]
    dec.prefill(prompt[:-1])
    toks = dec.decode_device(prompt[-1], steps)
    more = dec.decode_device(None, 4)                 # continues from the token left on the device
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print(json.dumps({"tokens": toks + more, "position": stage.position()}), flush=True)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
