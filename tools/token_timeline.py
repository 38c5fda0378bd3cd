#!/usr/bin/env python3
"""Per-node timeline of one decode step of the DEFAULT graph, from in-kernel stamps (csrc/timeline.h; diagnostic build:
`make -C llama-gguf_amd stamps`, selected here through LGH_LIB_VARIANT=stamps).

    python tools/token_timeline.py [--model llama-3-8b] [--mix Q4_K_M] [--kv 128] [--steps 12]

Every workgroup of every kernel on the decode path stamps s_memrealtime (100 MHz) at its start and end.  Per graph node:
  gap   last workgroup of the previous node ended -> first workgroup of this node started (the kernel boundary as the
        shader sees it: end-of-kernel release, dispatch, wave launch, the first instructions up to the stamp)
  ramp  first -> last workgroup start            body  median over workgroups of (end - start)
  tail  first -> last workgroup end              span  first start -> last end
Printed: one layer node by node, and per node class the mean over the layers (first two skipped)."""
import argparse
import ctypes as C
import os
import sys

os.environ.setdefault("LGH_LIB_VARIANT", "stamps")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as graft

KIND = {1: "embed", 2: "mvq", 3: "attn", 4: "combine", 5: "argmax1", 6: "argmax2", 7: "advance", 8: "xq", 9: "mv", 10: "other"}
SLOTS, WG = 1024, 320


def read_nodes(lib):
    size = lib.lgh_debug_timeline(0, None)
    assert size == SLOTS * WG * (16 + 16 + 4) + SLOTS * 8 + 4 * SLOTS * 4, size
    nodes = []
    for which in range(4):
        buf = (C.c_ubyte * size)()
        assert lib.lgh_debug_timeline(which, buf) == 0
        raw = np.frombuffer(buf, dtype=np.uint8)
        t = raw[: SLOTS * WG * 16].view(np.uint64).reshape(SLOTS, WG, 2).astype(np.int64)
        o = SLOTS * WG * 16
        clk = raw[o: o + SLOTS * WG * 16].view(np.uint64).reshape(SLOTS, WG, 2).astype(np.int64)
        o += SLOTS * WG * 16
        xcc = raw[o: o + SLOTS * WG * 4].view(np.uint32).reshape(SLOTS, WG)
        o += SLOTS * WG * 4
        packet = raw[o: o + SLOTS * 8].view(np.uint64)
        o += SLOTS * 8
        kind = raw[o: o + SLOTS * 4].view(np.uint32); o += SLOTS * 4
        grid = raw[o: o + SLOTS * 4].view(np.uint32); o += SLOTS * 4
        aux = raw[o: o + SLOTS * 4].view(np.uint32)
        for s in range(SLOTS):
            if kind[s] == 0:
                continue
            n = min(int(grid[s]), WG)
            t0, t1 = t[s, :n, 0], t[s, :n, 1]
            if (t1 < t0).any() or t0.min() == 0:
                continue   # a slot caught between two dispatches
            nodes.append(dict(kind=KIND.get(int(kind[s]), "?"), grid=int(grid[s]), aux=int(aux[s]), packet=int(packet[s]),
                              t0=t0.copy(), t1=t1.copy(), c0=clk[s, :n, 0].copy(), c1=clk[s, :n, 1].copy(), xcc=(xcc[s, :n] & 0xF).copy()))
    nodes.sort(key=lambda d: d["t0"].min())
    return nodes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--mix", default="Q4_K_M")
    ap.add_argument("--kv", type=int, default=128)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--layer", type=int, default=5, help="layer printed node by node")
    ap.add_argument("--stages", type=int, default=0, help="the in-library pipeline with this many stages on device 0 (prints the gaps > 3 us of a token)")
    a = ap.parse_args()
    pkg = graft.load_package()
    cfg = pkg.make_config(a.model, max_seq_len=a.kv + a.steps + 64)
    lib = pkg.hip_backend.load_library()
    lib.lgh_debug_timeline.restype = C.c_longlong
    lib.lgh_debug_timeline.argtypes = [C.c_int, C.c_void_p]
    if a.stages:
        eng = pkg.HipPipeline.from_model(pkg.SynthModel(cfg, mix=a.mix), a.kv + a.steps + 64, a.stages, devices=[0] * a.stages)
        for t in range(a.kv):
            eng.prefill_token(t % cfg.vocab_size)
        eng.decode_greedy(5, a.steps)
        nodes = read_nodes(lib)
        emb = [i for i, d in enumerate(nodes) if d["kind"] == "embed"]
        tok = nodes[emb[-3]: emb[-2] + 1]
        t0 = tok[0]["t0"].min()
        print(f"# {a.model} {a.mix}, {a.stages} stages on one device: one token embed -> next embed {(tok[-1]['t0'].min() - t0) / 100.0:.1f} us, {len(tok) - 1} stamped nodes")
        prev, idle = None, 0.0
        for d in tok:
            s0, e0 = d["t0"].min(), d["t1"].max()
            if prev is not None:
                gap = (s0 - prev) / 100.0
                idle += max(gap - 1.6, 0.0)
                if gap > 3.0:
                    print(f"  +{(s0 - t0) / 100.0:9.2f} us  gap {gap:7.2f} before {d['kind']} (grid {d['grid']})")
            prev = e0
        print(f"  idle beyond 1.6 us per boundary: {idle:.1f} us")
        return
    eng = pkg.HipGpuInference.from_model(pkg.SynthModel(cfg, mix=a.mix), a.kv + a.steps + 64)
    eng.forward_batch([t % cfg.vocab_size for t in range(a.kv)])
    eng.decode_greedy(5, a.steps)
    nodes = read_nodes(lib)
    # the last complete token: from its embed node to its argmax2 node
    ends = [i for i, d in enumerate(nodes) if d["kind"] == "argmax2"]
    starts = [i for i, d in enumerate(nodes) if d["kind"] == "embed"]
    assert ends and starts, "no complete token in the buffers"
    e = ends[-1]
    s = max(i for i in starts if i < e)
    tok = nodes[s: e + 1]
    # label the mat-vec nodes by their position inside a layer
    names, k = [], 0
    per_layer = ["qkv", "attn", "combine", "wo", "gate_up", "down"]
    for d in tok:
        if d["kind"] in ("mvq", "mv", "attn", "combine"):
            li, j = divmod(k, 6)
            nm = per_layer[j] if li < cfg.num_layers else "output"
            if li < cfg.num_layers and ((j in (1, 2)) != (d["kind"] in ("attn", "combine"))):
                nm = d["kind"] + "?"
            names.append((li if li < cfg.num_layers else -1, nm))
            k += 1
        else:
            names.append((-1, d["kind"]))
    prev_end = None
    rows = []
    for (li, nm), d in zip(names, tok):
        t0, t1 = d["t0"], d["t1"]
        gap = (t0.min() - prev_end) / 100.0 if prev_end is not None else float("nan")
        rows.append(dict(layer=li, name=nm, grid=d["grid"], gap=gap, ramp=(t0.max() - t0.min()) / 100.0,
                         body=float(np.median(t1 - t0)) / 100.0, tail=(t1.max() - t1.min()) / 100.0, span=(t1.max() - t0.min()) / 100.0))
        prev_end = t1.max()
    total = (tok[-1]["t1"].max() - tok[0]["t0"].min()) / 100.0
    print(f"# {a.model} {a.mix}, kv {a.kv}+{a.steps}: last token {len(tok)} nodes, {total:.1f} us first start -> last end "
          f"(diagnostic build: the stamps cost a store per workgroup; read SHARES)")
    print(f"\n## layer {a.layer}, node by node (us)\n| node | grid | gap | ramp | body (median wg) | tail | span | gap+span |\n|---|---|---|---|---|---|---|---|")
    for r in rows:
        if r["layer"] == a.layer:
            print(f"| {r['name']} | {r['grid']} | {r['gap']:.2f} | {r['ramp']:.2f} | {r['body']:.2f} | {r['tail']:.2f} | {r['span']:.2f} | {r['gap'] + r['span']:.2f} |")
    print(f"\n## mean over layers 2..{cfg.num_layers - 1} (us)\n| node | gap | ramp | body | tail | span | gap+span |\n|---|---|---|---|---|---|---|")
    layer_total = 0.0
    for nm in per_layer:
        sel = [r for r in rows if r["name"] == nm and r["layer"] >= 2]
        if not sel:
            continue
        m = {k2: float(np.mean([r[k2] for r in sel])) for k2 in ("gap", "ramp", "body", "tail", "span")}
        layer_total += m["gap"] + m["span"]
        print(f"| {nm} | {m['gap']:.2f} | {m['ramp']:.2f} | {m['body']:.2f} | {m['tail']:.2f} | {m['span']:.2f} | {m['gap'] + m['span']:.2f} |")
    print(f"| layer | | | | | | {layer_total:.2f} |")
    # where does the tail come from?  per-workgroup end times of the gate_up node of the printed layer, by XCD (blockIdx % 8)
    # and by dispatch order (blockIdx / 8 = the order in which an XCD receives its workgroups)
    # is the skew between XCDs stable?  per layer: the median end time of each physical XCD's workgroups in gate_up, and
    # the shader clock each XCD ran at (d s_memtime / d s_memrealtime)
    print("\n## gate_up per layer: median workgroup end (us since first start) by PHYSICAL XCD (HW_REG_XCC_ID), then shader clock GHz by XCD")
    for (li, nm), d in zip(names, tok):
        if nm == "gate_up" and li % 4 == 1:
            base = d["t0"].min()
            e = (d["t1"] - base) / 100.0
            ghz = (d["c1"] - d["c0"]) / np.maximum(d["t1"] - d["t0"], 1) * 0.1
            print(f"layer {li:2d}: end " + " ".join(f"{np.median(e[d['xcc'] == x]):6.2f}" if (d['xcc'] == x).any() else "   -  " for x in range(8))
                  + "   clock " + " ".join(f"{np.median(ghz[d['xcc'] == x]):5.2f}" if (d['xcc'] == x).any() else "  -  " for x in range(8))
                  + "   blockIdx%8 of XCD: " + " ".join(str(int(np.bincount((np.nonzero(d['xcc'] == x)[0] % 8), minlength=8).argmax())) if (d['xcc'] == x).any() else "-" for x in range(8)))
    for want in ("gate_up", "wo"):
        for (li, nm), d in zip(names, tok):
            if li == a.layer and nm == want:
                t0, t1 = d["t0"], d["t1"]
                base = t0.min()
                n = len(t0)
                print(f"\n## {want} of layer {a.layer}: workgroup start / end (us since the first start) by XCD = blockIdx % 8\n| XCD | starts min..max | ends min / median / max |\n|---|---|---|")
                for x in range(8):
                    s0 = (t0[x::8] - base) / 100.0
                    e0 = (t1[x::8] - base) / 100.0
                    print(f"| {x} | {s0.min():.2f}..{s0.max():.2f} | {e0.min():.2f} / {np.median(e0):.2f} / {e0.max():.2f} |")
                order = np.argsort(t1)
                print("last 12 workgroups to end (blockIdx: end us, body us): " + ", ".join(f"{int(w)}: {(t1[w] - base) / 100.0:.2f}, {(t1[w] - t0[w]) / 100.0:.2f}" for w in order[-12:]))
                print("first 12 workgroups to end: " + ", ".join(f"{int(w)}: {(t1[w] - base) / 100.0:.2f}" for w in order[:12]))
                q = np.percentile((t1 - base) / 100.0, [0, 10, 25, 50, 75, 90, 95, 99, 100])
                print("end-time percentiles 0/10/25/50/75/90/95/99/100: " + " ".join(f"{v:.2f}" for v in q))
    print("\n## the rest of the token (us)\n| node | grid | gap | ramp | body | tail | span |\n|---|---|---|---|---|---|---|")
    for r in rows:
        if r["layer"] < 0:
            print(f"| {r['name']} | {r['grid']} | {r['gap']:.2f} | {r['ramp']:.2f} | {r['body']:.2f} | {r['tail']:.2f} | {r['span']:.2f} |")


if __name__ == "__main__":
    main()
