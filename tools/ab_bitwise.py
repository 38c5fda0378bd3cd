"""A/B: two engine configurations (environment of the second given as KEY=VAL args) must produce identical logits and greedy
tokens.  usage: ab_bitwise.py <model> <layers> <mix> KEY=VAL ...   (the first engine is built with the current environment)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
name, layers, mix = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kw = dict(num_layers=layers) if layers else {}
cfg = pkg.make_config(name, max_seq_len=320, **kw)
model = pkg.SynthModel(cfg, mix=mix)
a = pkg.HipGpuInference.from_model(model, 320)
for kv in sys.argv[4:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
b = pkg.HipGpuInference.from_model(model, 320)
bad = 0
for i in range(200):
    t = (37 * i + 5) % cfg.vocab_size
    x, y = a.forward(t), b.forward(t)
    if not np.array_equal(x, y):
        bad += 1
        if bad < 4:
            print("step", i, "differs", float(np.abs(x - y).max()), flush=True)
ta, tb = a.decode_greedy(9, 100).tolist(), b.decode_greedy(9, 100).tolist()
print(name, layers, mix, sys.argv[4:], "logit mismatches", bad, "greedy equal", ta == tb, flush=True)
