import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kw = dict(num_layers=layers) if layers else {}
cfg = pkg.make_config(name, max_seq_len=256, **kw)
model = pkg.SynthModel(cfg, mix=sys.argv[3] if len(sys.argv) > 3 else "Q4_K_M")
eng = pkg.HipGpuInference.from_model(model, 256, flags=pkg.hip_backend.FLAG_OVERLAP)
ref = pkg.HipGpuInference.from_model(model, 256)
try:
    for i in range(80):
        a = eng.forward(i % cfg.vocab_size)
        b = ref.forward(i % cfg.vocab_size)
        if not np.array_equal(a, b):
            print("step", i, "differs", float(np.abs(a - b).max()), flush=True)
    print("forward ok", flush=True)
    ta = eng.decode_greedy(5, 64).tolist(); tb = ref.decode_greedy(5, 64).tolist()
    print("greedy equal", ta == tb, flush=True)
except Exception as e:
    print("FAILED at position", eng.position(), e, flush=True)
