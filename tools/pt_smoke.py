import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
def run(name, mix, n=12, **kw):
    cfg = pkg.make_config(name, max_seq_len=96, **kw)
    model = pkg.SynthModel(cfg, mix=mix)
    ref = orc.Model(cfg.as_dict())
    for nm, t, ne, data in model.tensors(keep=True): ref.add_tensor(nm, t, ne, data)
    ref.finalize()
    a = pkg.HipGpuInference.from_model(model, 96, flags=pkg.hip_backend.FLAG_PERSISTENT)
    b = pkg.HipGpuInference.from_model(model, 96)
    print(name, mix, 'graph nodes pt', a.stats()['graph_nodes'], flush=True)
    toks = [3, 17, 255, 9, 5]
    worst = 0
    for i in range(n):
        t = toks[i % len(toks)] + i
        ga, gb, w = a.forward(t), b.forward(t), ref.forward([t])
        ea, eb = float(np.abs(ga - w).max()), float(np.abs(gb - w).max())
        worst = max(worst, ea)
        print(f'  pos {i}: |pt-orc| {ea:.3e}  |old-orc| {eb:.3e}  |pt-old| {float(np.abs(ga-gb).max()):.3e} nodes {a.stats()["graph_nodes"]}', flush=True)
    tol = 2e-3 * float(np.abs(w).max()) + 2e-3
    print('  worst', worst, 'tol', tol, 'OK' if worst <= tol else 'FAIL', flush=True)
    da = a.decode_greedy(7, 40).tolist(); db = b.decode_greedy(7, 40).tolist()
    print('  greedy equal:', da == db, da[:8], flush=True)
    a.close(); b.close(); ref.close()
for args in [("test-dense", "Q4_K_M"), ("test-dense-d128", "Q4_K_M"), ("test-dense", "Q8_0"), ("test-dense", "Q5_K_M"), ("test-dense", "Q6_K")]:
    run(*args)
