"""Mutation fuzz of the GGUF header parser (lgh_gguf_inspect / lgh_gguf_get) on a library built with ASan + UBSan:
bit flips, random bytes and truncations in the header / metadata / tensor-info region of a valid file.  tools/sanitize/run.sh"""
import ctypes as C, os, sys, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import __graft_entry__ as g
pkg = g.load_package()
from write_gguf import write_gguf
hb = pkg.hip_backend
lib = C.CDLL(os.environ.get("LGH_GGUF_LIB", "/tmp/libgguf_asan.so"))
lib.lgh_gguf_inspect.argtypes = [C.c_char_p, C.POINTER(hb.GgufInfo), C.c_char_p, C.c_size_t]
lib.lgh_gguf_get.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(hb.GgufValue), C.c_char_p, C.c_size_t]
cfg = pkg.make_config("test-dense", max_seq_len=32)
model = pkg.SynthModel(cfg, mix="Q4_K_M")
path = "/tmp/fz_base.gguf"
write_gguf(path, cfg, model.tensors(), arch="llama")
blob = bytearray(open(path, "rb").read())
hdr = 6000          # the header + metadata + tensor infos live at the front; mutate there
rnd = random.Random(7)
info, val = hb.GgufInfo(), hb.GgufValue()
err = C.create_string_buffer(512)
n_ok = n_bad = 0
for it in range(3000):
    b = bytearray(blob[: max(64, len(blob) if it % 3 else rnd.randrange(16, min(len(blob), hdr)))])
    for _ in range(rnd.randrange(1, 4)):
        k = rnd.randrange(0, min(len(b), hdr))
        b[k] = rnd.randrange(256) if it % 2 else (b[k] ^ (1 << rnd.randrange(8)))
    p = "/tmp/fz_case.gguf"
    open(p, "wb").write(b)
    rc = lib.lgh_gguf_inspect(p.encode(), C.byref(info), err, 512)
    lib.lgh_gguf_get(p.encode(), b"general.architecture", C.byref(val), err, 512)
    lib.lgh_gguf_get(p.encode(), b"llama.block_count", C.byref(val), err, 512)
    n_ok += rc == 0
    n_bad += rc != 0
print("cases", n_ok + n_bad, "accepted", n_ok, "rejected", n_bad)
