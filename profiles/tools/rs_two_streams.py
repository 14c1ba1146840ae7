"""RS(255,223) BM decode of 2^20 frames: one call on one stream against two half-size calls on two streams (do stages
that are bound by different units -- syndromes: VALU, Berlekamp-Massey / corrector: LDS look-ups -- overlap?)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import channelcoding_amd as cc
from channelcoding_amd import capi

dev = torch.device("cuda", 0)
lib = capi.lib()
vp = lambda t: C.c_void_p(t.data_ptr())
g = torch.Generator(device=dev)
g.manual_seed(99)
B = 1 << 20
codes = [cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag()) for _ in range(4)]
code = codes[0]
msg = torch.randint(0, 256, (B, code.l), dtype=torch.uint8, device=dev, generator=g)
cw = code.encode_batch(msg)
nerr = torch.randint(0, 17, (B,), device=dev, generator=g)
perm = torch.rand((B, code.n), device=dev, generator=g).argsort(dim=1)[:, :16]
vals = torch.randint(1, 256, (B, 16), dtype=torch.uint8, device=dev, generator=g)
vals = torch.where(torch.arange(16, device=dev)[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
rx = cw.clone()
rx.scatter_(1, perm, rx.gather(1, perm) ^ vals)
out = torch.empty_like(rx)
ne = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(4)]


def run(parts):
    h = B // parts
    for k in range(parts):
        s = streams[k]
        sl = slice(k * h, (k + 1) * h)
        lib.cc_correct_hard_batch_dev(codes[k]._h, vp(rx[sl]), None, None, vp(out[sl]), vp(ne[sl]), vp(st[sl]), h,
                                      C.c_void_p(s.cuda_stream))


for parts in (1, 2, 4, 1, 2, 4):
    run(parts)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        run(parts)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 5
    ok = bool(torch.equal(out, cw)) and int((st != 0).sum()) == 0
    print("%d stream(s): %.3f ms per 2^20 frames = %.1f M frames/s  all corrected: %s" % (parts, ms, B / ms / 1e3, ok), flush=True)
