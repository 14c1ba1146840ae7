"""Hard-decision decode and encode throughput for the registry's BCH codes and a few RS codes (frames in HBM)."""
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
sh = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(11)
B = 1 << 19


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("%-20s %-8s %12s %12s %12s  %s" % ("code", "alg", "dec Mfr/s", "enc Mfr/s", "dec GB/s", "kernel"))
cases = [(cc.primitive_bch, q, t, 2) for q in (5, 6, 7, 8) for t in (1, 2, 3, 4)] + \
        [(cc.rs, 4, 3, 16), (cc.rs, 6, 8, 64), (cc.rs, 8, 4, 256), (cc.rs, 8, 8, 256), (cc.rs, 8, 16, 256), (cc.rs, 8, 32, 256)]
for cls, q, t, hi in cases:
    for name, tag in (("BM", cc.berlekamp_massey_tag()), ("EUKLID", cc.euklid_tag())):
        if name == "EUKLID" and t > 31:
            continue
        code = cls(q, cc.errors(t), tag)
        msg = torch.randint(0, hi, (B, code.l), dtype=torch.uint8, device=dev, generator=g)
        cw = code.encode_batch(msg)
        nerr = torch.randint(0, t + 1, (B,), device=dev, generator=g)
        perm = torch.rand((B, code.n), device=dev, generator=g).argsort(dim=1)[:, :t]
        vals = torch.randint(1, hi, (B, t), dtype=torch.uint8, device=dev, generator=g)
        vals = torch.where(torch.arange(t, device=dev)[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
        rx = cw.clone()
        rx.scatter_(1, perm, rx.gather(1, perm) ^ vals)
        out = torch.empty_like(rx)
        ne = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        ms = timed(lambda: lib.cc_correct_hard_batch_dev(code._h, vp(rx), None, None, vp(out), vp(ne), vp(st), B, sh))
        ok = bool(torch.equal(out, cw))
        enc = timed(lambda: lib.cc_encode_batch_dev(code._h, vp(msg), vp(cw), B, sh)) if name == "BM" else float("nan")
        print("%-20s %-8s %12.1f %12.1f %12.1f  %s%s" % (code.to_string().rsplit("-", 1)[0], name, B / ms / 1e3, B / enc / 1e3,
                                                        (2 * code.n + 8) * B / ms / 1e6, code.kernel_info()["kernel"][:24],
                                                        "" if ok else "  NOT ALL CORRECTED"), flush=True)
