"""RS(255,223) / BCH hard-decode timing only (BASELINE configs[3]): python rs_bench.py [log2 frames]"""
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
g.manual_seed(99)
B = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(code, hi, maxerr, label):
    msg = torch.randint(0, hi, (B, code.l), dtype=torch.uint8, device=dev, generator=g)
    cw = code.encode_batch(msg)
    nerr = torch.randint(0, maxerr + 1, (B,), device=dev, generator=g)
    perm = torch.rand((B, code.n), device=dev, generator=g).argsort(dim=1)[:, :maxerr]
    vals = torch.randint(1, hi, (B, maxerr), dtype=torch.uint8, device=dev, generator=g)
    vals = torch.where(torch.arange(maxerr, device=dev)[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
    rx = cw.clone()
    rx.scatter_(1, perm, rx.gather(1, perm) ^ vals)
    out = torch.empty_like(rx)
    ne = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    ms = timed(lambda: lib.cc_correct_hard_batch_dev(code._h, vp(rx), None, None, vp(out), vp(ne), vp(st), B, sh))
    ok = bool(torch.equal(out, cw)) and int((st != 0).sum()) == 0
    print("%-28s %8.1f Mframes/s  %7.3f ms  all corrected: %s  [%s]" % (label, B / ms / 1e3, ms, ok,
                                                                       code.kernel_info()["kernel"]), flush=True)


def run_erasures(code, rho, maxerr, label):
    """rho erased positions per frame (symbols zeroed, positions passed as CSR) + 0 .. maxerr errors elsewhere."""
    msg = torch.randint(0, 256, (B, code.l), dtype=torch.uint8, device=dev, generator=g)
    cw = code.encode_batch(msg)
    perm = torch.rand((B, code.n), device=dev, generator=g).argsort(dim=1)[:, :rho + maxerr]
    nerr = torch.randint(0, maxerr + 1, (B,), device=dev, generator=g)
    vals = torch.randint(1, 256, (B, maxerr), dtype=torch.uint8, device=dev, generator=g)
    vals = torch.where(torch.arange(maxerr, device=dev)[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
    rx = cw.clone()
    rx.scatter_(1, perm[:, rho:], rx.gather(1, perm[:, rho:]) ^ vals)
    rx.scatter_(1, perm[:, :rho], torch.zeros((B, rho), dtype=torch.uint8, device=dev))
    er = perm[:, :rho].sort(dim=1).values.to(torch.int16).contiguous().view(-1)
    off = (torch.arange(B + 1, device=dev, dtype=torch.int64) * rho).to(torch.int32)
    out = torch.empty_like(rx)
    ne = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    ms = timed(lambda: lib.cc_correct_hard_batch_dev(code._h, vp(rx), vp(er), vp(off), vp(out), vp(ne), vp(st), B, sh))
    ok = bool(torch.equal(out, cw)) and int((st != 0).sum()) == 0
    print("%-28s %8.1f Mframes/s  %7.3f ms  all corrected: %s" % (label, B / ms / 1e3, ms, ok), flush=True)


run(cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag()), 256, 16, "RS(255,223) BM e<=16")
if len(sys.argv) > 2 and sys.argv[2] == "erasures":
    run_erasures(cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag()), 8, 12, "RS(255,223) BM 8 erasures + e<=12")
    run_erasures(cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag()), 4, 6, "RS(255,223) BM 4 erasures + e<=6")
    run_erasures(cc.rs(8, cc.errors(16), cc.euklid_tag()), 8, 12, "RS(255,223) EUKLID 8 erasures + e<=12")
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "only":  # counter passes: BASELINE configs[3] alone
    sys.exit(0)
run(cc.rs(8, cc.errors(16), cc.euklid_tag()), 256, 16, "RS(255,223) EUKLID e<=16")
run(cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag()), 256, 2, "RS(255,223) BM e<=2")
run(cc.primitive_bch(8, cc.errors(3), cc.berlekamp_massey_tag()), 2, 3, "BCH(255,231) BM e<=3")
