"""Hard decoding of BCH(255,231) / BCH(255,247) / RS(255,223) over the batch size: frames/s per call size (device
pointers).  Used with an experiment library and CC_EXP_MIN_T2=8 / 2 to place the size switch of launch_algebraic."""
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
g.manual_seed(5)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for cls, t, hi in ((cc.primitive_bch, 3, 2), (cc.primitive_bch, 1, 2), (cc.rs, 16, 256)):
    code = cls(8, cc.errors(t), cc.berlekamp_massey_tag())
    line = "%-16s" % ("%s(255,%d)" % ("BCH" if hi == 2 else "RS", code.l))
    for lg in range(10, 21, 2):
        B = 1 << lg
        msg = torch.randint(0, hi, (B, code.l), dtype=torch.uint8, device=dev, generator=g)
        rx = code.encode_batch(msg)
        pos = torch.randint(0, 255, (B, t), device=dev, generator=g)
        val = torch.randint(1, hi, (B, t), dtype=torch.uint8, device=dev, generator=g)
        rx.scatter_(1, pos, rx.gather(1, pos) ^ val)
        out = torch.empty_like(rx)
        ne = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        ms = timed(lambda: lib.cc_correct_hard_batch_dev(code._h, vp(rx), None, None, vp(out), vp(ne), vp(st), B, sh),
                   200 if lg < 16 else 20)
        line += "  2^%d: %7.1f M (%6.1f us)" % (lg, B / ms / 1e3, ms * 1e3)
    print(line, flush=True)
