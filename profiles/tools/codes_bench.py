"""Min-sum decode throughput (frames resident in HBM) for the BCH codes of the reference's benchmark registry
(benchmark.c++:23-166: k = 5..7, dmin<3..9>) plus the n = 255 codes: MS<50> and SCMS1<50>, all-zero word at 4 dB."""
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
g.manual_seed(7)
B = 1 << 18


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


print("%-22s %-44s %10s %8s %8s" % ("code", "kernel", "Mframes/s", "GB/s", "iters"))
for q, ts in ((5, (1, 2, 3, 4)), (6, (1, 2, 3, 4)), (7, (1, 2, 3, 4)), (8, (1, 2, 3, 4))):
    for t in ts:
        tags = (cc.min_sum_tag(50), cc.self_correcting_1_min_sum_tag(50))
        if len(sys.argv) > 1 and sys.argv[1] == "scms":  # the two self-correcting variants only
            tags = (cc.self_correcting_1_min_sum_tag(50), cc.self_correcting_2_min_sum_tag(50))
        for tag in tags:
            code = cc.primitive_bch(q, cc.errors(t), tag)
            n = code.n
            y = torch.empty((B, n), dtype=torch.float32, device=dev).normal_(1.0, float(code.sigma(4.0)), generator=g)
            hard = torch.empty((B, n), dtype=torch.uint8, device=dev)
            it = torch.empty(B, dtype=torch.int16, device=dev)
            st = torch.empty(B, dtype=torch.int32, device=dev)
            ms = timed(lambda: lib.cc_correct_soft_batch_dev(code._h, vp(y), None, None, vp(hard), None, vp(it), vp(st), B, sh))
            run = torch.where(st == 0, it.to(torch.int32) + 1, it.to(torch.int32)).float().mean().item()
            print("%-22s %-44s %10.1f %8.1f %8.2f" % (code.to_string(), code.kernel_info()["kernel"][:44], B / ms / 1e3,
                                                      (5 * n + 6) * B / ms / 1e6, run), flush=True)
