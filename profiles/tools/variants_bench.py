import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, channelcoding_amd as cc
from channelcoding_amd import capi
import ctypes as C
dev = torch.device("cuda", 0)
B = 1 << 20
for tag, name in ((cc.self_correcting_2_min_sum_tag(20), "SCMS2"), (cc.self_correcting_1_min_sum_tag(20), "SCMS1"), (cc.offset_min_sum_tag(20, 0.01), "OMS"), (cc.normalized_2d_min_sum_tag(20, 0.8, 0.9), "2DNMS")):
    code = cc.primitive_bch(8, cc.errors(3), tag)
    sigma = code.sigma(4.0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    y = 1.0 + sigma * torch.randn((B, code.n), device=dev, generator=g)
    r = code.correct_batch(y); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): r = code.correct_batch(y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(name, "%.1f M frames/s" % (B / dt / 1e6), code.kernel_info()["kernel"][:60], flush=True)
