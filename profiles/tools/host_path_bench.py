"""PCIe-inclusive rate of the host-pointer entry point cc_correct_soft_batch (never the headline `value`):
caller-owned, pre-touched buffers passed straight to the C ABI -- pageable and page-locked."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import channelcoding_amd as cc
from channelcoding_amd import capi

code = cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(20))
lib = capi.lib()
rng = np.random.default_rng(0)
P = lambda a: a.ctypes.data_as(C.c_void_p)


def run(y, hard, iters, status, B, reps=3):
    lib.cc_correct_soft_batch(code._h, P(y), None, None, P(hard), None, P(iters), P(status), B)
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = lib.cc_correct_soft_batch(code._h, P(y), None, None, P(hard), None, P(iters), P(status), B)
        assert rc == 0
    return (time.perf_counter() - t0) / reps


for log2b, ebno in ((10, 4.0), (14, 4.0), (16, 4.0), (18, 4.0), (20, 4.0), (20, 8.0)):
    B = 1 << log2b
    y = (1.0 + code.sigma(ebno) * rng.standard_normal((B, 255), dtype=np.float32)).astype(np.float32)
    hard = np.ones((B, 255), np.uint8)
    iters = np.ones(B, np.uint16)
    status = np.ones(B, np.int32)
    dt = run(y, hard, iters, status, B)
    print("pageable buffers, B=2^%d, %.0f dB: %.2f M frames/s  (%.2f ms, %.2f GB/s of LLR in)" % (
        log2b, ebno, B / dt / 1e6, dt * 1e3, B * 1020 / dt / 1e9), flush=True)
    if log2b == 20:
        yp = torch.from_numpy(y).pin_memory().numpy()
        hp = torch.from_numpy(hard).pin_memory().numpy()
        ip = torch.from_numpy(iters.view(np.int16)).pin_memory().numpy()
        sp = torch.from_numpy(status).pin_memory().numpy()
        dt = run(yp, hp, ip, sp, B)
        print("page-locked buffers, B=2^%d, %.0f dB: %.2f M frames/s  (%.2f ms, %.2f GB/s of LLR in)" % (
            log2b, ebno, B / dt / 1e6, dt * 1e3, B * 1020 / dt / 1e9), flush=True)
