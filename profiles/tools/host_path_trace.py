"""One host-pointer call of 2^20 frames (pinned caller buffers) for a rocprofv3 kernel + memory-copy trace:
    rocprofv3 --kernel-trace --memory-copy-trace -d OUT -- python3 profiles/tools/host_path_trace.py [ebno]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import channelcoding_amd as cc
from channelcoding_amd import capi

ebno = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
pageable = len(sys.argv) > 2 and sys.argv[2] == "pageable"
code = cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(20))
lib = capi.lib()
rng = np.random.default_rng(0)
P = lambda a: a.ctypes.data_as(C.c_void_p)
B = 1 << 20
y = torch.from_numpy((1.0 + code.sigma(ebno) * rng.standard_normal((B, 255), dtype=np.float32)).astype(np.float32)).pin_memory().numpy()
hard = torch.ones((B, 255), dtype=torch.uint8).pin_memory().numpy()
iters = torch.ones(B, dtype=torch.int16).pin_memory().numpy()
status = torch.ones(B, dtype=torch.int32).pin_memory().numpy()
mode = sys.argv[2] if len(sys.argv) > 2 else "pinned"
if mode == "mixed":  # a pageable call first, then the page-locked buffers
    y2, h2, i2, s2 = y.copy(), hard.copy(), iters.copy(), status.copy()
    for rep in range(2):
        t0 = time.perf_counter()
        lib.cc_correct_soft_batch(code._h, P(y2), None, None, P(h2), None, P(i2), P(s2), B)
        print("pageable call %d: %.2f ms" % (rep, (time.perf_counter() - t0) * 1e3), flush=True)
if mode == "alloc":  # new page-locked allocations after the handle exists
    keep = [torch.empty(40 << 20, dtype=torch.uint8).pin_memory() for _ in range(10)]
if mode == "tinypageable":
    y2, h2, i2, s2 = y[:1024].copy(), hard[:1024].copy(), iters[:1024].copy(), status[:1024].copy()
    lib.cc_correct_soft_batch(code._h, P(y2), None, None, P(h2), None, P(i2), P(s2), 1024)
if mode == "small":  # small page-locked calls first
    for b in (1 << 10, 1 << 14, 1 << 18):
        lib.cc_correct_soft_batch(code._h, P(y), None, None, P(hard), None, P(iters), P(status), b)
if pageable:
    y, hard, iters, status = y.copy(), hard.copy(), iters.copy(), status.copy()
print("buffers:", "pageable" if pageable else "page-locked", flush=True)
for rep in range(4):
    t0 = time.perf_counter()
    rc = lib.cc_correct_soft_batch(code._h, P(y), None, None, P(hard), None, P(iters), P(status), B)
    print("call %d: %.2f ms rc %d" % (rep, (time.perf_counter() - t0) * 1e3, rc), flush=True)
