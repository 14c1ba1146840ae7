"""PCIe-inclusive rate of the host-pointer entry point cc_correct_soft_batch (never the headline `value`)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import channelcoding_amd as cc

code = cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(20))
rng = np.random.default_rng(0)
for log2b, ebno in ((16, 4.0), (18, 4.0), (20, 4.0), (20, 8.0)):
    B = 1 << log2b
    y = (1.0 + code.sigma(ebno) * rng.standard_normal((B, 255), dtype=np.float32)).astype(np.float32)
    code.correct_batch(y[:1024])
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        res = code.correct_batch(y)
    dt = (time.perf_counter() - t0) / reps
    print("host buffers, B=2^%d, %.0f dB: %.2f M frames/s  (%.1f ms, %.2f GB/s of LLR in)" % (
        log2b, ebno, B / dt / 1e6, dt * 1e3, B * 1020 / dt / 1e9), flush=True)
