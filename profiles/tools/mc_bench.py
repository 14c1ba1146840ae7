"""Throughput of the on-device Monte-Carlo pipeline (cc_mc_run_dev: AWGN -> LLR -> decode -> count)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import channelcoding_amd as cc
from channelcoding_amd.montecarlo import DeviceBackend

code = cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(20))
for rc in (False, True):
    be = DeviceBackend(code, random_codewords=rc)
    for ebno in (2.0, 4.0, 6.0, 8.0):
        frames = 1 << 22
        be.run(ebno, 0, 0, 1 << 16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = be.run(ebno, 0, 0, frames)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("random codewords=%s  %.1f dB: %.2f M frames/s  (%.1f ms for 2^22 frames)  wer=%.4g" % (
            rc, ebno, frames / dt / 1e6, dt * 1e3, int(c[1]) / int(c[0])), flush=True)

# hard decoding in the same pipeline (the reference's simulation runs its hard decoders over the same channel)
for name, tag in (("BM", cc.berlekamp_massey_tag()), ("PGZ", cc.peterson_gorenstein_zierler_tag())):
    be = DeviceBackend(cc.primitive_bch(8, cc.errors(3), tag), random_codewords=True)
    for ebno in (4.0, 8.0):
        frames = 1 << 22
        be.run(ebno, 0, 0, 1 << 16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = be.run(ebno, 0, 0, frames)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("hard %-3s random codewords=True  %.1f dB: %.2f M frames/s  (%.1f ms for 2^22 frames)  wer=%.4g" % (
            name, ebno, frames / dt / 1e6, dt * 1e3, int(c[1]) / int(c[0])), flush=True)
