"""Differential soak of the self-correcting min-sum variants: the same seeded channel values through the diagonal
kernel (default; SCMS1 / SCMS2 keep q and recompute r on the large geometries, E38) and through the generic kernel
(CC_AMD_FORCE_GENERIC=1): run once each and compare the printed digests (hard decisions, iteration index, status and
a-posteriori values of every frame).   python scms_soak.py [log2 frames]"""
import ctypes as C
import hashlib
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
B = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 15)
for q, t in ((8, 3), (7, 4), (8, 4), (7, 2), (6, 4)):
    for name, tag in (("SCMS1", cc.self_correcting_1_min_sum_tag(20)), ("SCMS2", cc.self_correcting_2_min_sum_tag(20))):
        for rule in (1, 2):
            code = cc.primitive_bch(q, cc.errors(t), tag, stop_rule=rule)
            n = code.n
            g = torch.Generator(device=dev)
            g.manual_seed(100 * q + t)
            y = torch.empty((B, n), dtype=torch.float32, device=dev)
            third = B // 3
            for k, ebno in enumerate((2.0, 4.5, 7.0)):
                lo, hi = k * third, (B if k == 2 else (k + 1) * third)
                y[lo:hi].normal_(1.0, float(code.sigma(ebno)), generator=g)
            hard = torch.empty((B, n), dtype=torch.uint8, device=dev)
            L = torch.empty((B, n), dtype=torch.float32, device=dev)
            it = torch.empty(B, dtype=torch.int16, device=dev)
            st = torch.empty(B, dtype=torch.int32, device=dev)
            rc = lib.cc_correct_soft_batch_dev(code._h, vp(y), None, None, vp(hard), vp(L), vp(it), vp(st), B, sh)
            torch.cuda.synchronize()
            h = hashlib.sha256()
            for x in (hard, it, st, L):
                h.update(x.cpu().numpy().tobytes())
            print("BCH(%d,%d) %s O%d rc %d converged %d  %s  digest %s" % (n, code.l, name, rule, rc, int((st == 0).sum()),
                                                                       code.kernel_info()["kernel"][:28], h.hexdigest()[:16]), flush=True)
