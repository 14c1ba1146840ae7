"""Differential soak of erasure decoding: the same seeded batch (random erasure counts 0 .. 2t + 2, errors up to and
beyond the capability, clean frames) through whatever path the environment selects -- run once with
CC_AMD_PLANES_MIN_WORK=0 (bit-plane chain) and once with CC_AMD_PLANES_MIN_WORK=1000000000000 (one wavefront per
frame) and compare the printed digests.   python erasure_soak.py [log2 frames]"""
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
B = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 17)
for fam, t, hi in (("rs", 16, 256), ("rs", 5, 256), ("bch", 4, 2), ("bch", 9, 2)):
    for tagname, tag in (("BM", cc.berlekamp_massey_tag()), ("EUKLID", cc.euklid_tag())):
        g = torch.Generator(device=dev)
        g.manual_seed(1000 * t + (7 if fam == "rs" else 3))
        code = (cc.rs if fam == "rs" else cc.primitive_bch)(8, cc.errors(t), tag)
        n, t2 = code.n, 2 * code.t
        msg = torch.randint(0, hi, (B, code.l), dtype=torch.uint8, device=dev, generator=g)
        cw = code.encode_batch(msg)
        rho = torch.randint(0, t2 + 3, (B,), device=dev, generator=g)
        rho[::7] = 0
        room = torch.clamp((t2 - rho) // 2, min=0)
        nerr = (torch.rand(B, device=dev, generator=g) * (room + 2).float()).long()
        order = torch.rand((B, n), device=dev, generator=g).argsort(dim=1)
        col = torch.arange(n, device=dev)[None, :]
        is_er = col < rho[:, None]
        is_err = (col >= rho[:, None]) & (col < (rho + nerr)[:, None])
        rx = cw.clone()
        noise = torch.randint(1, hi, (B, n), dtype=torch.uint8, device=dev, generator=g)
        junk = torch.randint(0, hi, (B, n), dtype=torch.uint8, device=dev, generator=g)
        flat_err = torch.zeros((B, n), dtype=torch.uint8, device=dev).scatter_(1, order, torch.where(is_err, noise, torch.zeros_like(noise)))
        rx ^= flat_err
        er_mask = torch.zeros((B, n), dtype=torch.bool, device=dev).scatter_(1, order, is_er)
        rx = torch.where(er_mask, junk, rx)
        rx[5::11] = cw[5::11]  # clean frames that carry erasures
        pos = torch.nonzero(er_mask)[:, 1].to(torch.int16).contiguous()
        off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        off[1:] = torch.cumsum(rho, 0).to(torch.int32)
        out = torch.empty_like(rx)
        ne = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        rc = lib.cc_correct_hard_batch_dev(code._h, vp(rx), vp(pos), vp(off), vp(out), vp(ne), vp(st), B, sh)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for x in (out, ne, st):
            h.update(x.cpu().numpy().tobytes())
        print("%s(255,%d) %-6s rc %d  ok %d  locator %d  recheck %d  erasures %d  digest %s" % (
            fam.upper(), code.l, tagname, rc, int((st == 0).sum()), int((st == 2).sum()), int((st == 3).sum()),
            int((st == 4).sum()), h.hexdigest()[:16]), flush=True)
