#!/usr/bin/env python3
"""Golden vectors for GF(2^q), q > 8 (16-bit symbols) from the REAL reference (oracle/_ref, built by oracle/Makefile
from /root/reference; this script only runs in the dev container).  Output: tests/golden/wide.npz.

Codes: wide id 0 = primitive_bch<9, errors<3>> (modular polynomial 0x211), id 1 = rs<10, errors<4>> (0x409).
Per code: constants (g, h, roots, n/k/l/t/dmin, to_string), encode vectors, and hard decoding with PGZ / BM / Euklid of
frames carrying 0 .. t + 2 random symbol errors; for BM / Euklid (and PGZ on the BCH code) a second set with erasures
(positions zeroed)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from checkers import BM, EUKLID, PGZ, RefWide  # noqa: E402

assert RefWide.available(), "build oracle/_ref first (make -C oracle ref)"
out = {}
for wid, (fam, q, t, poly) in RefWide.CODES.items():
    r = RefWide(wid)
    rng = np.random.default_rng(900 + wid)
    hi = 2 if fam == 0 else r.n + 1
    p = "w%d_" % wid
    out[p + "params"] = np.array([fam, q, t, poly, r.n, r.k, r.l, r.dmin], np.int64)
    out[p + "g"], out[p + "h"], out[p + "roots"] = r.poly(0), r.poly(1), r.poly(2)
    out[p + "names"] = np.array([r.to_string(a) for a in (PGZ, BM, EUKLID)])
    frames = 32
    msg = rng.integers(0, hi, (frames, r.l)).astype(np.uint16)
    msg[0] = 0
    cw = r.encode(msg)
    out[p + "msg"], out[p + "cw"] = msg, cw
    rx = cw.copy()
    for f in range(frames):
        for pos in rng.choice(r.n, int(rng.integers(0, t + 3)), replace=False):
            rx[f, pos] ^= 1 if fam == 0 else int(rng.integers(1, hi))
    out[p + "rx"] = rx
    for alg, name in ((PGZ, "pgz"), (BM, "bm"), (EUKLID, "euklid")):
        o, st, _ = r.correct(alg, rx)
        out[p + name + "_out"], out[p + name + "_status"] = o, st
        d, std, _ = r.correct(alg, rx, decode=True)
        out[p + name + "_msg"] = d
        assert np.array_equal(st, std)
    # erasures: e erased positions (zeroed) + up to (2t - e) / 2 + 1 errors elsewhere
    rxe = cw.copy()
    per = []
    for f in range(frames):
        ne = int(rng.integers(0, 2 * t + 2))
        er = sorted(rng.choice(r.n, ne, replace=False).tolist())
        for e in er:
            rxe[f, e] = 0
        free = np.setdiff1d(np.arange(r.n), er)
        for pos in rng.choice(free, int(rng.integers(0, max(1, (2 * t - ne) // 2 + 2))), replace=False):
            rxe[f, pos] ^= 1 if fam == 0 else int(rng.integers(1, hi))
        per.append(er)
    out[p + "rxe"] = rxe
    out[p + "er_off"] = np.concatenate([[0], np.cumsum([len(e) for e in per])]).astype(np.uint32)
    out[p + "er"] = np.array([e for l_ in per for e in l_], np.uint16)
    # (PGZ with erasures exists for BCH only: the two-trial rule of bch.h:97-149; RS throws, hard_decision.h:66-68)
    for alg, name in ((BM, "bm"), (EUKLID, "euklid")) + (((PGZ, "pgz"),) if fam == 0 else ()):
        o, st, _ = r.correct(alg, rxe, erasures=per)
        out[p + name + "_e_out"], out[p + name + "_e_status"] = o, st
np.savez_compressed(os.path.join(HERE, "wide.npz"), **out)
print("wrote wide.npz:", {k: v.shape for k, v in out.items() if k.endswith("status")})
