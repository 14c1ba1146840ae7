"""Sweep over every code the constructor accepts for q = 3..8 (BCH and RS, all t): encode, extraction, the three
hard algorithms and plain min-sum against the oracle on a few frames each.  Catches geometry limits (LDS scratch
sizes, lanes per coefficient, state placement) that the fixed test codes do not reach."""
import numpy as np
import pytest

from checkers import BCH, BM, EUKLID, O2, PGZ, RS, Oracle, awgn_llr

import channelcoding_amd as cc
from channelcoding_amd import capi

pytestmark = pytest.mark.gpu
TAGS = {PGZ: cc.peterson_gorenstein_zierler_tag, BM: cc.berlekamp_massey_tag, EUKLID: cc.euklid_tag}


def codes_for(fam, q):
    n = (1 << q) - 1
    seen = set()
    for t in range(1, 33):
        try:
            o = Oracle(fam, q, t)
        except ValueError:
            continue
        if o.l < 1 or (o.k, o.l) in seen:
            continue
        seen.add((o.k, o.l))
        yield t, o
    assert seen, (fam, q, n)


@pytest.mark.parametrize("fam", [BCH, RS])
@pytest.mark.parametrize("q", [3, 4, 5, 6, 7, 8])
def test_every_code_of_the_field(fam, q):
    rng = np.random.default_rng(1000 * fam + q)
    cls = cc.primitive_bch if fam == BCH else cc.rs
    hi = 2 if fam == BCH else 1 << q
    frames = 24 if q == 8 else 48
    for t, o in codes_for(fam, q):
        msg = rng.integers(0, hi, (frames, o.l)).astype(np.uint8)
        cw = o.encode(msg)
        rx = cw.copy()
        for f in range(frames):
            for p in rng.choice(o.n, int(rng.integers(0, min(o.n, o.t + 2))), replace=False):
                rx[f, p] ^= 1 if fam == BCH else int(rng.integers(1, hi))
        want, wnerr, wst, _ = o.correct_hard(BM, rx)
        for alg in (BM, PGZ, EUKLID):
            code = cls(q, cc.errors(t), TAGS[alg]())
            assert (code.n, code.k, code.l, code.dmin) == (o.n, o.k, o.l, o.dmin), (fam, q, t)
            if alg == BM:
                assert np.array_equal(code.encode_batch(msg), cw), (fam, q, t, "encode")
                assert np.array_equal(code.extract_batch(cw), msg), (fam, q, t, "extract")
            try:
                res = code.correct_batch(rx)
            except cc.CcError as e:  # documented limits only: t > 32 syndromes per lane, Euklid needs coefficient 2t
                assert e.status == capi.ERR_UNSUPPORTED and (alg == EUKLID and 2 * o.t > 63), (fam, q, t, alg, str(e))
                continue
            out, nerr, st, _ = o.correct_hard(alg, rx)
            ok = st == 0
            assert np.array_equal(res["status"] == 0, ok), (fam, q, t, alg)
            assert np.array_equal(res["out"][ok], out[ok]) and np.array_equal(res["nerr"][ok], nerr[ok]), (fam, q, t, alg)
            assert np.array_equal(res["out"][~ok], rx[~ok]), (fam, q, t, alg)
        if fam == BCH:
            y = awgn_llr(rng, cw, o.l / o.n, 6.0)
            soft = cc.primitive_bch(q, cc.errors(t), cc.min_sum_tag(5))
            res = soft.correct_batch(y, want_L=True)
            b, L, it, st = o.minsum(0, 5, y, 1.0, 0.0, O2, fast=True)
            assert np.array_equal(res["out"], b) and np.array_equal(res["L"], L), (q, t, soft.kernel_info())
            assert np.array_equal(res["iters"], it) and np.array_equal(res["status"] != 0, st != 0), (q, t)


@pytest.mark.parametrize("fam", [BCH, RS])
@pytest.mark.parametrize("q", [3, 4, 5, 6, 7, 8])
def test_every_code_erasures_multiplication_variants(fam, q):
    """Same sweep for the rarer paths: BM / Euklid with erasures, multiplication_tag coding, and (BCH) the
    self-correcting and normalised min-sum variants, which run the generic kernel (state in LDS or HBM)."""
    rng = np.random.default_rng(2000 * fam + q)
    cls = cc.primitive_bch if fam == BCH else cc.rs
    hi = 2 if fam == BCH else 1 << q
    frames = 16 if q == 8 else 32
    for t, o in codes_for(fam, q):
        msg = rng.integers(0, hi, (frames, o.l)).astype(np.uint8)
        om = Oracle(fam, q, t, coding=1)
        mult = cls(q, cc.errors(t), cc.berlekamp_massey_tag(), coding="multiplication")
        cwm = om.encode(msg)
        assert np.array_equal(mult.encode_batch(msg), cwm), (fam, q, t, "a*g")
        assert np.array_equal(mult.extract_batch(cwm), msg), (fam, q, t, "b/g")
        # erasures: e errors + r erasures with 2e + r <= 2t (and a few frames beyond)
        cw = o.encode(msg)
        r = int(rng.integers(1, min(2 * o.t, 6) + 1))
        er = sorted(int(p) for p in rng.choice(o.n, r, replace=False))
        rx = cw.copy()
        rx[:, er] = 0
        for f in range(frames):
            e = int(rng.integers(0, (2 * o.t - r) // 2 + 2))
            free = [p for p in range(o.n) if p not in er]
            for p in rng.choice(free, min(e, len(free)), replace=False):
                rx[f, p] ^= 1 if fam == BCH else int(rng.integers(1, hi))
        for alg in (BM, EUKLID):
            code = cls(q, cc.errors(t), TAGS[alg]())
            try:
                res = code.correct_batch(rx, erasures=er)
            except cc.CcError as e:
                assert e.status == capi.ERR_UNSUPPORTED and alg == EUKLID and 2 * o.t > 32, (fam, q, t, str(e))
                continue
            out, nerr, st, ub = o.correct_hard(alg, rx, er)
            keep = ub == 0  # frames on which the reference's BM reads out of bounds are undefined there (F3)
            ok = (st == 0) & keep
            assert np.array_equal((res["status"] == 0)[keep], (st == 0)[keep]), (fam, q, t, alg, er)
            assert np.array_equal(res["out"][ok], out[ok]), (fam, q, t, alg, er)
        if fam == BCH and o.k * 4 <= 1200:
            y = awgn_llr(rng, cw, o.l / o.n, 6.0)
            for ov, tag, alpha in ((1, cc.normalized_min_sum_tag(5, 0.8), 0.8), (4, cc.self_correcting_2_min_sum_tag(5), 1.0)):
                soft = cc.primitive_bch(q, cc.errors(t), tag)
                res = soft.correct_batch(y, want_L=True)
                b, L, it, st = o.minsum(ov, 5, y, alpha, 0.0, O2, fast=True)
                assert np.array_equal(res["out"], b) and np.array_equal(res["L"], L), (q, t, ov, soft.kernel_info())
                assert np.array_equal(res["iters"], it) and np.array_equal(res["status"] != 0, st != 0), (q, t, ov)


def test_batch_sizes_are_prefix_consistent():
    """Every kernel family (diagonal, register, generic, HBM-state generic, chunked and per-wavefront algebraic,
    encoder) on ragged batch sizes: decoding the first B frames alone must give the first B rows of the full
    batch (persistent groups, 32-frame chunks and 4-frames-per-workgroup tails)."""
    rng = np.random.default_rng(5)
    sizes = (1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 129, 255, 257, 1000)
    soft = [cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(8)),                  # diagonal kernel
            cc.primitive_bch(6, cc.errors(3), cc.min_sum_tag(8)),                  # register kernel
            cc.primitive_bch(4, cc.errors(2), cc.self_correcting_1_min_sum_tag(8)),  # generic, 4 frames per wave
            cc.primitive_bch(8, cc.errors(20), cc.min_sum_tag(4))]                 # generic, state in HBM
    for code in soft:
        y = (1.0 + code.sigma(5.0) * rng.standard_normal((max(sizes), code.n), dtype=np.float32)).astype(np.float32)
        full = code.correct_batch(y, want_L=True)
        for B in sizes:
            part = code.correct_batch(y[:B], want_L=True)
            for key in ("out", "L", "iters", "status"):
                assert np.array_equal(part[key], full[key][:B]), (code.to_string(), B, key)
    hard = [cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag()),   # chunked kernel
            cc.rs(8, cc.errors(16), cc.euklid_tag()),             # one wavefront per frame
            cc.primitive_bch(8, cc.errors(3), cc.berlekamp_massey_tag())]
    for code in hard:
        hi = 2 if isinstance(code, cc.primitive_bch) else 256
        msg = rng.integers(0, hi, (max(sizes), code.l)).astype(np.uint8)
        cw = code.encode_batch(msg)
        rx = cw.copy()
        for f in range(len(rx)):
            for p in rng.choice(code.n, int(rng.integers(0, code.t + 2)), replace=False):
                rx[f, p] ^= 1 if hi == 2 else int(rng.integers(1, 256))
        full = code.correct_batch(rx)
        for B in sizes:
            assert np.array_equal(code.encode_batch(msg[:B]), cw[:B]), (code.to_string(), B)
            part = code.correct_batch(rx[:B])
            for key in ("out", "nerr", "status"):
                assert np.array_equal(part[key], full[key][:B]), (code.to_string(), B, key)
