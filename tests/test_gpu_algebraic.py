"""GPU parity tests of the algebraic chain (syndromes -> Berlekamp-Massey -> root search -> error
values -> re-check), through the C ABI (cc_correct_hard_batch / _f32 / _dev), against the golden
vectors of the real reference and against the plain-C oracle on fresh seeded inputs.  Bit-exact.

Reference defects fenced exactly as in tests/test_oracle_golden.py: F3 (BM out-of-bounds read ->
reference result undefined on frames the oracle flags), Q9 (reference PGZ / rs::error_values Gauss
elimination).  The device implements the correct algorithms, so on fenced frames it is compared with
the oracle instead of the reference.
"""
import numpy as np
import pytest

import golden_util as G
from checkers import BCH, BM, EUKLID, PGZ, REF_CODES, RS, Oracle, awgn_llr

import channelcoding_amd as cc
from channelcoding_amd import capi

pytestmark = pytest.mark.gpu

TAGS = {PGZ: cc.peterson_gorenstein_zierler_tag, BM: cc.berlekamp_massey_tag, EUKLID: cc.euklid_tag}
KEYS = {PGZ: "pgz", BM: "bm", EUKLID: "euklid"}


def make_code(cid, alg):
    fam, q, t = REF_CODES[cid]
    cls = cc.primitive_bch if fam == BCH else cc.rs
    return cls(q, cc.errors(t), TAGS[alg]())


def corrupt(rng, o, cw, nerr):
    b = cw.copy()
    for p in rng.choice(o.n, nerr, replace=False):
        b[p] ^= 1 if o.family == BCH else int(rng.integers(1, 1 << o.q))
    return b


def check_against_oracle(res, o, alg, rx, erasures=None):
    out, nerr, st, ub = o.correct_hard(alg, rx, erasures or ())
    assert np.array_equal(res["status"] == 0, st == 0)
    ok = st == 0
    assert np.array_equal(res["out"][ok], out[ok])
    assert np.array_equal(res["nerr"][ok], nerr[ok])
    assert (res["nerr"][~ok] == -1).all()
    if alg == BM:  # failure class (root count vs re-check) is only defined algorithm-for-algorithm for BM
        assert np.array_equal(res["status"], st)
    # failed frames return the (hard-decided) input unchanged
    sym = (rx < 0).astype(np.uint8) if rx.dtype == np.float32 else rx
    assert np.array_equal(res["out"][~ok], sym.reshape(-1, o.n)[~ok])


@pytest.mark.parametrize("cid", G.HARD_CIDS)
def test_hard_golden(cid):
    d = G.load("hard", cid)
    o = Oracle(*REF_CODES[cid])
    for alg in (PGZ, BM, EUKLID):
        code = make_code(cid, alg)
        res = code.correct_batch(d["rx"])
        key = KEYS[alg]
        r_st, r_out = d["status_" + key], d["out_" + key]
        _, _, _, ub = o.correct_hard(alg, d["rx"])
        skip = d["notsolvable_" + key].copy()
        if alg == BM:
            skip |= ub.astype(bool)
        if alg == PGZ:
            skip |= (r_st == 0) != (d["status_euklid"] == 0)
        keep = ~skip
        assert keep.sum() >= 0.8 * len(keep)
        assert np.array_equal((res["status"] == 0)[keep], (r_st == 0)[keep]), key
        ok = keep & (r_st == 0)
        assert np.array_equal(res["out"][ok], r_out[ok]), key
        easy = d["nerr"] <= o.t
        assert (res["status"][easy] == 0).all() and np.array_equal(res["out"][easy], d["cw"][easy])
        assert np.array_equal(res["nerr"][easy], d["nerr"][easy])
        check_against_oracle(res, o, alg, d["rx"])  # every frame, including the fenced ones


def test_exercises_kat():
    """src/exercises.c++ tasks 6.1-6.10 through the single-frame API (exceptions as in the reference)."""
    for case in G.exercises():
        if case["erasures"] and case["alg"] == PGZ:
            continue
        fam, q, t = REF_CODES[case["code"]]
        code = make_code(case["code"], case["alg"])
        rx = np.array(case["rx"], np.uint8)
        if case["status"] == 0:
            assert list(code.correct(rx, case["erasures"])) == case["out"], case["task"]
        else:
            with pytest.raises(cc.decoding_failure):
                code.correct(rx, case["erasures"])
    a = np.array([1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1], np.uint8)
    code = cc.primitive_bch(4, cc.dmin(7))  # task 6.1, default algorithm PGZ
    assert np.array_equal(code.correct([1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 1, 1]), a)
    assert np.array_equal(code.decode([1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1]), a[code.k:])


@pytest.mark.parametrize("cid,frames", [(0, 999), (1, 513), (4, 1000), (12, 300), (5, 1001), (13, 500), (11, 500),
                                        (6, 3000), (7, 300), (8, 1000), (9, 1000), (10, 1500)])
def test_hard_vs_oracle_seeded(cid, frames):
    o = Oracle(*REF_CODES[cid])
    rng = np.random.default_rng(6000 + cid)
    hi = 2 if o.family == BCH else 1 << o.q
    cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
    rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, o.t + 4))) for f in range(frames)])
    rx[0] = cw[0]  # a clean frame
    for alg in (PGZ, BM, EUKLID):
        res = make_code(cid, alg).correct_batch(rx)
        check_against_oracle(res, o, alg, rx)


@pytest.mark.parametrize("cid", [5, 6])
def test_hard_from_soft_values(cid):
    """signed input sequence: bit = (x < 0) (cyclic.h:163-173); 0.0 and -0.0 decide for bit 0 (codes.h:51)."""
    o = Oracle(*REF_CODES[cid])
    rng = np.random.default_rng(61)
    cw = o.encode(rng.integers(0, 2, (400, o.l)).astype(np.uint8))
    y = awgn_llr(rng, cw, o.l / o.n, 6.0)
    y[0, :4] = [0.0, -0.0, -1e-30, 1e-30]
    for alg in (BM, EUKLID):
        res = make_code(cid, alg).correct_batch(y)
        check_against_oracle(res, o, alg, y)


@pytest.mark.parametrize("cid", [8, 9, 10, 5])
def test_erasures_bm_euklid(cid):
    """hard_decision.h:128-131,:171-172: erasure locators pre-loaded into lambda; per-frame CSR lists."""
    o = Oracle(*REF_CODES[cid])
    rng = np.random.default_rng(6200 + cid)
    hi = 2 if o.family == BCH else 1 << o.q
    frames = 200
    cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
    rx = cw.copy()
    per = []
    for f in range(frames):
        ne = int(rng.integers(0, 2 * o.t + 1))
        er = sorted(rng.choice(o.n, ne, replace=False).tolist())
        for e in er:
            rx[f, e] = 0
        nerr = int(rng.integers(0, max(1, (2 * o.t - ne) // 2 + 2)))
        free = [p for p in range(o.n) if p not in er]
        for p in rng.choice(free, nerr, replace=False):
            rx[f, p] ^= 1 if o.family == BCH else int(rng.integers(1, hi))
        per.append(er)
    for alg in (BM, EUKLID):
        res = make_code(cid, alg).correct_batch(rx, erasures=per)
        for f in range(frames):
            out, nerr, st, ub = o.correct_hard(alg, rx[f], per[f])
            assert (res["status"][f] == 0) == (st[0] == 0), (alg, f, per[f])
            if alg == BM:
                assert res["status"][f] == st[0]
            if st[0] == 0:
                assert np.array_equal(res["out"][f], out[0]) and res["nerr"][f] == nerr[0]


@pytest.mark.parametrize("alg", [BM, EUKLID, PGZ])
def test_erasures_on_a_signed_input_sequence(alg):
    """cyclic.h:207-252 takes erasures for ANY InputSequence: a signed one is hard-decided first (bit = x < 0,
    :163-173), then decoded like the symbols (VERDICT r1 Missing #2).  Host and device entry points, all three
    algorithms (PGZ: the two-trial rule of bch.h:97-149), against the oracle on the same soft values."""
    import torch
    o = Oracle(BCH, 6, 3)
    rng = np.random.default_rng(4100 + alg)
    frames = 240
    cw = o.encode(rng.integers(0, 2, (frames, o.l)).astype(np.uint8))
    rx = cw.copy()
    per = []
    for f in range(frames):
        ne = int(rng.integers(0, 2 * o.t + 2))  # up to 7 > 2t
        er = sorted(rng.choice(o.n, ne, replace=False).tolist())
        for e in er:
            rx[f, e] = int(rng.integers(0, 2))
        free = [p_ for p_ in range(o.n) if p_ not in er]
        for p_ in rng.choice(free, int(rng.integers(0, 3)), replace=False):
            rx[f, p_] ^= 1
        per.append(er)
    y = ((1.0 - 2.0 * rx) * rng.uniform(0.1, 2.0, rx.shape)).astype(np.float32)
    y[0, :2] = [0.0, -0.0]  # both decide for bit 0 (codes.h:51)
    bits = (y < 0).astype(np.uint8)
    code = make_code(5, alg)
    host = code.correct_batch(y, erasures=per)
    dev = code.correct_batch(torch.from_numpy(y).cuda(), erasures=per)
    sym = code.correct_batch(bits, erasures=per)
    for key in ("out", "status", "nerr"):
        assert np.array_equal(host[key], dev[key].cpu().numpy()), key
        assert np.array_equal(host[key], sym[key]), key  # same answer as on the hard-decided symbols
    for f in range(frames):
        out, nerr, st, ub = o.correct_hard(alg, y[f], per[f])
        if alg == PGZ and len(per[f]):
            continue  # the oracle's PGZ refuses erasures (hard_decision.h:66-68); the two-trial rule is checked on symbols
        assert (host["status"][f] == 0) == (st[0] == 0), (alg, f, per[f])
        if st[0] == 0:
            assert np.array_equal(host["out"][f], out[0]) and host["nerr"][f] == nerr[0]


def test_host_entry_points_in_chunks(monkeypatch):
    """The host-pointer entry points cut a batch into chunks on two private streams with handle-owned buffers
    (no allocation or device-wide synchronisation per call): many small chunks, ragged erasure lists, repeated
    calls of growing and shrinking size -- all equal to the device-pointer path."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import channelcoding_amd as cc
rng = np.random.default_rng(5)
soft = cc.primitive_bch(6, cc.errors(3), cc.min_sum_tag(10))
hard = cc.primitive_bch(6, cc.errors(3), cc.berlekamp_massey_tag())
for B in (3000, 17, 5000, 1):
    y = (1.0 + 0.6 * rng.standard_normal((B, 63))).astype(np.float32)
    per = [sorted(rng.choice(63, int(rng.integers(0, 4)), replace=False).tolist()) for _ in range(B)]
    a = soft.correct_batch(y, erasures=per, want_L=True)
    b = soft.correct_batch(torch.from_numpy(y).cuda(), erasures=per, want_L=True)
    for k in ("out", "status", "iters", "L"):
        assert np.array_equal(a[k], b[k].cpu().numpy()), (B, k)
    sym = (y < 0).astype(np.uint8)
    a = hard.correct_batch(sym, erasures=per)
    b = hard.correct_batch(torch.from_numpy(sym).cuda(), erasures=per)
    for k in ("out", "status", "nerr"):
        assert np.array_equal(a[k], b[k].cpu().numpy()), (B, k)
    msg = rng.integers(0, 2, (B, 45)).astype(np.uint8)
    cw = hard.encode_batch(msg)
    assert np.array_equal(cw, hard.encode_batch(torch.from_numpy(msg).cuda()).cpu().numpy())
    assert np.array_equal(hard.extract_batch(cw), msg)
print("ok")
""" % (root, root)
    env = dict(os.environ, CC_AMD_HOST_CHUNK_BYTES="40000")  # ~160 soft frames, ~630 symbol frames per chunk
    run = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


def test_pgz_erasure_two_trial_rule_bch():
    """bch.h:97-149: PGZ + erasures decodes twice (erasures := 0, := 1) and keeps the result with fewer errors."""
    o = Oracle(BCH, 6, 3)
    rng = np.random.default_rng(77)
    frames = 300
    cw = o.encode(rng.integers(0, 2, (frames, o.l)).astype(np.uint8))
    rx = cw.copy()
    per = []
    for f in range(frames):
        er = sorted(rng.choice(o.n, int(rng.integers(0, 8)), replace=False).tolist())  # up to 7 > 2t = 6
        for p in rng.choice(o.n, int(rng.integers(0, 4)), replace=False):
            rx[f, p] ^= 1
        per.append(er)
    res = make_code(5, PGZ).correct_batch(rx, erasures=per)
    for f in range(frames):
        out, nerr, st, ub = o.correct_hard(PGZ, rx[f], per[f])
        assert res["status"][f] == st[0], (f, per[f], res["status"][f], st[0])
        if st[0] == 0:
            assert np.array_equal(res["out"][f], out[0]) and res["nerr"][f] == nerr[0]
        else:
            assert np.array_equal(res["out"][f], rx[f]) and res["nerr"][f] == -1


def test_api_errors_and_device_pointers():
    import torch
    code = cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag())
    with pytest.raises(cc.CcError) as e:
        cc.rs(4, cc.errors(3), cc.berlekamp_massey_tag()).correct_batch(np.full((1, 15), 16, np.uint8))
    assert e.value.status == capi.ERR_NOT_IN_FIELD  # galois.h:149-152
    with pytest.raises(cc.CcError) as e:
        code.correct_batch(np.zeros((2, 254), np.uint8))
    assert e.value.status == capi.ERR_LENGTH
    with pytest.raises(cc.CcError) as e:
        cc.rs(4, cc.errors(3)).correct_batch(np.zeros((1, 15), np.uint8), erasures=[1])  # PGZ + erasures
    assert e.value.status == capi.ERR_UNSUPPORTED
    assert code.correct_batch(np.zeros((0, 255), np.uint8))["out"].shape == (0, 255)
    # device pointers on a side stream
    o = Oracle(RS, 8, 16)
    rng = np.random.default_rng(63)
    cw = o.encode(rng.integers(0, 256, (777, o.l)).astype(np.uint8))
    rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, 17))) for f in range(len(cw))])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        res = code.correct_batch(torch.from_numpy(rx).cuda())
    s.synchronize()
    assert (res["status"] == 0).all() and np.array_equal(res["out"].cpu().numpy(), cw)


@pytest.mark.parametrize("cid,frames", [(0, 30000), (1, 20000), (4, 20000), (7, 20000), (8, 30000), (9, 20000),
                                        (10, 600)])
def test_recheck_decided_by_bm_length(cid, frames):
    """The kernel skips the syndrome re-check (cyclic.h:243-248) when BM's LFSR length equals deg lambda and the
    root count matched (proof in algebraic.hip); uniformly random words and heavy error patterns hit every
    other case.  Status classes (root count vs re-check) must match the oracle frame for frame."""
    o = Oracle(*REF_CODES[cid])
    rng = np.random.default_rng(9000 + cid)
    hi = 2 if o.family == BCH else 1 << o.q
    cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
    rx = cw.copy()
    third = frames // 3
    rx[:third] = rng.integers(0, hi, (third, o.n))  # arbitrary words
    for f in range(third, frames):
        rx[f] = corrupt(rng, o, cw[f], int(rng.integers(o.t + 1, min(o.n, 3 * o.t + 3))))
    for alg in (BM, PGZ):
        res = make_code(cid, alg).correct_batch(rx)
        check_against_oracle(res, o, alg, rx)
    st = make_code(cid, BM).correct_batch(rx)["status"]
    assert (st == 2).any() and ((st == 0).any() or cid == 10)
    if cid in (8, 9):  # re-check failures exist for the small RS codes (never for a binary word, see the proof)
        assert (st == 3).any()


def test_both_algebraic_kernels_agree(monkeypatch):
    """The chunked kernel (Berlekamp-Massey with one lane per frame, algebraic_chunk.hip) and the
    one-wavefront-per-frame kernel must agree bit for bit; CC_AMD_NO_CHUNK is read once per process, so the
    second kernel is reached through its own dispatch conditions instead: erasure arrays (empty) force it."""
    for fam, q, t in ((RS, 8, 16), (RS, 6, 5), (BCH, 7, 4)):
        o = Oracle(fam, q, t)
        rng = np.random.default_rng(123 + q)
        frames = 700
        hi = 2 if fam == BCH else 1 << q
        cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
        rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, o.t + 4))) for f in range(frames)])
        code = (cc.primitive_bch if fam == BCH else cc.rs)(q, cc.errors(t), cc.berlekamp_massey_tag())
        assert code.kernel_info()["kernel"].startswith("algebraic_chunk_kernel")
        a = code.correct_batch(rx)
        b = code.correct_batch(rx, erasures=[[] for _ in range(frames)])  # CSR with no entries -> algebraic_kernel
        for key in ("out", "status", "nerr"):
            assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("fam,q,t,frames", [(BCH, 3, 1, 500), (BCH, 4, 1, 500), (BCH, 5, 7, 400), (BCH, 8, 1, 300),
                                            (RS, 8, 1, 300), (RS, 8, 32, 150), (RS, 8, 31, 150), (RS, 5, 12, 300), (BCH, 8, 30, 100)])
def test_extreme_code_parameters(fam, q, t, frames):
    """Smallest and largest supported geometries: n = 7, t = 1, 64 syndromes (the per-lane / LDS limits of both
    algebraic kernels), very low rate."""
    o = Oracle(fam, q, t)
    rng = np.random.default_rng(31 * q + t)
    hi = 2 if fam == BCH else 1 << q
    cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
    rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, min(o.n, o.t + 3)))) for f in range(frames)])
    cls = cc.primitive_bch if fam == BCH else cc.rs
    for alg in (PGZ, BM, EUKLID):
        code = cls(q, cc.errors(t), TAGS[alg]())
        assert np.array_equal(code.encode_batch(o.extract(cw)), cw)
        # (Euklid at t = 32 needs coefficient 64 of x^2t: algebraic_long.hip, four coefficients per lane)
        check_against_oracle(code.correct_batch(rx), o, alg, rx)


def check_erasure_frames(code, o, alg, rxe, ers):
    """per-frame erasure lists against the oracle (which takes one list per call)"""
    res = code.correct_batch(rxe, erasures=ers)
    for f in range(len(rxe)):
        out, nerr, st, ub = o.correct_hard(alg, rxe[f], ers[f])
        assert (res["status"][f] == 0) == (st[0] == 0), (alg, f, ers[f], res["status"][f], st[0])
        if alg == BM:
            assert res["status"][f] == st[0], (f, res["status"][f], st[0])
        if st[0] == 0:
            assert np.array_equal(res["out"][f], out[0]) and res["nerr"][f] == nerr[0], (alg, f)
        else:
            assert np.array_equal(res["out"][f], rxe[f])


@pytest.mark.parametrize("fam,t,frames", [(RS, 33, 120), (RS, 40, 100), (RS, 64, 80), (RS, 100, 40), (RS, 120, 30),
                                          (BCH, 40, 100), (BCH, 43, 60), (BCH, 63, 60)])
def test_more_than_64_syndromes(fam, t, frames):
    """errors<t> with t > 32 (bch.h:28-46, rs.h:18-28 instantiate any t): algebraic_long.hip, every tag, 0 .. t + 3
    errors per frame, against the oracle frame for frame; then the same codes with erasures (BM and Euklid: erasure
    pre-load hard_decision.h:128-131, :171-172), and Euklid with erasures at t = 20 and 32 (2t > 32: the Sugiyama
    kernel's lane budget ends there)."""
    o = Oracle(fam, 8, t)
    rng = np.random.default_rng(700 + t)
    hi = 2 if fam == BCH else 256
    cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
    rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, min(o.n, o.t + 4)))) for f in range(frames)])
    cls = cc.primitive_bch if fam == BCH else cc.rs
    for alg in (PGZ, BM, EUKLID):
        code = cls(8, cc.errors(t), TAGS[alg]())
        assert np.array_equal(code.encode_batch(o.extract(cw)), cw)
        res = code.correct_batch(rx)
        check_against_oracle(res, o, alg, rx)
        assert (res["status"] == 0).sum() >= frames // 2
    # erasures: e erased positions (zeroed) + up to (2t - e) / 2 errors elsewhere, a few frames beyond that
    for alg in (BM, EUKLID):
        code = cls(8, cc.errors(t), TAGS[alg]())
        rxe, ers = cw[:40].copy(), []
        for f in range(len(rxe)):
            ne = int(rng.integers(0, min(2 * o.t + 1, o.n // 2)))  # (more than 2t erasures: status 4 on the device)
            er = sorted(rng.choice(o.n, ne, replace=False).tolist())
            rxe[f, er] = 0
            free = np.setdiff1d(np.arange(o.n), er)
            for pos in rng.choice(free, int(rng.integers(0, max(1, (2 * o.t - ne) // 2 + 2))), replace=False):
                rxe[f, pos] ^= 1 if fam == BCH else int(rng.integers(1, hi))
            ers.append(er)
        check_erasure_frames(code, o, alg, rxe, ers)


@pytest.mark.parametrize("fam,t", [(RS, 20), (RS, 32), (BCH, 21)])
def test_euklid_with_erasures_beyond_32_syndromes(fam, t):
    o = Oracle(fam, 8, t)
    rng = np.random.default_rng(800 + t)
    hi = 2 if fam == BCH else 256
    cw = o.encode(rng.integers(0, hi, (60, o.l)).astype(np.uint8))
    code = (cc.primitive_bch if fam == BCH else cc.rs)(8, cc.errors(t), TAGS[EUKLID]())
    rxe, ers = cw.copy(), []
    for f in range(len(rxe)):
        ne = int(rng.integers(0, 2 * o.t + 1))
        er = sorted(rng.choice(o.n, ne, replace=False).tolist())
        rxe[f, er] = 0
        free = np.setdiff1d(np.arange(o.n), er)
        for pos in rng.choice(free, int(rng.integers(0, max(1, (2 * o.t - ne) // 2 + 2))), replace=False):
            rxe[f, pos] ^= 1 if fam == BCH else int(rng.integers(1, hi))
        ers.append(er)
    check_erasure_frames(code, o, EUKLID, rxe, ers)


@pytest.mark.parametrize("fam,q,t,frames", [(RS, 4, 4, 30000), (RS, 5, 4, 20000), (RS, 5, 8, 8000), (BCH, 6, 4, 20000)])
def test_recheck_paths_of_the_chunked_kernel(fam, q, t, frames):
    """Codes with at least 8 syndromes run algebraic_chunk_kernel; random words and heavy error patterns make its
    rare branches fire (L != deg lambda -> explicit re-check, re-check failures, PGZ degree bound)."""
    o = Oracle(fam, q, t)
    rng = np.random.default_rng(777 + 31 * q + t)
    hi = 2 if fam == BCH else 1 << q
    cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
    rx = cw.copy()
    half = frames // 2
    rx[:half] = rng.integers(0, hi, (half, o.n))
    for f in range(half, frames):
        rx[f] = corrupt(rng, o, cw[f], int(rng.integers(0, min(o.n, 3 * o.t))))
    cls = cc.primitive_bch if fam == BCH else cc.rs
    for alg in (BM, PGZ):
        code = cls(q, cc.errors(t), TAGS[alg]())
        assert code.kernel_info()["kernel"].startswith("algebraic_chunk_kernel")
        check_against_oracle(code.correct_batch(rx), o, alg, rx)
    st = cls(q, cc.errors(t), TAGS[BM]()).correct_batch(rx)["status"]
    assert (st == 0).any() and (st == 2).any()
    if fam == RS and t == 4:
        assert (st == 3).any()


@pytest.mark.parametrize("fam,q,t,frames", [(RS, 8, 16, 20000), (BCH, 8, 3, 70000), (RS, 8, 16, 300), (BCH, 6, 2, 5000),
                                            (RS, 6, 8, 5000), (RS, 8, 40, 600)])
def test_in_place_calls(fam, q, t, frames):
    """cc_correct_hard_batch_dev with d_out == d_in (the bit-plane chain then skips its copy of the words): the same
    words, counts and flags as the out-of-place call, on every kernel family (planes, one wavefront per frame, chunks,
    long locators)."""
    import ctypes as C

    import torch
    from channelcoding_amd import capi
    o = Oracle(fam, q, t)
    code = (cc.primitive_bch if fam == BCH else cc.rs)(q, cc.errors(t), TAGS[BM]())
    rng = np.random.default_rng(q * 1000 + t)
    hi = 2 if fam == BCH else (1 << q)
    base = o.encode(rng.integers(0, hi, (200, o.l)).astype(np.uint8))
    rx = np.stack([corrupt(rng, o, base[f], int(rng.integers(0, o.t + 3))) for f in range(200)])
    rx = np.tile(rx, ((frames + 199) // 200, 1))[:frames]
    lib = capi.lib()
    vp = lambda x: C.c_void_p(x.data_ptr())
    d_in = torch.from_numpy(rx).cuda()
    d_out = torch.empty_like(d_in)
    ne = [torch.empty(frames, dtype=torch.int32, device="cuda") for _ in range(2)]
    st = [torch.empty(frames, dtype=torch.int32, device="cuda") for _ in range(2)]
    sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.cc_correct_hard_batch_dev(code._h, vp(d_in), None, None, vp(d_out), vp(ne[0]), vp(st[0]), frames, sh) == 0
    assert lib.cc_correct_hard_batch_dev(code._h, vp(d_in), None, None, vp(d_in), vp(ne[1]), vp(st[1]), frames, sh) == 0
    torch.cuda.synchronize()
    assert torch.equal(d_in, d_out) and torch.equal(ne[0], ne[1]) and torch.equal(st[0], st[1])
    head = dict(out=d_out[:200].cpu().numpy(), nerr=ne[0][:200].cpu().numpy(), status=st[0][:200].cpu().numpy())
    check_against_oracle(head, o, BM, rx[:200])
