"""GPU tests of the batched encoder / message extraction (cc_encode_batch, cc_extract_batch) and
full-size round-trip properties of the whole chain (encode -> corrupt -> decode)."""
import numpy as np
import pytest

import golden_util as G
from checkers import BCH, REF_CODES, RS, Oracle

import channelcoding_amd as cc
from channelcoding_amd import capi

pytestmark = pytest.mark.gpu


def make_code(cid, tag=None, **kw):
    fam, q, t = REF_CODES[cid]
    cls = cc.primitive_bch if fam == BCH else cc.rs
    return cls(q, cc.errors(t), tag or cc.berlekamp_massey_tag(), **kw)


@pytest.mark.parametrize("cid", G.HARD_CIDS)
def test_encode_golden(cid):
    d = G.load("encode", cid)
    code = make_code(cid)
    cw = code.encode_batch(d["msg"])
    assert np.array_equal(cw, d["cw"])
    assert np.array_equal(code.extract_batch(cw), d["msg"])
    assert np.array_equal(code.encode(d["msg"][3]), d["cw"][3])  # single-frame API


@pytest.mark.parametrize("cid", [0, 5, 6, 9, 10])
def test_encode_multiplication_tag(cid):
    """cyclic.h:29-33: c = a * g."""
    fam, q, t = REF_CODES[cid]
    o = Oracle(fam, q, t, coding=1)
    code = make_code(cid, coding="multiplication")
    rng = np.random.default_rng(70 + cid)
    hi = 2 if fam == BCH else 1 << q
    msg = rng.integers(0, hi, (300, o.l)).astype(np.uint8)
    msg[0] = 0
    assert np.array_equal(code.encode_batch(msg), o.encode(msg))


def test_encode_errors():
    code = make_code(9)
    with pytest.raises(cc.CcError) as e:
        code.encode_batch(np.zeros((2, 8), np.uint8))  # "Source code word has wrong length", cyclic.h:291-296
    assert e.value.status == capi.ERR_LENGTH
    with pytest.raises(cc.CcError) as e:
        code.encode_batch(np.full((1, 9), 200, np.uint8))
    assert e.value.status == capi.ERR_NOT_IN_FIELD
    assert code.encode_batch(np.zeros((0, 9), np.uint8)).shape == (0, 15)


@pytest.mark.parametrize("cid,log2b", [(6, 20), (10, 18), (10, 20)])  # (10, 20) = BASELINE configs[3] at full size
def test_full_size_roundtrip(cid, log2b):
    """BASELINE sizes through size-independent properties, all on device:
    every codeword has zero syndromes (decoder returns it untouched with nerr = 0);
    encode -> <= t random symbol errors -> decode returns the transmitted word and the error count;
    extract(encode(m)) == m; encoding is linear."""
    import torch
    fam, q, t = REF_CODES[cid]
    code = make_code(cid)
    B = 1 << log2b
    g = torch.Generator(device="cuda")
    g.manual_seed(cid)
    hi = 2 if fam == BCH else 1 << q
    msg = torch.randint(0, hi, (B, code.l), dtype=torch.uint8, device="cuda", generator=g)
    cw = code.encode_batch(msg)
    assert torch.equal(code.extract_batch(cw), msg)
    clean = code.correct_batch(cw)
    assert int((clean["status"] != 0).sum()) == 0 and int(clean["nerr"].abs().sum()) == 0
    assert torch.equal(clean["out"], cw)
    # linearity: enc(a) ^ enc(b) == enc(a ^ b)
    msg2 = torch.randint(0, hi, (B, code.l), dtype=torch.uint8, device="cuda", generator=g)
    assert torch.equal(cw ^ code.encode_batch(msg2), code.encode_batch(msg ^ msg2))
    # <= t errors at distinct random positions, non-zero values
    nerr = torch.randint(0, t + 1, (B,), device="cuda", generator=g)
    perm = torch.rand((B, code.n), device="cuda", generator=g).argsort(dim=1)[:, :t]
    vals = torch.randint(1, hi, (B, t), dtype=torch.uint8, device="cuda", generator=g)
    vals = torch.where(torch.arange(t, device="cuda")[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
    rx = cw.clone()
    rx.scatter_(1, perm, rx.gather(1, perm) ^ vals)
    res = code.correct_batch(rx)
    assert int((res["status"] != 0).sum()) == 0
    assert torch.equal(res["out"], cw)
    assert torch.equal(res["nerr"], nerr.to(torch.int32))


def test_rs255_223_full_size_against_the_oracle():
    """BASELINE configs[3] at its full size: RS(255,223), 2^20 frames, 0..18 random symbol errors per frame (two
    beyond the capability), Berlekamp-Massey: the seven-kernel bit-plane path on the whole batch; 2048 frames sampled
    at a stride are compared with the oracle frame for frame (corrected word, error count, failure class), every frame
    within the capability must come back as the transmitted word."""
    import torch
    from checkers import BM
    from test_gpu_algebraic import check_against_oracle
    code = make_code(10)
    o = Oracle(RS, 8, 16)
    B, t = 1 << 20, 16
    g = torch.Generator(device="cuda")
    g.manual_seed(1020)
    msg = torch.randint(0, 256, (B, code.l), dtype=torch.uint8, device="cuda", generator=g)
    cw = code.encode_batch(msg)
    nerr = torch.randint(0, t + 3, (B,), device="cuda", generator=g)
    perm = torch.rand((B, code.n), device="cuda", generator=g).argsort(dim=1)[:, :t + 2]
    vals = torch.randint(1, 256, (B, t + 2), dtype=torch.uint8, device="cuda", generator=g)
    vals = torch.where(torch.arange(t + 2, device="cuda")[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
    del msg
    rx = cw.clone()
    rx.scatter_(1, perm, rx.gather(1, perm) ^ vals)
    del perm, vals
    res = code.correct_batch(rx)
    easy = nerr <= t
    assert int((res["status"][easy] != 0).sum()) == 0
    assert torch.equal(res["out"][easy], cw[easy]) and torch.equal(res["nerr"][easy], nerr[easy].to(torch.int32))
    hard = ~easy  # beyond the capability: a failure, or a mis-correction to another codeword (then H c = 0 again)
    bad = res["status"][hard] != 0
    assert torch.equal(res["out"][hard][bad], rx[hard][bad])
    idx = torch.arange(0, B, B // 2048, device="cuda")
    sample = {k: v[idx].cpu().numpy() for k, v in res.items()}
    check_against_oracle(sample, o, BM, rx[idx].cpu().numpy())


@pytest.mark.parametrize("cid", G.MULT_CIDS)
def test_multiplication_tag_golden(cid):
    """multiplication_tag against the reference's committed outputs: encode c = a g, message extraction
    a = b / g (extract_multiplication_kernel), also for words that are not codewords."""
    msg, cw, rx, quot = G.mult_case(cid)
    code = make_code(cid, coding="multiplication")
    assert np.array_equal(code.encode_batch(msg), cw)
    assert np.array_equal(code.extract_batch(rx), quot)
    fam, q, t = REF_CODES[cid]
    o = Oracle(fam, q, t, coding=1)
    rng = np.random.default_rng(cid)
    hi = 2 if fam == BCH else 1 << q
    big = rng.integers(0, hi, (5000, o.n)).astype(np.uint8)
    assert np.array_equal(code.extract_batch(big), o.extract(big))
    # decode = correct + extract with t errors on a * g
    bad = cw.copy()
    for f in range(len(bad)):
        for p in rng.choice(o.n, t, replace=False):
            bad[f, p] ^= 1 if fam == BCH else int(rng.integers(1, hi))
    res = code.decode_batch(bad)
    assert (res["status"] == 0).all() and np.array_equal(res["msg"], msg)


def test_decode_entry_points():
    """cc_decode_hard_batch / cc_decode_soft_batch = correct + extract (cyclic::decode, cyclic.h:313-327)."""
    rng = np.random.default_rng(17)
    for cid, alg in ((10, cc.berlekamp_massey_tag()), (6, cc.euklid_tag()), (9, cc.peterson_gorenstein_zierler_tag())):
        fam, q, t = REF_CODES[cid]
        o = Oracle(fam, q, t)
        code = make_code(cid, alg)
        hi = 2 if fam == BCH else 1 << q
        msg = rng.integers(0, hi, (500, o.l)).astype(np.uint8)
        rx = o.encode(msg)
        injected = rng.integers(0, t + 3, len(rx))
        for f in range(len(rx)):
            for p in rng.choice(o.n, int(injected[f]), replace=False):
                rx[f, p] ^= 1 if fam == BCH else int(rng.integers(1, hi))
        a = code.decode_batch(rx)
        b = code.correct_batch(rx)
        assert np.array_equal(a["out"], b["out"]) and np.array_equal(a["status"], b["status"])
        assert np.array_equal(a["nerr"], b["nerr"]) and np.array_equal(a["msg"], o.extract(b["out"]))
        ok = a["status"] == 0
        easy = injected <= t  # beyond t a decoder may land on another codeword
        assert ok[easy].all() and (~ok).any() and np.array_equal(a["msg"][easy], msg[easy])
    soft = cc.primitive_bch(6, cc.errors(3), cc.normalized_min_sum_tag(10, 0.8))
    o = Oracle(BCH, 6, 3)
    msg = rng.integers(0, 2, (400, o.l)).astype(np.uint8)
    y = (1.0 - 2.0 * o.encode(msg) + soft.sigma(5.0) * rng.standard_normal((400, 63))).astype(np.float32)
    a, b = soft.decode_batch(y), soft.correct_batch(y)
    assert np.array_equal(a["out"], b["out"]) and np.array_equal(a["iters"], b["iters"])
    assert np.array_equal(a["msg"], b["out"][:, soft.k:])
    hardf = cc.primitive_bch(6, cc.errors(3), cc.berlekamp_massey_tag()).decode_batch(y)
    assert np.array_equal(hardf["msg"], hardf["out"][:, soft.k:]) and (hardf["status"] == 0).any()


@pytest.mark.parametrize("q,t", [(4, 2), (5, 3), (6, 3), (7, 2), (8, 3), (8, 4), (8, 5)])
def test_bch_encode_accepts_any_field_element(q, t):
    """The reference's BCH codes carry GF(2^q) elements as symbols (cyclic.h:300-301): a message symbol may be any
    field element, and the parity symbols are then XORs of such elements.  The bit-plane encoder (k <= 32) and the
    table encoder (k > 32) must both agree with the oracle."""
    o = Oracle(BCH, q, t)
    code = cc.primitive_bch(q, cc.errors(t), cc.berlekamp_massey_tag())
    rng = np.random.default_rng(40 * q + t)
    msg = rng.integers(0, o.n + 1, (777, o.l)).astype(np.uint8)
    msg[0] = 0
    msg[1] = o.n
    cw = code.encode_batch(msg)
    assert np.array_equal(cw, o.encode(msg))
    assert np.array_equal(code.extract_batch(cw), msg)
    bits = (msg & 1).astype(np.uint8)
    assert np.array_equal(code.encode_batch(bits), o.encode(bits))
