#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REAL reference
(oracle/_ref/libccref_o{0,1}.so, built from /root/reference by oracle/Makefile).

Run in the development container only:  python tests/golden/make_golden.py
The fixtures are data (inputs + the reference's outputs); no reference source
is stored.  LLR inputs are stored themselves, never RNG seeds of the reference
(libc++ and libstdc++ normal_distribution streams differ).

Files
  constants.json      n, k, l, t, dmin, g, h, roots, H row-0 support, to_string per code
  exercises.json      the known-answer vectors of src/exercises.c++ (tasks 6.1-6.10) with the reference's results
  encode_<code>.npz   msg -> codeword
  hard_<code>.npz     received words with 0..t+2 errors -> corrected word / status, for PGZ, BM, EUKLID
  mult.npz            multiplication_tag: c = a*g and a = b/g (also for words that are not codewords)
  minsum_alt.npz      H_alt<uint8_t>() of three codes and min_sum<float,uint8_t>(H_alt, y, tag) on it (O0/O1;
                      H_alt<gf2> is ill-formed in the reference, so there is no O2 leg)
  minsum_<code>.npz   LLR frames -> b, L, iteration, status for every variant x stop rule O0/O1/O2
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from checkers import (BCH, BM, EUKLID, PGZ, REF_CODES, REF_VARIANTS, RS, Oracle, RefLib, awgn_llr)  # noqa: E402

NAMES = {0: "bch15_7", 1: "bch15_5", 4: "bch31_16", 5: "bch63_45", 6: "bch255_231", 8: "rs7_3", 9: "rs15_9",
         10: "rs255_223", 11: "bch127_113", 13: "bch63_39"}


def constants(ref0):
    out = {}
    for cid in sorted(REF_CODES):
        i = ref0.info(cid)
        h = ref0.poly(cid, 1)
        support = [int(j) for j in range(len(h)) if h[len(h) - 1 - j]]
        out[str(cid)] = dict(
            family=i["family"], q=i["q"], cap_kind=i["cap_kind"], cap=i["cap"], n=i["n"], k=i["k"], l=i["l"],
            t=i["t"], dmin=i["dmin"], rate=i["rate"],
            g=[int(x) for x in ref0.poly(cid, 0)], h=[int(x) for x in h], roots=[int(x) for x in ref0.poly(cid, 2)],
            row0_support=support,
            to_string={a: ref0.to_string(cid, k) for k, a in ((PGZ, "PGZ"), (BM, "BM"), (EUKLID, "EUKLID"))})
    return out


def exercises(ref0):
    """Vectors of /root/reference/src/exercises.c++ (cited by task); alpha^p is stored as its field value."""
    fam, q, t = REF_CODES[8]
    f3 = Oracle(RS, 3, 2)  # only for exp table: alpha^p -> value
    f4 = Oracle(RS, 4, 3)

    def a3(p):
        return int(f3.exp[p])

    cases = []

    def add(task, cid, alg, rx, erasures=(), note=""):
        out, st, msg = ref0.correct(cid, alg, np.array(rx, np.uint8), erasures)
        cases.append(dict(task=task, code=cid, alg=alg, rx=[int(x) for x in rx], erasures=list(erasures),
                          status=int(st[0]), out=[int(x) for x in out[0]] if st[0] == 0 else None,
                          message=msg[0].split("\n")[0], note=note))

    # 6.1 exercises.c++:36-53  primitive_bch<4, dmin<7>>, PGZ; both must correct to a
    add("6.1 b1", 1, PGZ, [1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 1, 1], note="expect a=111000100110101")
    add("6.1 b2", 1, PGZ, [1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1], note="expect a=111000100110101")
    # 6.2 :55-78 dmin<5>: decoding failure expected
    add("6.2", 2, PGZ, [1, 0, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0], note="expect decoding_failure")
    # 6.3 :80-106 dmin<6>
    add("6.3 b1", 3, PGZ, [1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 1], note="expect a=111101110100011")
    add("6.3 b2", 3, PGZ, [1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 0, 0, 1, 0, 1], note="expect a=111101110100011")
    add("6.3 b3", 3, PGZ, [0, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 1, 1])
    # 6.4 :108-136 rs<3, errors<1>>
    add("6.4 b1", 7, PGZ, [0, 0, 0, 0, 0, 0, a3(4)])
    add("6.4 b2", 7, PGZ, [a3(2), a3(2), 1, 0, 0, 0, a3(4)])
    # 6.5 :138-160 rs<4, errors<3>>
    add("6.5", 9, PGZ, [1, 1, 1, 1] + [0] * 11, note="expect decoding_failure")
    # 6.6 :162-179 rs<3, errors<2>> PGZ
    add("6.6", 8, PGZ, [a3(6), a3(2), a3(2), a3(5), 0, 0, a3(5)])
    # 6.7 :182-206 BM with erasures {5,4,3,2}
    a = [a3(p) for p in (6, 2, 2, 5, 4, 6, 5)]
    b = list(a)
    for e in (5, 4, 3, 2):
        b[e] = 0
    add("6.7", 8, BM, b, erasures=(5, 4, 3, 2), note="expect_equal(a): " + str(a))
    # 6.8 :208-229
    add("6.8", 8, BM, [a3(p) for p in (2, 0, 4, 0, 5, 0, 2)], erasures=(1, 3))
    # 6.9 :231-256
    b9 = [a3(p) for p in (3, 4, 0, 3, 4, 3, 3)]
    add("6.9 bm", 8, BM, b9)
    add("6.9 pgz", 8, PGZ, b9)
    # 6.10 :258-274 primitive_bch<4, errors<2>> all three algorithms
    a10 = [1, 0, 1, 0, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1]
    for alg in (PGZ, BM, EUKLID):
        add("6.10", 0, alg, a10)
    return cases


def corrupt(rng, o, cw, nerr):
    b = cw.copy()
    for p in rng.choice(o.n, nerr, replace=False):
        b[p] ^= 1 if o.family == BCH else int(rng.integers(1, 1 << o.q))
    return b


def alt_golden(ref0, ref1):
    """cyclic::H_alt (cyclic.h:361-385) and min-sum over it.  Only all-zero codewords can be accepted under
    O1, so the frames are noisy all-zero words plus a few random codewords (those fail)."""
    d = {}
    for cid, iters, frames, ebno in ((0, 10, 48, 3.0), (1, 10, 48, 3.0), (5, 20, 32, 4.0), (6, 20, 8, 6.0)):
        fam, q, t = REF_CODES[cid]
        o = Oracle(fam, q, t)
        rng = np.random.default_rng(3000 + cid)
        c = np.zeros((frames, o.n), np.uint8)
        c[-4:] = ref0.encode(cid, rng.integers(0, 2, (4, o.l)).astype(np.uint8))
        y = awgn_llr(rng, c, o.l / o.n, ebno)
        pre = "c%d_" % cid
        d[pre + "H"] = np.packbits(ref0.H_alt(cid), axis=1)
        d[pre + "y"] = y
        d[pre + "iterations"] = np.array(iters)
        for v in sorted(REF_VARIANTS):
            for rule, lib in (("o0", ref0), ("o1", ref1)):
                b, L, it, st = lib.minsum_alt(cid, v, iters, 0, y)
                key = pre + "v%d_%s" % (v, rule)
                d[key + "_b"] = np.packbits(b, axis=1)
                d[key + "_L"] = L
                d[key + "_it"] = it.astype(np.uint16)
                d[key + "_st"] = st.astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "minsum_alt.npz"), **d)


def mult_golden(ref0):
    """multiplication_tag coding (cyclic.h:29-33, :42-46): c = a g, a = b / g."""
    d = {}
    for cid, frames in ((0, 64), (5, 48), (6, 32), (8, 64), (9, 48), (10, 24)):
        fam, q, t = REF_CODES[cid]
        o = Oracle(fam, q, t)
        rng = np.random.default_rng(4000 + cid)
        hi = 2 if fam == BCH else 1 << q
        msg = rng.integers(0, hi, (frames, o.l)).astype(np.uint8)
        msg[0] = 0
        msg[1] = hi - 1
        msg[2, -3:] = 0  # short quotient: decode pads with zeros
        cw = ref0.encode_mult(cid, msg)
        rx = cw.copy()  # arbitrary words: the quotient ignores the remainder
        for f in range(frames // 2, frames):
            rx[f] = rng.integers(0, hi, o.n)
        pre = "c%d_" % cid
        d[pre + "msg"], d[pre + "cw"], d[pre + "rx"] = msg, cw, rx
        d[pre + "quot"] = ref0.decode_mult(cid, rx)
    np.savez_compressed(os.path.join(HERE, "mult.npz"), **d)


def headline_golden(ref0, ref1):
    """SURVEY 8(c) F-MS for the headline code: BCH(255,231) MS<20> family, 256 frames at each of 2 / 4 / 6 dB (half
    all-zero words, half random codewords), every variant setting x O0 / O1 / O2 = 27 cases from the real reference
    (a few minutes on 8 cores: ctypes releases the GIL).  Hard decisions, iteration index and failure flag for every
    frame; L for the first LSUB frames of each Eb/N0 block (the full L would be 21 MB)."""
    from concurrent.futures import ThreadPoolExecutor
    cid, iters, per, LSUB = 6, 20, 256, 16
    fam, q, t = REF_CODES[cid]
    o = Oracle(fam, q, t)
    rng = np.random.default_rng(2600)
    ys, sent = [], []
    for e in (2.0, 4.0, 6.0):
        c = np.concatenate([np.zeros((per // 2, o.n), np.uint8),
                            ref0.encode(cid, rng.integers(0, 2, (per // 2, o.l)).astype(np.uint8))])
        sent.append(c)
        ys.append(awgn_llr(rng, c, o.l / o.n, e))
    y = np.concatenate(ys)
    lsel = np.concatenate([np.arange(k * per, k * per + LSUB) for k in range(3)])
    d = dict(y=y, sent=np.packbits(np.concatenate(sent), axis=1), iterations=np.array(iters), lsel=lsel)
    cases = [(v, rule, lib, utype) for v in sorted(REF_VARIANTS)
             for rule, (lib, utype) in (("o0", (ref0, 0)), ("o1", (ref1, 0)), ("o2", (ref1, 1)))]

    def run(case):
        v, rule, lib, utype = case
        parts = [lib.minsum(cid, v, iters, utype, y[i:i + 32]) for i in range(0, len(y), 32)]
        return case, [np.concatenate([p[k] for p in parts]) for k in range(4)]

    with ThreadPoolExecutor(8) as ex:
        for (v, rule, _, _), (b, L, it, st) in ex.map(run, cases):
            key = "v%d_%s" % (v, rule)
            d[key + "_b"] = np.packbits(b, axis=1)
            d[key + "_L"] = L[lsel]
            d[key + "_it"] = it.astype(np.uint16)
            d[key + "_st"] = st.astype(np.int8)
            print(key, "failures", int((st != 0).sum()), flush=True)
    np.savez_compressed(os.path.join(HERE, "minsum_bch255_231_headline.npz"), **d)


def main():
    ref0, ref1 = RefLib.get(0), RefLib.get(1)
    if sys.argv[1:] == ["headline"]:
        return headline_golden(ref0, ref1)
    if sys.argv[1:] == ["alt"]:
        return alt_golden(ref0, ref1)
    if sys.argv[1:] == ["mult"]:
        return mult_golden(ref0)
    alt_golden(ref0, ref1)
    mult_golden(ref0)
    headline_golden(ref0, ref1)
    with open(os.path.join(HERE, "constants.json"), "w") as f:
        json.dump(constants(ref0), f, indent=1)
    with open(os.path.join(HERE, "exercises.json"), "w") as f:
        json.dump(exercises(ref0), f, indent=1, ensure_ascii=False)

    # encode + hard decode
    hard_frames = {0: 256, 1: 128, 4: 256, 5: 256, 6: 192, 8: 256, 9: 256, 10: 96, 11: 96, 13: 96}
    for cid, frames in hard_frames.items():
        fam, q, t = REF_CODES[cid]
        o = Oracle(fam, q, t)
        rng = np.random.default_rng(1000 + cid)
        hi = 2 if fam == BCH else 1 << q
        msg = rng.integers(0, hi, (frames, o.l)).astype(np.uint8)
        msg[0] = 0
        msg[1] = hi - 1
        cw = ref0.encode(cid, msg)
        np.savez_compressed(os.path.join(HERE, "encode_%s.npz" % NAMES[cid]), msg=msg, cw=cw)
        nerr = rng.integers(0, t + 3, frames)
        rx = np.stack([corrupt(rng, o, cw[f], int(nerr[f])) for f in range(frames)])
        d = dict(rx=rx, cw=cw, nerr=nerr.astype(np.int32))
        for alg, name in ((PGZ, "pgz"), (BM, "bm"), (EUKLID, "euklid")):
            out, st, msgs = ref0.correct(cid, alg, rx)
            d["out_" + name] = out
            d["status_" + name] = st
            d["recheck_" + name] = np.array(["not a codeword" in m for m in msgs])
            d["notsolvable_" + name] = np.array(["not solvable" in m for m in msgs])
        np.savez_compressed(os.path.join(HERE, "hard_%s.npz" % NAMES[cid]), **d)

    # min-sum: (cid, iterations, frames, ebno list)
    soft = [(0, 10, 64, (1.0, 4.0, 7.0)), (4, 50, 32, (3.0, 6.0)), (5, 10, 64, (2.0, 4.0, 6.0)),
            (6, 20, 16, (4.0, 6.0))]
    for cid, iters, frames, ebnos in soft:
        fam, q, t = REF_CODES[cid]
        o = Oracle(fam, q, t)
        rng = np.random.default_rng(2000 + cid)
        per = frames // len(ebnos)
        ys, sent = [], []
        for e in ebnos:
            zero = np.zeros((per // 2, o.n), np.uint8)
            rnd = ref0.encode(cid, rng.integers(0, 2, (per - per // 2, o.l)).astype(np.uint8))
            c = np.concatenate([zero, rnd])
            sent.append(c)
            ys.append(awgn_llr(rng, c, o.l / o.n, e))
        y = np.concatenate(ys)
        # a few hand-made edge frames: exact zeros, ties, a huge and a tiny LLR
        y[0, :5] = [0.0, 0.5, 0.5, -0.5, 0.0]
        y[1, :3] = [1e30, 1e-30, -1e-38]
        d = dict(y=y, sent=np.concatenate(sent), iterations=np.array(iters))
        for v in sorted(REF_VARIANTS):
            for rule, (lib, utype) in (("o0", (ref0, 0)), ("o1", (ref1, 0)), ("o2", (ref1, 1))):
                b, L, it, st = lib.minsum(cid, v, iters, utype, y)
                key = "v%d_%s" % (v, rule)
                d[key + "_b"] = np.packbits(b, axis=1)
                d[key + "_L"] = L
                d[key + "_it"] = it.astype(np.uint16)
                d[key + "_st"] = st.astype(np.int8)
        np.savez_compressed(os.path.join(HERE, "minsum_%s.npz" % NAMES[cid]), **d)

    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("golden files written, %.1f KiB total" % (total / 1024))


if __name__ == "__main__":
    main()
