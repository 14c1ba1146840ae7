"""Loaders for the committed golden vectors (tests/golden/, made by make_golden.py
from the real reference).  Data only; numpy.load with allow_pickle=False."""
import json
import os

import numpy as np

from checkers import BCH, REF_CODES, REF_VARIANTS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = {0: "bch15_7", 1: "bch15_5", 4: "bch31_16", 5: "bch63_45", 6: "bch255_231", 8: "rs7_3", 9: "rs15_9",
         10: "rs255_223", 11: "bch127_113", 13: "bch63_39"}
SOFT_CIDS = (0, 4, 5, 6)
HARD_CIDS = tuple(sorted(NAMES))
RULES = {"o0": 0, "o1": 1, "o2": 2}


def constants():
    with open(os.path.join(GOLDEN, "constants.json")) as f:
        return {int(k): v for k, v in json.load(f).items()}


def exercises():
    with open(os.path.join(GOLDEN, "exercises.json")) as f:
        return json.load(f)


def load(kind, cid):
    return np.load(os.path.join(GOLDEN, "%s_%s.npz" % (kind, NAMES[cid])), allow_pickle=False)


def minsum_cases(cid):
    """Yields (variant_id, (oracle_variant, alpha, beta), rule_id, b, L, it, st) for one code."""
    d = load("minsum", cid)
    n = REF_CODES[cid]
    for v in sorted(REF_VARIANTS):
        for rule, rid in RULES.items():
            key = "v%d_%s" % (v, rule)
            width = d["y"].shape[1]
            b = np.unpackbits(d[key + "_b"], axis=1)[:, :width]
            yield v, REF_VARIANTS[v], rid, b, d[key + "_L"], d[key + "_it"].astype(np.uint32), d[key + "_st"].astype(
                np.int32)


def minsum_headline_cases():
    """BCH(255,231) MS<20> family, 256 frames at each of 2 / 4 / 6 dB (minsum_bch255_231_headline.npz): y, iterations,
    the frame indices whose L is stored, and an iterator of (variant_id, (oracle_variant, alpha, beta), rule_id, b, L
    [of those frames], it, st)."""
    d = np.load(os.path.join(GOLDEN, "minsum_bch255_231_headline.npz"), allow_pickle=False)
    y = d["y"]
    width = y.shape[1]

    def cases():
        for v in sorted(REF_VARIANTS):
            for rule, rid in RULES.items():
                key = "v%d_%s" % (v, rule)
                b = np.unpackbits(d[key + "_b"], axis=1)[:, :width]
                yield v, REF_VARIANTS[v], rid, b, d[key + "_L"], d[key + "_it"].astype(np.uint32), d[key + "_st"].astype(
                    np.int32)

    return y, int(d["iterations"]), d["lsel"], cases()


ALT_CIDS = (0, 1, 5, 6)


def minsum_alt_cases(cid):
    """(H_alt, y, iterations, iterator of (variant_id, (oracle_variant, alpha, beta), rule_id, b, L, it, st))."""
    d = np.load(os.path.join(GOLDEN, "minsum_alt.npz"), allow_pickle=False)
    pre = "c%d_" % cid
    y = d[pre + "y"]
    width = y.shape[1]
    H = np.unpackbits(d[pre + "H"], axis=1)[:, :width]

    def cases():
        for v in sorted(REF_VARIANTS):
            for rule in ("o0", "o1"):
                key = pre + "v%d_%s" % (v, rule)
                b = np.unpackbits(d[key + "_b"], axis=1)[:, :width]
                yield v, REF_VARIANTS[v], RULES[rule], b, d[key + "_L"], d[key + "_it"].astype(np.uint32), d[
                    key + "_st"].astype(np.int32)

    return H, y, int(d[pre + "iterations"]), cases()


MULT_CIDS = (0, 5, 6, 8, 9, 10)


def mult_case(cid):
    """multiplication_tag vectors: msg, cw = msg*g, rx (codewords and arbitrary words), quot = rx / g."""
    d = np.load(os.path.join(GOLDEN, "mult.npz"), allow_pickle=False)
    pre = "c%d_" % cid
    return d[pre + "msg"], d[pre + "cw"], d[pre + "rx"], d[pre + "quot"]
