"""Randomised cross-check of the bit-plane RS / BCH path against the plain-C oracle: random batch sizes, error counts
up to t + 6, both tags; prints one line per code.  (Ad-hoc confidence run, tests/test_gpu_bitslice.py is the gate.)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from checkers import BCH, BM, PGZ, RS, Oracle
from test_gpu_algebraic import TAGS, check_against_oracle, corrupt
import channelcoding_amd as cc

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2024)
for fam, t in ((RS, 16), (RS, 8), (RS, 12), (BCH, 4), (BCH, 7), (RS, 4)):
    o = Oracle(fam, 8, t)
    hi = 2 if fam == BCH else 256
    total = 0
    for alg in (BM, PGZ):
        code = (cc.primitive_bch if fam == BCH else cc.rs)(8, cc.errors(t), TAGS[alg]())
        for rep in range(6):
            frames = int(rng.integers(1, 3000))
            cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
            mx = int(rng.choice([1, t // 2, t, t + 1, t + 6]))
            rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, mx + 1))) for f in range(frames)])
            check_against_oracle(code.correct_batch(rx), o, alg, rx)
            enc = code.encode_batch(o.extract(cw))
            assert np.array_equal(enc, cw)
            total += frames
    print("ok", "RS" if fam == RS else "BCH", t, total, "frames", flush=True)
print("ALL OK")
