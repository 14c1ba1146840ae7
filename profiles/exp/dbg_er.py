import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
os.environ.setdefault("CC_AMD_PLANES_MIN_WORK", "0")
import numpy as np
from checkers import BCH, BM, EUKLID, Oracle
from test_gpu_algebraic import REF_CODES, make_code
cid = 10
o = Oracle(*REF_CODES[cid])
print("code", REF_CODES[cid], "n", o.n, "t", o.t, "BM", BM, "EUKLID", EUKLID)
rng = np.random.default_rng(6200 + cid)
hi = 2 if o.family == BCH else 1 << o.q
frames = 200
cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
rx = cw.copy(); per = []; nerrs = []
for f in range(frames):
    ne = int(rng.integers(0, 2 * o.t + 1))
    er = sorted(rng.choice(o.n, ne, replace=False).tolist())
    for e in er: rx[f, e] = 0
    nerr = int(rng.integers(0, max(1, (2 * o.t - ne) // 2 + 2)))
    free = [p for p in range(o.n) if p not in er]
    for p in rng.choice(free, nerr, replace=False):
        rx[f, p] ^= 1 if o.family == BCH else int(rng.integers(1, hi))
    per.append(er); nerrs.append(nerr)
for alg in (BM, EUKLID):
    res = make_code(cid, alg).correct_batch(rx, erasures=per)
    bad = []
    for f in range(frames):
        out, nerr, st, ub = o.correct_hard(alg, rx[f], per[f])
        if (res["status"][f] == 0) != (st[0] == 0) or (st[0] == 0 and not np.array_equal(res["out"][f], out[0])):
            bad.append((f, len(per[f]), nerrs[f], int(res["status"][f]), int(st[0]), int(res["nerr"][f]), int(nerr[0])))
    print("alg", alg, "mismatches (frame, rho, errors, dev status, oracle status, dev nerr, oracle nerr):", bad[:12], len(bad))
