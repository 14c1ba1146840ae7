"""As deal_ilp.py, but minimising the number of read instructions that are not conflict-free (max multiplicity M_j)."""
import sys
import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds
from scipy.sparse import lil_matrix
from deal_model import SUPPORT_255_231


def solve(sup, NL, NS, W, delta, mod=32, wmax=2, time_limit=60, P=2):
    S = set(sup)
    idx = {s: i for i, s in enumerate(sup)}
    n = len(sup)
    roles = ["t%d" % k for k in range(NL)] + ["h%d" % k for k in range(NL)] + ["s%d" % k for k in range(NS)]
    R = len(roles)
    nx = P * n * R
    nv = nx + R  # + M_r for read roles (unused for tails)
    var = lambda p, s, r: (p * n + idx[s]) * R + r
    rows, lo, hi = [], [], []
    def add(coefs, l, h):
        rows.append(coefs); lo.append(l); hi.append(h)
    for p in range(P):
        for s in sup:
            add([(var(p, s, r), 1) for r in range(R)], 1, 1)
            for k in range(NL):
                if s + 1 in S:
                    add([(var(p, s, k), 1), (var(p, s + 1, NL + k), -1)], 0, 0)
                else:
                    add([(var(p, s, k), 1)], 0, 0)
                if s - 1 not in S:
                    add([(var(p, s, NL + k), 1)], 0, 0)
        for r in range(R):
            add([(var(p, s, r), 1) for s in sup], W, W)
    for r in range(R):
        is_read = r >= NL
        is_write = r < NL or r >= 2 * NL
        for b in range(mod):
            coefs = [(var(0, s, r), 1) for s in sup if s % mod == b]
            coefs += [(var(P - 1, s, r), 1) for s in sup if (s + delta) % mod == b]
            if not coefs:
                continue
            if is_read:
                add(coefs + [(nx + r, -1)], -100, 0)
            if is_write:
                add(coefs, 0, wmax)
    A = lil_matrix((len(rows), nv))
    for i, c in enumerate(rows):
        for j, v in c:
            A[i, j] += v
    cobj = np.zeros(nv)
    cobj[nx + NL:] = 1
    lb = np.zeros(nv); ub = np.ones(nv)
    ub[nx:] = 4; lb[nx:] = 1
    res = milp(c=cobj, constraints=LinearConstraint(A.tocsr(), lo, hi), integrality=np.ones(nv),
               bounds=Bounds(lb, ub), options={"time_limit": time_limit, "disp": False})
    if res.x is None:
        return None, None
    x = np.round(res.x).astype(int)
    out = []
    for p in range(P):
        deal = {r: [] for r in roles}
        for s in sup:
            for r in range(R):
                if x[var(p, s, r)]:
                    deal[roles[r]].append(s)
        out.append(deal)
    return out, x[nx + NL:]


if __name__ == "__main__":
    tl = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for delta in [int(a) for a in sys.argv[1].split(",")]:
        r, M = solve(SUPPORT_255_231, 2, 3, 16, delta, time_limit=tl, P=P)
        print("delta", delta, "M", M, flush=True)
        if r is not None and M.sum() <= 6:
            for p, d in enumerate(r):
                print(" parity", p, d)
