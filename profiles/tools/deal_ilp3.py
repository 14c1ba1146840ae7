"""Deal for the shipped LDS layout (two frames of a 32-lane group 16 bank pairs apart, so a slot instruction is
conflict-free iff its 16 diagonals are distinct modulo 16): minimise the number of read instructions (heads + singles)
with a doubled residue class, then the same for the write instructions (tails + singles).  scipy / HiGHS."""
import sys
import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds
from scipy.sparse import lil_matrix
from deal_model import SUPPORT_255_231, mult


def solve(sup, NL, NS, W, mod=16, time_limit=120, wweight=0.0):
    S = set(sup)
    idx = {s: i for i, s in enumerate(sup)}
    n = len(sup)
    roles = ["t%d" % k for k in range(NL)] + ["h%d" % k for k in range(NL)] + ["s%d" % k for k in range(NS)]
    R = len(roles)
    nx = n * R
    # slack variables: excess[r][b] >= count - 1 for every role and bank (reads weighted 1, writes wweight)
    ne = R
    nv = nx + ne
    var = lambda s, r: idx[s] * R + r
    ev = lambda r, b: nx + r
    rows, lo, hi = [], [], []
    def add(coefs, l, h):
        rows.append(coefs); lo.append(l); hi.append(h)
    for s in sup:
        add([(var(s, r), 1) for r in range(R)], 1, 1)
        for k in range(NL):
            if s + 1 in S:
                add([(var(s, k), 1), (var(s + 1, NL + k), -1)], 0, 0)
            else:
                add([(var(s, k), 1)], 0, 0)
            if s - 1 not in S:
                add([(var(s, NL + k), 1)], 0, 0)
    for r in range(R):
        add([(var(s, r), 1) for s in sup], W, W)
        for b in range(mod):
            coefs = [(var(s, r), 1) for s in sup if s % mod == b]
            add(coefs + [(ev(r, b), -1)], -100, 0)  # count <= M_r
    A = lil_matrix((len(rows), nv))
    for i, c in enumerate(rows):
        for j, v in c:
            A[i, j] += v
    cobj = np.zeros(nv)
    for r in range(R):
        is_read = r >= NL
        is_write = r < NL or r >= 2 * NL
        cobj[nx + r] = (1.0 if is_read else 0.0) + (wweight if is_write else 0.0)
    lb = np.zeros(nv); ub = np.ones(nv); ub[nx:] = 8; lb[nx:] = 1
    integ = np.ones(nv)
    res = milp(c=cobj, constraints=LinearConstraint(A.tocsr(), lo, hi), integrality=integ,
               bounds=Bounds(lb, ub), options={"time_limit": time_limit, "disp": False})
    if res.x is None:
        return None
    x = np.round(res.x[:nx]).astype(int)
    deal = {r: [] for r in roles}
    for s in sup:
        for r in range(R):
            if x[var(s, r)]:
                deal[roles[r]].append(s)
    return deal, res.fun


if __name__ == "__main__":
    d, f = solve(SUPPORT_255_231, 2, 3, 16, time_limit=int(sys.argv[1]) if len(sys.argv) > 1 else 120)
    print("objective", f)
    table = []
    for k in range(2):
        t = sorted(d["t%d" % k]); h = [s + 1 for s in t]
        assert sorted(d["h%d" % k]) == h
        table += [t, h]
    for k in range(3):
        table.append(sorted(d["s%d" % k]))
    for i, row in enumerate(table):
        print(i, row, "mult16", mult(row, 16))
    print("table =", [x for row in table for x in row])
