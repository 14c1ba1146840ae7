"""Search for a deal of the diagonals of H's first row to (frame parity, slot, lane) whose LDS accesses are
bank-conflict-free (profiles/tools/deal_search.py; model: deal_model.py / MI355X_MICROARCH.md, LDS section).

Layout assumed: CY cells of 8 bytes and CN floats of 4 bytes, both with a frame stride congruent to DELTA modulo 32
cells, so that the two frames of a 32-lane group see each other DELTA banks (bank pairs) apart.  A deal per frame
parity: NL links (tail s, head s + 1) + D - 2 NL single slots, 16 lanes each.  Reads: heads + singles; writes:
tails + singles.  Energy = extra LDS cycles per row (reads: max multiplicity - 1 per instruction; writes: beyond
2-way) with the number of colliding pairs as the tie-break.
"""
import random
import sys
from deal_model import SUPPORT_255_231


def search(sup, D, NL, W, delta, seed, iters=400000, mod=32, verbose=False):
    rnd = random.Random(seed)
    S = set(sup)
    adj = [s for s in sup if s + 1 in S]
    nsingle = D - 2 * NL

    def initial():
        # random maximal-ish matching of NL*W adjacent pairs
        while True:
            used = set()
            pairs = []
            cand = adj[:]
            rnd.shuffle(cand)
            for s in cand:
                if s in used or s + 1 in used:
                    continue
                pairs.append(s)
                used.add(s)
                used.add(s + 1)
                if len(pairs) == NL * W:
                    break
            if len(pairs) == NL * W:
                break
        singles = [s for s in sup if s not in used]
        rnd.shuffle(singles)
        links = [pairs[i * W:(i + 1) * W] for i in range(NL)]
        sing = [singles[i * W:(i + 1) * W] for i in range(nsingle)]
        return links, sing

    st = [initial(), initial()]  # per parity

    def instr_sets(st):
        # returns lists of (residues parity0, residues parity1) for read instrs and write instrs
        reads, writes = [], []
        for k in range(NL):
            reads.append(([s + 1 for s in st[0][0][k]], [s + 1 for s in st[1][0][k]]))
            writes.append((st[0][0][k], st[1][0][k]))
        for k in range(nsingle):
            reads.append((st[0][1][k], st[1][1][k]))
            writes.append((st[0][1][k], st[1][1][k]))
        return reads, writes

    def energy(st):
        reads, writes = instr_sets(st)
        e_cyc = 0
        e_col = 0
        for a, b in reads:
            cnt = [0] * mod
            for s in a:
                cnt[s % mod] += 1
            for s in b:
                cnt[(s + delta) % mod] += 1
            e_cyc += max(cnt) - 1
            e_col += sum(c - 1 for c in cnt if c > 1)
        w_cyc = 0
        for a, b in writes:
            cnt = [0] * mod
            for s in a:
                cnt[s % mod] += 1
            for s in b:
                cnt[(s + delta) % mod] += 1
            w_cyc += max(0, max(cnt) - 2)
            e_col += 0.1 * sum(c - 1 for c in cnt if c > 1)
        return 100 * e_cyc + 100 * w_cyc + e_col, e_cyc, w_cyc

    cur = energy(st)
    best = (cur, None)
    T = 30.0
    for it in range(iters):
        T = max(0.3, 30.0 * (1 - it / iters))
        p = rnd.randrange(2)
        links, sing = st[p]
        mv = rnd.random()
        undo = None
        if mv < 0.5:  # swap two singles between slots
            a, b = rnd.sample(range(nsingle), 2) if nsingle > 1 else (0, 0)
            i, j = rnd.randrange(W), rnd.randrange(W)
            sing[a][i], sing[b][j] = sing[b][j], sing[a][i]
            undo = ("ss", a, i, b, j)
        elif mv < 0.7 and NL > 1:  # swap links between link slots
            a, b = rnd.sample(range(NL), 2)
            i, j = rnd.randrange(W), rnd.randrange(W)
            links[a][i], links[b][j] = links[b][j], links[a][i]
            undo = ("ll", a, i, b, j)
        else:  # re-link: break one link, make another out of two singles that are adjacent
            pos = {}
            for k in range(nsingle):
                for i, s in enumerate(sing[k]):
                    pos[s] = (k, i)
            cands = [s for s in adj if s in pos and s + 1 in pos]
            if not cands:
                continue
            u = rnd.choice(cands)
            a, i = rnd.randrange(NL), rnd.randrange(W)
            old = links[a][i]
            (k1, i1), (k2, i2) = pos[u], pos[u + 1]
            links[a][i] = u
            if rnd.random() < 0.5:
                sing[k1][i1], sing[k2][i2] = old, old + 1
            else:
                sing[k1][i1], sing[k2][i2] = old + 1, old
            undo = ("rl", a, i, old, k1, i1, k2, i2, u)
        new = energy(st)
        d = new[0] - cur[0]
        if d <= 0 or rnd.random() < pow(2.718281828, -d / T):
            cur = new
            if cur[0] < best[0][0]:
                best = (cur, ([[list(x) for x in st[q][0]] for q in range(2)], [[list(x) for x in st[q][1]] for q in range(2)]))
                if verbose:
                    print(it, cur)
                if cur[1] == 0 and cur[2] == 0 and cur[0] < 1:
                    break
        else:
            if undo[0] == "ss":
                _, a, i, b, j = undo
                sing[a][i], sing[b][j] = sing[b][j], sing[a][i]
            elif undo[0] == "ll":
                _, a, i, b, j = undo
                links[a][i], links[b][j] = links[b][j], links[a][i]
            else:
                _, a, i, old, k1, i1, k2, i2, u = undo
                links[a][i] = old
                sing[k1][i1], sing[k2][i2] = u, u + 1
    return best


if __name__ == "__main__":
    deltas = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 11, 21, 24, 16]
    for delta in deltas:
        for seed in range(3):
            b = search(SUPPORT_255_231, 7, 2, 16, delta, seed, iters=200000)
            print("delta", delta, "seed", seed, "energy", b[0])
            if b[0][1] == 0 and b[0][2] == 0:
                print(b[1])
                break
