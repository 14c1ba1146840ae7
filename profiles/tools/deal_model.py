"""LDS bank model of the diagonal min-sum kernel's deal (profiles/tools/deal_model.py).

Banking rules: /opt/skills/guides/MI355X_MICROARCH.md, section LDS (ds_read_b64: groups of 32 lanes, bank pair =
(a/8) mod 32; ds_read_b32 / ds_write_b32: groups of 32 lanes, bank = (a/4) mod 32; an extra distinct address on a busy
bank costs one LDS cycle; a b32 store hides a 2-way conflict behind its 4-cycle register transfer).
"""
import sys

SUPPORT_255_231 = [0,1,2,5,6,7,9,20,24,29,33,35,37,41,42,43,44,45,47,52,53,55,58,59,60,61,63,67,68,70,72,78,80,83,84,88,89,92,94,95,100,102,103,106,109,112,115,117,118,119,120,121,122,124,125,128,129,131,133,136,137,140,142,144,145,147,148,151,153,154,160,161,162,163,167,168,169,170,171,173,174,175,180,182,183,185,189,190,191,195,197,198,201,202,205,206,207,210,211,213,215,217,218,219,220,221,222,224,225,226,229,231]


def round2_paired_table(sup, n, D=7, W=16, gaps=(1, 1)):
    """Port of build_paired_table (csrc/minsum_diag.hip, round 2): the deal the shipped kernel used."""
    np_ = len(gaps)
    in_sup = [0] * (n + 512)
    for s in sup:
        in_sup[s] = 1
    singles_per_lane = D - 2 * np_
    state = [0x9E3779B97F4A7C15]

    def nxt():
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return state[0] >> 33

    def passes(v):
        cnt = [0] * 16
        for s in v:
            cnt[s % 16] += 1
        return max(cnt)

    best_cost, best_low, best_single = 1 << 30, None, None
    for attempt in range(4000):
        if best_cost <= np_ + singles_per_lane:
            break
        used = [0] * (n + 512)
        low = [[] for _ in range(np_)]
        ok = True
        for p in range(np_):
            if not ok:
                break
            c = [s for s in sup if in_sup[s + gaps[p]]]
            i = len(c)
            while i > 1:
                j = nxt() % i
                c[i - 1], c[j] = c[j], c[i - 1]
                i -= 1
            res = [False] * 16
            for ps in range(2):
                if len(low[p]) >= W:
                    break
                for s in c:
                    if used[s] or used[s + gaps[p]] or (ps == 0 and res[s % 16]):
                        continue
                    used[s] = used[s + gaps[p]] = 1
                    res[s % 16] = True
                    low[p].append(s)
                    if len(low[p]) == W:
                        break
            ok = len(low[p]) == W
        if not ok:
            continue
        single = [s for s in sup if not used[s]]
        cost = sum(passes(l) for l in low)
        cnt = [0] * 16
        for s in single:
            cnt[s % 16] += 1
        cost += max(singles_per_lane, max(cnt))
        if cost < best_cost:
            best_cost, best_low, best_single = cost, [list(l) for l in low], single
    out = [[None] * W for _ in range(D)]
    for p in range(np_):
        best_low[p].sort()
        for l in range(W):
            out[2 * p][l] = best_low[p][l]
            out[2 * p + 1][l] = best_low[p][l] + gaps[p]
    cls = [[] for _ in range(16)]
    for s in best_single:
        cls[s % 16].append(s)
    slot = [[] for _ in range(singles_per_lane)]
    turn = 0
    for r in range(16):
        for s in cls[r]:
            tries = 0
            while len(slot[turn % singles_per_lane]) >= W and tries < singles_per_lane:
                tries += 1
                turn += 1
            slot[turn % singles_per_lane].append(s)
            turn += 1
    for g in range(singles_per_lane):
        for l, s in enumerate(slot[g]):
            out[2 * np_ + g][l] = s
    return out


def mult(cols, mod):
    cnt = {}
    for c in cols:
        cnt[c % mod] = cnt.get(c % mod, 0) + 1
    return max(cnt.values())


def cost_round2_layout(deal, nlinks=2):
    """LDS cycles per row and wave beyond the conflict-free count, round-2 layout: CY cells of 8 bytes, frame stride
    272 cells (frame 1 of a half-wave 16 bank pairs later), CN floats at an 8-byte stride with the two frames of a
    half-wave on even / odd dwords."""
    D = len(deal)
    reads = [d for d in range(D) if not (d < 2 * nlinks and d % 2 == 0)]   # heads + singles (tails take the carry)
    writes = [d for d in range(D) if not (d < 2 * nlinks and d % 2 == 1)]  # tails + singles (heads hand on)
    extra_cy = extra_cnr = extra_cnw = 0
    for d in reads:
        cols = [s for s in deal[d]]
        # b64: 32 lanes = frames 0 and 1, bank pair (272 f + col) mod 32
        m = mult(cols + [c + 16 for c in cols], 32)
        extra_cy += 2 * (m - 1)
        # CN b32 read: bank = 2 (col mod 16) + f -> within a frame col mod 16
        m = mult(cols, 16)
        extra_cnr += 2 * (m - 1)
    for d in writes:
        m = mult(deal[d], 16)
        extra_cnw += 2 * (m - 1)
    return extra_cy, extra_cnr, extra_cnw


if __name__ == "__main__":
    deal = round2_paired_table(SUPPORT_255_231, 255)
    for d, row in enumerate(deal):
        print(d, row, "mult16", mult(row, 16))
    print("extra LDS cycles per row and wave (CY reads, CN reads, CN writes):", cost_round2_layout(deal))
