#!/usr/bin/env python3
"""xchain2.py S_COST M_COST max_odd r_max -- the search behind tools/kgen4_prog.py X_DIGITS (round 5): for every digit set D = {1} + up to
r_max odd powers <= max_odd, the OPTIMAL signed recoding x = sum d_i 2^i with d_i in +-D u {0} (dynamic programme over (bit, carry): a
millisecond per set, where tools/exp/xchain.py's recursion took seconds and was only run over small powers) and the cheapest chain of
cyclotomic squarings / multiplications producing the table (depth-first); costs in instructions of the shipped routines (4515 / 12500)
or in fqmul (18 / 54).  Result: D = {1, 15, 19} (and, equal, {1, 17, 35}): 62 S + 13 M against {1, 5, 9, 13}'s 61 S + 15 M.
(`python tools/exp/xchain2.py 4515 12500 63 2` ranks the sets by the loop cost first; the table search of large powers is slow.)"""
import itertools, functools, sys, time
X = 4965661367192848881
NB = X.bit_length()
def recode(D):
    """min number of non-zero digits d in +-D with sum d_i 2^i = X; returns (count, digits) -- DP over (bit, carry)"""
    digs = sorted(set(D) | {-d for d in D})
    maxd = max(D)
    INF = 10**9
    # forward DP: state carry c at position i: value consumed so far; v = bit_i + c
    from collections import defaultdict
    cur = {0: (0, None)}
    hist = []
    L = NB + 8
    for i in range(L):
        bit = (X >> i) & 1
        nxt = {}
        for c, (cnt, _) in cur.items():
            v = bit + c
            if v % 2 == 0:
                nc = v // 2
                if nc not in nxt or nxt[nc][0] > cnt: nxt[nc] = (cnt, (c, 0))
            else:
                for d in digs:
                    nc = (v - d) // 2
                    if abs(nc) > maxd: continue
                    if nc not in nxt or nxt[nc][0] > cnt + 1: nxt[nc] = (cnt + 1, (c, d))
        hist.append(nxt)
        cur = nxt
    # best end: carry 0 at some position >= NB-1 with all higher bits zero; choose minimal count then minimal length
    best = None
    for i in range(NB - 6, L):
        if (X >> (i + 1)) == 0 and 0 in hist[i]:
            cnt = hist[i][0][0]
            if best is None or (cnt, i) < (best[0], best[1]): best = (cnt, i)
    cnt, end = best
    digits = []
    c = 0
    for i in range(end, -1, -1):
        pc, d = hist[i][c][1]
        digits.append(d); c = pc
    digits.reverse()
    while digits and digits[-1] == 0: digits.pop()
    assert sum(d << i for i, d in enumerate(digits)) == X
    return cnt, digits
def pre_cost(D, S_COST=18, M_COST=54):
    """cheapest (cost, S, M, steps) producing all of D from 1 by squarings (2a) and multiplications (a+b, a-b: the inverse is the conjugate)"""
    target = set(D) - {1}
    best = [None]
    mx = max(D)
    def dfs(have, steps, S, M):
        cost = S_COST * S + M_COST * M
        if best[0] is not None and cost >= best[0][0]: return
        if target <= set(have):
            best[0] = (cost, S, M, list(steps)); return
        if len(steps) >= 9: return
        cands = set()
        for a in have:
            if 2 * a <= 2 * mx and 2 * a not in have: cands.add((2 * a, 'S', a, a))
            for b in have:
                if a < b: continue
                for s in (a + b, a - b):
                    if 0 < s <= 2 * mx and s not in have: cands.add((s, 'M', a, b))
        # prefer candidates that are targets
        for (v, kind, a, b) in sorted(cands, key=lambda t: (t[0] not in target, t[1] == 'M', t[0])):
            dfs(have + [v], steps + [(v, kind, a, b)], S + (kind == 'S'), M + (kind == 'M'))
    dfs([1], [], 0, 0)
    return best[0]
if __name__ == "__main__":
    SC, MC = float(sys.argv[1]), float(sys.argv[2])
    maxodd = int(sys.argv[3]); rmax = int(sys.argv[4])
    odds = list(range(3, maxodd + 1, 2))
    res = []
    t0 = time.time()
    for r in range(0, rmax + 1):
        for sub in itertools.combinations(odds, r):
            D = (1,) + sub
            cnt, digits = recode(D)
            loopS, loopM = len(digits) - 1, cnt - 1
            # lower bound of the table cost: r multiplications
            lb = SC * loopS + MC * (loopM + r)
            res.append((lb, D, loopS, loopM))
    res.sort()
    print("recoded", len(res), "sets in", round(time.time() - t0), "s")
    out = []
    for lb, D, loopS, loopM in res[:400]:
        if out and lb >= min(o[0] for o in out) : 
            pass
        pc = pre_cost(D, SC, MC) if len(D) > 1 else (0, 0, 0, [])
        if pc is None: continue
        out.append((SC * (loopS + pc[1]) + MC * (loopM + pc[2]), D, loopS + pc[1], loopM + pc[2], pc[3]))
    out.sort()
    for o in out[:10]: print(o)
