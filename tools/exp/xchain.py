#!/usr/bin/env python3
"""xchain.py -- search for a cheap way to raise a cyclotomic Fq12 element to BN_X (three times per final exponentiation).

Cost model: a cyclotomic squaring = 18 fqmul, an Fq12 multiplication = 54.  For every digit set
D = {1} + up to three odd powers, the optimal signed recoding x = sum d_i 2^i with d_i in +-D (dynamic programming) and the
cheapest addition chain producing b^d for d in D (depth-first search) are computed.  Result used by tools/kgen4_prog.py
(X_POWERS, X_DIGITS): D = {1, 5, 9, 13}: 61 squarings + 15 multiplications = 1908 against 2358 for the plain NAF of
pow_native (final_exp_native.rs:56-84).  The value b^x does not depend on the chain, so results stay bit-identical."""
import itertools, functools, sys
X = 4965661367192848881
def best_recoding(x, D):
    """min #nonzero digits (d in +-D) s.t. sum d_i 2^i = x; DP over bit position with signed carry (value remaining)."""
    digs = sorted(set(D) | set(-d for d in D))
    maxd = max(D)
    @functools.lru_cache(None)
    def f(v, depth):
        # v: remaining integer to represent (can be negative small), returns (count, length, digits tuple)
        if v == 0: return (0, ())
        if depth > 70: return (10**9, ())
        if v % 2 == 0:
            c, t = f(v // 2, depth + 1)
            return (c, (0,) + t)
        best = (10**9, ())
        for d in digs:
            if (v - d) % 2 == 0 and abs(v - d) < abs(v) * 2 + 2*maxd:
                if abs((v - d)//2) > abs(v) and abs(v) > maxd: continue
                c, t = f((v - d) // 2, depth + 1)
                if c + 1 < best[0]: best = (c + 1, (d,) + t)
        return best
    return f(x, 0)
def pre_cost(D):
    """cheapest way (S squarings, M muls) to get b^d for all d in D from b: BFS over small addition chains (values <= 2*max)."""
    D = sorted(D); target = set(D) - {1}
    if not target: return (0, 0, [])
    best = None
    # iterative deepening over chains
    def dfs(have, steps, S, M):
        nonlocal best
        cost = 18*S + 54*M
        if best is not None and cost >= best[0]: return
        if target <= set(have):
            best = (cost, S, M, list(steps)); return
        if len(steps) >= 8: return
        mx = max(target)
        cands = set()
        for a in have:
            if 2*a <= mx + 1 and 2*a not in have: cands.add((2*a, 'S', a, a))
            for b in have:
                if a < b: continue
                for s in (a + b, a - b):
                    if 0 < s <= mx and s not in have: cands.add((s, 'M', a, b))
        for (v, kind, a, b) in sorted(cands):
            dfs(have + [v], steps + [(v, kind, a, b)], S + (kind == 'S'), M + (kind == 'M'))
    dfs([1], [], 0, 0)
    return best
res = []
odds = [3,5,7,9,11,13,15,17,19,21,23,25,27,29,31]
for r in range(2, 4):
    for sub in itertools.combinations(odds, r):
        D = (1,) + sub
        c, digs = best_recoding(X, D)
        pc = pre_cost(D)
        if r and pc is None: continue
        pS, pM = (pc[1], pc[2]) if r else (0, 0)
        S = len(digs) - 1 + pS
        M = c - 1 + pM
        res.append((18*S + 54*M, D, S, M, digs, pc))
res.sort(key=lambda t: t[0])
for t in res[:8]:
    print(t[0], t[1], "S", t[2], "M", t[3], "len", len(t[4]), "pre", t[5][3] if t[1] != (1,) else None)
best = res[0]
print(best[4])
assert sum(d << i for i, d in enumerate(best[4])) == X
