#!/usr/bin/env python3
"""Per-function instruction statistics of a hipcc -save-temps gfx950 .s file."""
import sys, re
lines = open(sys.argv[1]).read().split('\n')
cur = None; stats = {}
for l in lines:
    m = re.match(r'^([A-Za-z_][\w.$]*):\s*(;.*)?$', l)
    if m and not l.startswith('.L'):
        cur = m.group(1); stats[cur] = dict(n=0, scratch=0, acc=0, mov=0, mad=0, call=0, ds=0, glob=0, nop=0)
        continue
    if cur is None: continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if re.match(r'^[vs]_|^ds_|^global_|^scratch_|^buffer_|^flat_', op):
        st = stats[cur]; st['n'] += 1
        if op.startswith('scratch_'): st['scratch'] += 1
        if op.startswith('v_accvgpr'): st['acc'] += 1
        if op.startswith('v_mov'): st['mov'] += 1
        if op.startswith('v_mad_u64'): st['mad'] += 1
        if op.startswith('s_swappc'): st['call'] += 1
        if op.startswith('ds_'): st['ds'] += 1
        if op.startswith('global_'): st['glob'] += 1
        if op == 's_nop': st['nop'] += 1
tot = 0
for k, v in stats.items():
    tot += v['n']
    if v['n'] > 50:
        print(f"{k[:64]:64s} n={v['n']:6d} mad={v['mad']:5d} mov={v['mov']:5d} acc={v['acc']:5d} scratch={v['scratch']:4d} ds={v['ds']:4d} glob={v['glob']:4d} call={v['call']:3d} nop={v['nop']}")
print("total instructions:", tot)
