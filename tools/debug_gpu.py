import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, fixtures, oracle_py
from pecaller_amd import PemapDev
name = sys.argv[1] if len(sys.argv) > 1 else 'r150'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
s = fixtures.SETS[name]
ix = fixtures.index()
r1, l1, r2, l2 = fixtures.reads(name)
r1, l1 = r1[:n], l1[:n]
if s['paired']:
    r2, l2 = r2[:n], l2[:n]
dev = PemapDev(0)
dev.build_index(ix['genome'], ix['contig_len'])
dev.set_params(paired=s['paired'], min_dist=0, max_dist=500, min_align=0.85)
m1, m2, mt = dev.map_batch(r1, l1, r2, l2)
ne = 2 * n if s['paired'] else n
dbg = dev.debug_hits(ne)
o = oracle_py.Oracle(ix, paired=s['paired'])
om1, om2, omt, d1, d2 = o.map_batch(r1, l1, r2, l2, debug=True, threads=8)
print('stats', dev.run_stats())
bad = np.nonzero((m1 != om1) | (mt != omt) | ((m2 != om2) if s['paired'] else False))[0]
print('mismatching reads', len(bad), bad[:20])
step = 2 if s['paired'] else 1
nbad_h = 0
for which, od in ((0, d1), (1, d2)):
    if od is None:
        continue
    nh = dbg['n_hits'][which::step]
    dif = np.nonzero(nh != od['n_hits'])[0]
    print('end', which, 'n_hits differ', len(dif), dif[:10], nh[dif[:10]], od['n_hits'][dif[:10]])
    for i in range(n):
        k = min(nh[i], od['n_hits'][i])
        e = step * i + which
        if k == 0: continue
        ok = (np.array_equal(dbg['spot'][e, :k], od['spot'][i, :k]) and np.array_equal(dbg['orient'][e, :k], od['orient'][i, :k]))
        ok2 = np.array_equal(dbg['score'][e, :k].view(np.uint64), od['score'][i, :k].view(np.uint64)) and np.array_equal(dbg['start_i'][e,:k], od['start'][i,:k,1]) and np.array_equal(dbg['start_k'][e,:k], od['start'][i,:k,0])
        ok3 = np.array_equal(dbg['win_start'][e, :k].astype(np.int32), od['win_start'][i, :k]) and np.array_equal(dbg['win_len'][e, :k], od['win_len'][i, :k])
        if not (ok and ok2 and ok3):
            nbad_h += 1
            if nbad_h < 8:
                print('read', i, 'end', which, 'k', k, 'hits ok', ok, 'score ok', ok2, 'win ok', ok3)
                print('  gpu spot', dbg['spot'][e, :min(k,6)], dbg['orient'][e, :min(k,6)], 'score', dbg['score'][e, :min(k,6)], dbg['start_k'][e,:min(k,6)], dbg['start_i'][e,:min(k,6)], 'win', dbg['win_start'][e,:min(k,6)], dbg['win_len'][e,:min(k,6)])
                print('  ora spot', od['spot'][i, :min(k,6)], od['orient'][i, :min(k,6)], 'score', od['score'][i, :min(k,6)], od['start'][i,:min(k,6),0], od['start'][i,:min(k,6),1], 'win', od['win_start'][i,:min(k,6)], od['win_len'][i,:min(k,6)])
print('ends with differing hit records', nbad_h)
for i in bad[:10]:
    print('read', i, 'gpu', m1[i], m2[i] if s['paired'] else None, mt[i], 'ora', om1[i], om2[i] if s['paired'] else None, omt[i], 'len', l1[i])
counts, ins = dev.fetch_pileup()
oc = o.counts()
dc = np.nonzero((counts != oc).any(axis=1))[0]
print('pileup sites differing', len(dc), dc[:20])
for p in dc[:10]:
    print(' ', p, counts[p], oc[p])
oi = o.insertions()
print('ins equal', ins == oi, len(ins), len(oi))
if ins != oi:
    a = set(ins); b = set(oi)
    print(' only gpu', sorted(a - b)[:10]); print(' only ora', sorted(b - a)[:10])
