"""CPU analysis for the expansion-driven neighbour search: expansions per component, candidates per kd group."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from util import brute_components, load_frame, load_stream_frame

name = sys.argv[1] if len(sys.argv) > 1 else "0000000000"
d2 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
pts = load_stream_frame(name)
obs = pts[oracle.segment(pts, oracle.SegCfg(number_of_planar_partitions=6, number_of_iterations=5))["obstacle_idx"]]
M = obs.shape[0]
a = np.ascontiguousarray(obs[:, :4])
lab = np.zeros(M, np.int32)
nc = C.c_uint32(0)
exp = np.zeros(M, np.uint8)
cfg = oracle.CluCfg(d2, 0.5)
oracle.lib().orc_cluster_trace(a.ctypes.data_as(C.c_void_p), C.c_size_t(16), C.c_uint32(M), C.byref(cfg),
                               lab.ctypes.data_as(C.c_void_p), C.byref(nc), exp.ctypes.data_as(C.c_void_p))
print(f"M={M} clusters={nc.value} expansions={int(exp.sum())} ({100.0 * exp.mean():.1f}% of points)")
root = brute_components(obs, d2)
uniq, inv, cnt = np.unique(root, return_inverse=True, return_counts=True)
e_per = np.bincount(inv, weights=exp)
print(f"true components: {len(uniq)}; largest {cnt.max()} pts; expansions per component: max {int(e_per.max())}, "
      f"top5 {sorted(e_per.astype(int))[-5:]}, sum {int(e_per.sum())}")
# kd pre-order groups: buckets at level D (<= 64 nodes)
pre = oracle.kd_preorder(obs)
P = obs[pre, :3]
r = np.sqrt(d2) * 1.0001 + 1e-3
D = 0
while (M >> D) > 64:
    D += 1
print("D", D, "buckets", 1 << D)


def groups():
    """yield (gid, level, rank, b, e) for every node of the top D levels and every bucket"""
    stack = [(0, 0, M, 0)]
    while stack:
        rank, b, e, lvl = stack.pop()
        if b >= e:
            continue
        if lvl == D:
            yield ("bucket", rank, e - b)
            continue
        mid = b + (e - b) // 2
        yield ("node", rank, 1)
        stack.append((rank + 1 + (mid - b), mid + 1, e, lvl + 1))
        stack.append((rank + 1, b, mid, lvl + 1))


def traverse(lo, hi):
    """candidate rank intervals for the box [lo, hi] walking the top D levels (like nb_traverse)"""
    out = []
    stack = [(0, 0, M, 0)]
    while stack:
        rank, b, e, lvl = stack.pop()
        if b >= e:
            continue
        if lvl == D:
            out.append((rank, e - b))
            continue
        mid = b + (e - b) // 2
        s = P[rank, lvl % 3]
        out.append((rank, 1))
        if mid + 1 < e and s <= hi[lvl % 3]:
            stack.append((rank + 1 + (mid - b), mid + 1, e, lvl + 1))
        if mid > b and s >= lo[lvl % 3]:
            stack.append((rank + 1, b, mid, lvl + 1))
    out.sort()
    return out


Ts, Ns, Tq = [], [], []
glist = list(groups())
rng = np.random.default_rng(0)
for kind, rank, n in glist:
    q = P[rank:rank + n]
    iv = traverse(q.min(0) - r, q.max(0) + r)
    # merge adjacent intervals
    merged = []
    for a0, c0 in iv:
        if merged and merged[-1][0] + merged[-1][1] == a0:
            merged[-1][1] += c0
        else:
            merged.append([a0, c0])
    Ts.append(sum(c for _, c in merged))
    Ns.append(len(merged))
Ts, Ns = np.array(Ts), np.array(Ns)
isb = np.array([g[0] == "bucket" for g in glist])
print(f"groups {len(glist)}: bucket T mean {Ts[isb].mean():.0f} p50 {np.median(Ts[isb]):.0f} p90 {np.percentile(Ts[isb], 90):.0f} max {Ts[isb].max()};"
      f" intervals mean {Ns[isb].mean():.1f} max {Ns[isb].max()}")
print(f"single-node groups T mean {Ts[~isb].mean():.0f} max {Ts[~isb].max()}; intervals mean {Ns[~isb].mean():.1f}")
# candidates weighted by expansions: which group does each expanded point belong to
rank_of = np.empty(M, np.int64)
rank_of[pre] = np.arange(M)
grank = np.array([g[1] for g in glist])
gn = np.array([g[2] for g in glist])
order = np.argsort(grank)
gr_sorted, gn_sorted, T_sorted = grank[order], gn[order], Ts[order]
er = rank_of[np.nonzero(exp)[0]]
# group of a rank: the last group whose rank <= er and rank + n > er (buckets are contiguous intervals, nodes single)
pos = np.searchsorted(gr_sorted, er, side="right") - 1
ok = er < gr_sorted[pos] + gn_sorted[pos]
assert ok.all()
print(f"candidates per EXPANSION (group T): mean {T_sorted[pos].mean():.0f}, total {int(T_sorted[pos].sum())} "
      f"= {T_sorted[pos].sum() / 64:.0f} wave steps; list entries of expanded points would be far fewer")

# ---- conservative components from group adjacency (group g is united with every group its candidate intervals touch)
parent = np.arange(len(glist))


def find(x):
    while parent[x] != x:
        parent[x] = parent[parent[x]]
        x = parent[x]
    return x


gid_of_rank = np.empty(M, np.int64)
for gi, (kind, rank, n) in enumerate(glist):
    gid_of_rank[rank:rank + n] = gi
for gi, (kind, rank, n) in enumerate(glist):
    q = P[rank:rank + n]
    for a0, c0 in traverse(q.min(0) - r, q.max(0) + r):
        for gj in np.unique(gid_of_rank[a0:a0 + c0]):
            ra, rb = find(gi), find(int(gj))
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
groot = np.array([find(i) for i in range(len(glist))])
sup = groot[gid_of_rank[rank_of]]  # per point (obstacle index)
u2, inv2 = np.unique(sup, return_inverse=True)
e2 = np.bincount(inv2, weights=exp)
print(f"group-adjacency super-components: {len(u2)}; expansions: max {int(e2.max())} of {int(e2.sum())}, top5 {sorted(e2.astype(int))[-5:]}")

# ---- conservative components from a uniform grid of edge c >= d with 26-adjacency of occupied cells
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
for scale, reach in ((1.001, 1), (0.5005, 2), (0.3337, 3)):
    c = np.sqrt(d2) * scale
    ijk = np.floor(obs[:, :3].astype(np.float64) / c).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1 + 2 * reach
    key = ((ijk[:, 0] + reach) * dims[1] + (ijk[:, 1] + reach)) * dims[2] + (ijk[:, 2] + reach)
    ukey, cell_of = np.unique(key, return_inverse=True)
    rows, cols = [], []
    rr = range(-reach, reach + 1)
    for dx in rr:
        for dy in rr:
            for dz in rr:
                if (dx, dy, dz) <= (0, 0, 0):
                    continue
                nk = ukey + (dx * dims[1] + dy) * dims[2] + dz
                pos = np.searchsorted(ukey, nk)
                pos[pos >= len(ukey)] = 0
                hit = ukey[pos] == nk
                rows.append(np.nonzero(hit)[0])
                cols.append(pos[hit])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    g = coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(len(ukey), len(ukey)))
    ncc, cl = connected_components(g, directed=False)
    e3 = np.bincount(cl[cell_of], weights=exp)
    print(f"grid c={c:.3f} reach {reach}: {len(ukey)} occupied cells, {len(rows)} adjacent pairs, {ncc} super-components; "
          f"expansions max {int(e3.max())}, top5 {sorted(e3.astype(int))[-5:]}")
