"""float32 transcription of the reference's plane fit as Eigen evaluates it, used ONLY to bound how far the
oracle's canonical arithmetic (exact integer moments) can sit from a real Eigen build.  Eigen is absent from
this image, so this is not the reference itself: it follows src/segmentation.cpp:62-102 (centroid by
column mean, centred buffer, covariance = centred^T centred / (n-1), all float32) and :287-307 (GEMV distances,
signed threshold) in every summation order an Eigen 3.4 build could plausibly use, and shares the oracle's restated
3x3 Jacobi SVD for the solve.

ORDERS (name -> how the long float32 sums and the per-row dot are evaluated):
  sequential  strictly ascending index (what a scalar loop does; the column sums of a row-major n x 3 matrix, whose
              three columns do not fill a packet, most likely reduce this way), dot (x a + y b) + z c
  pairwise    numpy's pairwise summation with 8-way unrolled blocks
  packet4     Eigen's linear vectorised redux on 4-lane (SSE) packets: two packet accumulators over strides of 8,
              combined, horizontal add (p0 + p2) + (p1 + p3), scalar tail
  packet8     the same on 8-lane (AVX) packets: strides of 16, halves added, then the 4-lane horizontal add
  gebp320     centroid sequential; the 3 x n . n x 3 product the way a blocked GEMM accumulates it: sequential
              inside depth blocks of 320 (the kc Eigen's blocking heuristic picks for a 32 KiB L1), block sums added
              to the result in order
  dotright    sequential sums, the per-row dot as x a + (y b + z c)
"""
import numpy as np

import oracle

F = np.float32
ORDERS = ("sequential", "pairwise", "packet4", "packet8", "gebp320", "dotright")


def _seq(v):
    return np.cumsum(v, dtype=F)[-1] if v.size else F(0)


def _sum_packet(v, lanes):
    n = v.size
    a2, a1 = (n // (2 * lanes)) * (2 * lanes), (n // lanes) * lanes
    if a1 == 0:
        return _seq(v)
    p0 = v[:lanes].astype(F)
    if a2 >= 2 * lanes:
        body = v[2 * lanes:a2].reshape(-1, 2, lanes)
        p1 = v[lanes:2 * lanes].astype(F)
        if body.shape[0]:
            p0 = np.cumsum(np.concatenate([p0[None], body[:, 0, :]]), axis=0, dtype=F)[-1]
            p1 = np.cumsum(np.concatenate([p1[None], body[:, 1, :]]), axis=0, dtype=F)[-1]
        p0 = (p0 + p1).astype(F)
    if a1 > a2:
        p0 = (p0 + v[a2:a1]).astype(F)
    if lanes == 8:
        p0 = (p0[:4] + p0[4:]).astype(F)
    res = F(F(p0[0] + p0[2]) + F(p0[1] + p0[3]))
    if a1 < n:
        res = np.cumsum(np.concatenate([[res], v[a1:]]), dtype=F)[-1]
    return res


def _sum_blocked(v, kc):
    n = v.size
    if n == 0:
        return F(0)
    full = (n // kc) * kc
    parts = list(np.cumsum(v[:full].reshape(-1, kc), axis=1, dtype=F)[:, -1]) if full else []
    if full < n:
        parts.append(_seq(v[full:]))
    return np.cumsum(np.array([F(0)] + parts, F), dtype=F)[-1]  # C += block, in order


def _sum(v, order, product=False):
    v = np.ascontiguousarray(v, dtype=F)
    if v.size == 0:
        return F(0)
    if order in ("sequential", "dotright") or (order == "gebp320" and not product):
        return _seq(v)
    if order == "pairwise":
        return np.add.reduce(v, dtype=F)  # numpy's pairwise summation with 8-way unrolled blocks
    if order == "packet4":
        return _sum_packet(v, 4)
    if order == "packet8":
        return _sum_packet(v, 8)
    if order == "gebp320":
        return _sum_blocked(v, 320)
    raise ValueError(order)


def plane_float32(xyz, order):
    """estimate_plane_coefficients, src/segmentation.cpp:62-102"""
    n = xyz.shape[0]
    if n < 3:
        return None
    cen = np.array([_sum(xyz[:, k], order) / F(n) for k in range(3)], F)          # :76
    c = (xyz - cen).astype(F)                                                      # :82
    cov = np.empty((3, 3), F)
    for i in range(3):
        for j in range(3):
            cov[i, j] = _sum(c[:, i] * c[:, j], order, product=True)                # :84
    cov = (cov / F(n - 1)).astype(F)                                               # :85
    v, _ = oracle.jacobi_svd3(cov)                                                 # :87 (restated Eigen 3.4 JacobiSVD)
    a, b, cc = v[0, 2], v[1, 2], v[2, 2]                                            # :94
    d = F(F(a * cen[0]) + F(b * cen[1])) + F(cc * cen[2])                          # :99
    return np.array([a, b, cc, d], F)


def segment_float32(pts, cfg, order):
    """Segmenter::segment with the canonical (x, index) tie order and the float32 plane fit above.
    Returns (labels, planes)."""
    pts = np.asarray(pts, F)
    n = pts.shape[0]
    P, I = cfg.number_of_planar_partitions, cfg.number_of_iterations
    labels = np.zeros(n, np.uint32)
    planes = np.zeros((P, 4), F)
    xs = np.lexsort((np.arange(n), pts[:, 0]))
    n_per = n // P
    for s in range(P):
        seg = xs[s * n_per:(s + 1) * n_per]
        if seg.size < 3:
            continue
        xyz = pts[seg, :3]
        z = xyz[:, 2]
        zo = np.lexsort((np.arange(seg.size), z))                                   # :165-168 (ties: canonical)
        zsorted = z[zo]
        above = np.nonzero(zsorted > F(-1.5) * F(cfg.sensor_height_m))[0]           # :171-182
        cut = int(above[0]) if above.size else 0
        rem, remz = zo[cut:], zsorted[cut:]
        n_rep = min(rem.size, cfg.number_of_lower_point_representatives)
        zmean = np.cumsum(remz[:n_rep], dtype=F)[-1] / F(n_rep) if n_rep else F(np.nan)   # :193-197 sequential
        over = np.nonzero(remz > zmean + F(cfg.initial_seed_threshold))[0]          # :202-210
        ground = rem[:int(over[0])] if over.size else rem[:0]
        dead = False
        plane = np.zeros(4, F)
        mask = np.zeros(seg.size, bool)
        for _ in range(I):
            if ground.size < 3:                                                     # :251-259
                dead = True
                break
            p = plane_float32(xyz[ground], order)
            if p is None or not np.isfinite(p).all():
                dead = True
                break
            plane = p
            a, b, c, d = plane
            if order == "dotright":
                dist = (xyz[:, 0] * a + (xyz[:, 1] * b + xyz[:, 2] * c)).astype(F) - d
            else:
                dist = (F(1) * (xyz[:, 0] * a + xyz[:, 1] * b) + xyz[:, 2] * c).astype(F) - d   # :287-291
            thr = F(cfg.orthogonal_distance_threshold) * np.sqrt(F(F(a * a + b * b) + c * c))  # :293
            mask = dist < thr                                                       # :299 signed
            ground = np.nonzero(mask)[0]
        planes[s] = plane
        if dead:
            labels[seg] = 2
        else:
            labels[seg] = np.where(mask, 1, 2)
    return labels, planes
