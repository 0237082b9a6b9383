"""float32 transcription of the reference's plane fit as Eigen evaluates it, used ONLY to bound how far the
oracle's canonical arithmetic (exact integer moments) can sit from a real Eigen build.  Eigen is absent from
this image, so this is not the reference itself: it follows src/segmentation.cpp:62-102 (centroid by
column mean, centred buffer, covariance = centred^T centred / (n-1), all float32) and :287-307 (GEMV distances,
signed threshold) with two plausible summation orders -- strictly sequential, and pairwise / blocked like a
vectorised reduction -- and shares the oracle's restated 3x3 Jacobi SVD for the solve."""
import numpy as np

import oracle

F = np.float32


def _sum(v, order):
    v = np.ascontiguousarray(v, dtype=F)
    if v.size == 0:
        return F(0)
    if order == "sequential":
        return np.cumsum(v, dtype=F)[-1]
    return np.add.reduce(v, dtype=F)  # numpy's pairwise summation with 8-way unrolled blocks


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
            cov[i, j] = _sum(c[:, i] * c[:, j], order)                              # :84
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
