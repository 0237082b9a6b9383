"""Shared helpers of the test-suite: golden frames, synthetic clouds, partition comparison."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = ["0000000000", "0000000077", "0000000153"]

STREAM = os.path.join(GOLDEN, "stream")
# the two configurations the 154-frame stream has goldens for: BASELINE configs[1]/[3] and the shipped defaults
STREAM_CONFIGS = {
    "p6i5_d025q05": (dict(number_of_planar_partitions=6, number_of_iterations=5),
                     dict(distance_squared=0.25, cluster_quality=0.5)),
    "p2i3_d018q05": (dict(number_of_planar_partitions=2, number_of_iterations=3),
                     dict(distance_squared=0.18, cluster_quality=0.5)),
}

_frames = None
_gold = None
_stream_gold = None


def load_frame(name):
    """(n,4) float32 x y z intensity, bit-identical to the reference's data/<name>.pcd payload"""
    global _frames
    # the lazily read archive keeps a file handle: a forked worker (tests/golden/make_eigen_bound.py) that inherited
    # the parent's would share its file OFFSET with its siblings and read garbage, so the handle is per process
    if _frames is None or _frames[0] != os.getpid():
        _frames = (os.getpid(), np.load(os.path.join(GOLDEN, "frames.npz")))
    z = _frames[1]
    pts = np.empty((z[f"{name}_xyz_mm"].shape[0], 4), np.float32)
    pts[:, :3] = (z[f"{name}_xyz_mm"] / 1000.0).astype(np.float32)
    pts[:, 3] = (z[f"{name}_intensity_c"] / 100.0).astype(np.float32)
    for r, c in z[f"{name}_negzero"]:
        pts[r, c] = -0.0
    return pts


def pack_frame(pts):
    """lossless repack of an (n,4) float32 x y z intensity frame whose coordinates are exact 1 mm multiples and
    intensities exact 0.01 multiples (every reference data/*.pcd frame): scan-order deltas of the integer
    millimetres, zigzag, three byte planes per coordinate, intensity bytes, negative-zero list; xz"""
    import lzma
    q = np.rint(pts[:, :3].astype(np.float64) * 1000.0).astype(np.int32)
    qi = np.rint(pts[:, 3].astype(np.float64) * 100.0).astype(np.int64)
    rec = np.empty_like(pts)
    rec[:, :3] = (q / 1000.0).astype(np.float32)
    rec[:, 3] = (qi / 100.0).astype(np.float32)
    assert np.array_equal(rec, pts), "frame is not 1 mm / 0.01 quantised"
    assert qi.min() >= 0 and qi.max() < 256
    d = np.diff(q, axis=0, prepend=np.zeros((1, 3), np.int32))
    zz = ((d << 1) ^ (d >> 31)).astype(np.uint32)
    assert zz.max() < (1 << 24)
    planes = b"".join(np.ascontiguousarray(((zz >> (8 * k)) & 255).astype(np.uint8).T).tobytes() for k in range(3))
    negzero = np.argwhere(np.signbit(pts) & (pts == 0)).astype(np.uint32)
    head = np.array([pts.shape[0], negzero.shape[0]], np.uint32).tobytes()
    return head + lzma.compress(planes + qi.astype(np.uint8).tobytes() + negzero.tobytes(),
                                preset=9 | lzma.PRESET_EXTREME)


def unpack_frame(blob):
    import lzma
    n, nz = (int(v) for v in np.frombuffer(blob[:8], np.uint32))
    raw = np.frombuffer(lzma.decompress(blob[8:]), np.uint8)
    zz = np.zeros((n, 3), np.uint32)
    for k in range(3):
        zz |= raw[3 * n * k:3 * n * (k + 1)].reshape(3, n).T.astype(np.uint32) << (8 * k)
    d = (zz >> 1).astype(np.int32) ^ -(zz & 1).astype(np.int32)
    q = np.cumsum(d, axis=0, dtype=np.int64)
    pts = np.empty((n, 4), np.float32)
    pts[:, :3] = (q / 1000.0).astype(np.float32)
    pts[:, 3] = (raw[9 * n:10 * n].astype(np.float64) / 100.0).astype(np.float32)
    for r, c in raw[10 * n:10 * n + 8 * nz].view(np.uint32).reshape(nz, 2):
        pts[r, c] = -0.0
    return pts


def stream_names():
    """the 154 frames of BASELINE configs[3] in filename order"""
    return sorted(f[:-3] for f in os.listdir(STREAM) if f.endswith(".xz"))


def load_stream_frame(name):
    """(n,4) float32 frame of the committed 154-frame stream, bit-identical to the reference's data/<name>.pcd"""
    with open(os.path.join(STREAM, name + ".xz"), "rb") as f:
        return unpack_frame(f.read())


def stream_gold():
    global _stream_gold
    if _stream_gold is None:
        with np.load(os.path.join(GOLDEN, "stream_golden.npz")) as z:
            _stream_gold = {k: z[k] for k in z.files}  # materialised: safe to use from forked workers
    return _stream_gold


def gold():
    global _gold
    if _gold is None:
        with np.load(os.path.join(GOLDEN, "golden.npz")) as z:
            _gold = {k: z[k] for k in z.files}  # materialised: no file handle to share with forked workers
    return _gold


def splitmix64(seed, n):
    """n uint64 values of the splitmix64 stream (SURVEY 8d generator)"""
    x = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    z = x.copy()
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def synthetic_scene(n_ground, n_boxes, pts_per_box, seed, extent=60.0):
    """Plane + box-surface cloud, coordinates rounded to 1 mm (KITTI tie structure), shuffled.
    SURVEY 8d: z = -1.73 + 0.010 x - 0.005 y + N(0, 0.02^2); boxes stand on the plane."""
    with np.errstate(over="ignore"):
        u = uniform(seed, 4 * n_ground)
        gx = (u[:n_ground] * 2 - 1) * extent
        gy = (u[n_ground:2 * n_ground] * 2 - 1) * extent
        g1 = np.sqrt(-2 * np.log(np.maximum(u[2 * n_ground:3 * n_ground], 1e-300)))
        gz = -1.73 + 0.010 * gx - 0.005 * gy + 0.02 * g1 * np.cos(2 * np.pi * u[3 * n_ground:])
        ub = uniform(seed + 1, 5 * n_boxes)
        cx = (ub[:n_boxes] * 2 - 1) * extent
        cy = (ub[n_boxes:2 * n_boxes] * 2 - 1) * extent
        sx = 0.5 + 3.5 * ub[2 * n_boxes:3 * n_boxes]
        sy = 0.5 + 3.5 * ub[3 * n_boxes:4 * n_boxes]
        h = 0.5 + 2.0 * ub[4 * n_boxes:]
        up = uniform(seed + 2, 3 * n_boxes * pts_per_box).reshape(3, n_boxes, pts_per_box)
    face = (up[0] * 5).astype(np.int64)
    a, b = up[1], up[2]
    bx = np.where(face == 0, -0.5, np.where(face == 1, 0.5, a - 0.5)) * sx[:, None] + cx[:, None]
    by = np.where(face == 2, -0.5, np.where(face == 3, 0.5, np.where(face < 2, a - 0.5, b - 0.5))) * sy[:, None] + \
        cy[:, None]
    base = -1.73 + 0.010 * cx - 0.005 * cy
    bz = np.where(face == 4, 1.0, np.where(face < 2, b, np.where(face < 4, a, 1.0))) * h[:, None] + base[:, None]
    pts = np.concatenate([np.stack([gx, gy, gz], 1), np.stack([bx.ravel(), by.ravel(), bz.ravel()], 1)])
    pts = np.rint(pts * 1000.0) / 1000.0
    perm = np.argsort(splitmix64(seed + 3, pts.shape[0]), kind="stable")
    out = np.zeros((pts.shape[0], 4), np.float32)
    out[:, :3] = pts[perm].astype(np.float32)
    return out


def partition_signature(labels):
    """label -> smallest member index, so two labelings compare up to a permutation of labels"""
    labels = np.asarray(labels)
    sig = np.full(labels.shape, -1, np.int64)
    valid = labels >= 0
    if valid.any():
        order = np.argsort(labels[valid], kind="stable")
        idx = np.nonzero(valid)[0][order]
        lab = labels[valid][order]
        first = np.r_[True, lab[1:] != lab[:-1]]
        rep = np.maximum.accumulate(np.where(first, idx, 0))
        # idx is ascending inside a label (stable sort), so the first entry is the smallest index
        sig[idx] = rep
    sig[labels == -1] = -1
    sig[labels < -1] = -2
    return sig


def brute_components(xyz, r2):
    """connected components of the d-graph with the reference's float32 dist_sqr; root = min index"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree
    p = np.asarray(xyz, np.float32)[:, :3]
    t = cKDTree(p.astype(np.float64))
    pairs = t.query_pairs(float(np.sqrt(r2)) * 1.001 + 1e-6, output_type="ndarray")
    a, b = p[pairs[:, 0]], p[pairs[:, 1]]
    d0, d1, d2 = (a[:, 0] - b[:, 0]), (a[:, 1] - b[:, 1]), (a[:, 2] - b[:, 2])
    dist = d0 * d0 + (d1 * d1 + (d2 * d2 + np.float32(0)))
    pairs = pairs[dist <= np.float32(r2)]
    m = p.shape[0]
    g = coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(m, m))
    _, lab = connected_components(g, directed=False)
    root = np.full(lab.max() + 1 if m else 0, m, np.int64)
    np.minimum.at(root, lab, np.arange(m))
    return root[lab].astype(np.uint32)
