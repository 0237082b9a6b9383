"""ctypes bindings of the parity oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (lidar_processing_amd) never does.

  liboracle.so       CPU restatement of the reference hot path (oracle/lidar_oracle.c)
  _ref/libkdref.so   the reference's own kdtree.hpp/queue.hpp compiled unmodified (ref_driver.cpp)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

UNKNOWN, GROUND, OBSTACLE = 0, 1, 2
UNDEFINED = -(2**31)
INVALID = -1
SEG_OK, SEG_TOO_FEW_POINTS, SEG_ALL_OBSTACLE = 0, 1, 2
ERR_RANGE = -2


class SegCfg(C.Structure):
    """mirrors SegmentationConfiguration (reference src/segmentation.hpp:48-56)"""

    _fields_ = [
        ("sensor_height_m", C.c_float),
        ("orthogonal_distance_threshold", C.c_float),
        ("initial_seed_threshold", C.c_float),
        ("number_of_iterations", C.c_uint32),
        ("number_of_planar_partitions", C.c_uint32),
        ("number_of_lower_point_representatives", C.c_uint32),
    ]

    def __init__(self, sensor_height_m=1.73, orthogonal_distance_threshold=0.3, initial_seed_threshold=0.6,
                 number_of_iterations=3, number_of_planar_partitions=2, number_of_lower_point_representatives=5000):
        super().__init__(sensor_height_m, orthogonal_distance_threshold, initial_seed_threshold,
                         number_of_iterations, number_of_planar_partitions, number_of_lower_point_representatives)


class CluCfg(C.Structure):
    """mirrors ClusteringConfiguration (reference src/clustering.hpp:42-48)"""

    _fields_ = [
        ("distance_squared", C.c_float),
        ("cluster_quality", C.c_float),
        ("min_cluster_size", C.c_uint32),
        ("max_cluster_size", C.c_uint32),
    ]

    def __init__(self, distance_squared=0.18, cluster_quality=0.5, min_cluster_size=4, max_cluster_size=2**32 - 1):
        super().__init__(distance_squared, cluster_quality, min_cluster_size, max_cluster_size)


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/libkdref.so."""
    if force or not os.path.exists(os.path.join(_HERE, "liboracle.so")) or (
            os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(_HERE, "_ref", "libkdref.so"))):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        _lib.orc_segment.restype = C.c_int
        _lib.orc_cluster_stats.restype = C.c_int
    return _lib


def ref():
    """The compiled reference kd-tree driver; None if it has not been built (no /root/reference)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libkdref.so")
        if not os.path.exists(path):
            build()
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_sort = None


def x_sort_order(x, mode):
    """index order of the reference's x sort (src/segmentation.cpp:114-122) for equal x, as this image's libstdc++
    produces it: mode "stable" (the canonical (x, index) order of this repository), "serial" (std::sort: a build
    without TBB headers) or "tbb" (leaves of <= 500 by std::sort, stable merges: oracle/sort_order.cpp)"""
    global _sort
    if _sort is None:
        path = os.path.join(_HERE, "libsortorder.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, "-s", "libsortorder.so"], check=True)
        _sort = C.CDLL(path)
    xs = np.ascontiguousarray(x, dtype=np.float32)
    idx = np.zeros(xs.shape[0], np.uint32)
    _sort.so_order(_p(xs), C.c_uint32(xs.shape[0]), _p(idx), C.c_int({"stable": 0, "serial": 1, "tbb": 2}[mode]))
    return idx


def _as_points(points):
    """(n,k>=3) float32 C-contiguous array -> (array, stride_bytes)"""
    a = np.ascontiguousarray(points, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a, a.shape[1] * 4


def segment(points, cfg=None):
    """Segmenter::segment restated.  Returns dict(labels, ground_idx, obstacle_idx, planes, status, rc)."""
    cfg = cfg or SegCfg()
    a, stride = _as_points(points)
    n = a.shape[0]
    P = cfg.number_of_planar_partitions
    labels = np.zeros(n, np.uint32)
    gi = np.zeros(max(n, 1), np.uint32)
    oi = np.zeros(max(n, 1), np.uint32)
    ng = C.c_uint32(0)
    no = C.c_uint32(0)
    planes = np.zeros((max(P, 1), 4), np.float32)
    status = np.zeros(max(P, 1), np.uint32)
    rc = lib().orc_segment(_p(a), C.c_size_t(stride), C.c_uint32(n), C.byref(cfg), _p(labels), _p(gi), C.byref(ng),
                           _p(oi), C.byref(no), _p(planes), _p(status))
    return dict(labels=labels, ground_idx=gi[:ng.value].copy(), obstacle_idx=oi[:no.value].copy(),
                planes=planes[:P], status=status[:P], rc=rc)


def set_sort_threads(n):
    """threads of the two index sorts in segment() (the reference's std::sort(std::execution::par)); default 1"""
    lib().orc_set_sort_threads(C.c_int(int(n)))


def cluster(points, cfg=None, stats=False):
    """Clusterer::cluster restated.  Returns (labels int32, n_clusters[, expansions, visits])."""
    cfg = cfg or CluCfg()
    a, stride = _as_points(points)
    m = a.shape[0]
    labels = np.zeros(m, np.int32)
    nc = C.c_uint32(0)
    ne = C.c_uint64(0)
    nv = C.c_uint64(0)
    rc = lib().orc_cluster_stats(_p(a), C.c_size_t(stride), C.c_uint32(m), C.byref(cfg), _p(labels), C.byref(nc),
                                 C.byref(ne), C.byref(nv))
    assert rc == 0
    if stats:
        return labels, nc.value, ne.value, nv.value
    return labels, nc.value


def _xyz(points):
    return np.ascontiguousarray(np.asarray(points, dtype=np.float32)[:, :3])


def kd_preorder(points):
    xyz = _xyz(points)
    out = np.zeros(xyz.shape[0], np.uint32)
    assert lib().orc_kd_preorder(_p(xyz), C.c_uint32(xyz.shape[0]), _p(out)) == 0
    return out


def kd_layout(points):
    xyz = _xyz(points)
    out = np.zeros(xyz.shape[0], np.uint32)
    assert lib().orc_kd_layout(_p(xyz), C.c_uint32(xyz.shape[0]), _p(out)) == 0
    return out


def radius_search(points, target, r2):
    xyz = _xyz(points)
    m = xyz.shape[0]
    idx = np.zeros(max(m, 1), np.uint32)
    dist = np.zeros(max(m, 1), np.float32)
    cnt = C.c_uint32(0)
    t = np.ascontiguousarray(target, dtype=np.float32)
    lib().orc_radius_search(_p(xyz), C.c_uint32(m), _p(t), C.c_float(r2), _p(idx), _p(dist), C.byref(cnt))
    return idx[:cnt.value].copy(), dist[:cnt.value].copy()


def nth_element(keys, payload, first, nth, last):
    k = np.array(keys, dtype=np.float32)
    p = np.array(payload, dtype=np.uint32)
    lib().orc_nth_element_u32(_p(k), _p(p), C.c_uint32(first), C.c_uint32(nth), C.c_uint32(last))
    return k, p


def plane_from_points(xyz):
    a = _xyz(xyz)
    plane = np.zeros(4, np.float32)
    rc = lib().orc_plane_from_points(_p(a), C.c_uint32(a.shape[0]), _p(plane))
    return plane, rc


def jacobi_svd3(mat):
    a = np.ascontiguousarray(mat, dtype=np.float32).reshape(9)
    v = np.zeros(9, np.float32)
    s = np.zeros(3, np.float32)
    lib().orc_jacobi_svd3(_p(a), _p(v), _p(s))
    return v.reshape(3, 3), s


class FitTrace:
    """records every plane fit the oracle makes inside the `with` block: .records is an (n, 16) float32 array of
    cov[9], plane[4], Jacobi sweeps, point count, failed (test instrumentation, tests/golden/make_jacobi_real.py)"""

    def __init__(self, cap=65536):
        self.buf = np.zeros((cap, 16), np.float32)
        self.records = self.buf[:0]

    def __enter__(self):
        lib().orc_trace_fits(_p(self.buf), C.c_uint32(self.buf.shape[0]))
        return self

    def __exit__(self, *exc):
        lib().orc_trace_count.restype = C.c_uint32
        n = lib().orc_trace_count()
        lib().orc_trace_fits(None, C.c_uint32(0))
        self.records = self.buf[:n].copy()
        return False


def convex_hull(xy):
    """Andrew monotone chain, CCW, collinear points excluded: indices into xy (restated published algorithm;
    reference: geom::constructConvexHull of the absent Convex-Hull submodule, src/polygon_simplification.cpp:109-110)"""
    a = np.ascontiguousarray(np.asarray(xy, dtype=np.float32)[:, :2])
    out = np.zeros(max(a.shape[0], 1), np.uint32)
    cnt = C.c_uint32(0)
    assert lib().orc_convex_hull(_p(a), C.c_uint32(a.shape[0]), _p(out), C.byref(cnt)) == 0
    return out[:cnt.value].copy()


def cluster_hulls(points, labels, n_clusters, max_points=20):
    """(hull_offsets[n_clusters + 1], hull_indices) of the clusters with fewer than max_points points"""
    a, stride = _as_points(points)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    off = np.zeros(n_clusters + 1, np.uint32)
    idx = np.zeros(max(a.shape[0], 1), np.uint32)
    assert lib().orc_cluster_hulls(_p(a), C.c_size_t(stride), C.c_uint32(a.shape[0]), _p(lab), C.c_uint32(n_clusters),
                                   C.c_uint32(max_points), _p(off), _p(idx)) == 0
    return off, idx[:int(off[n_clusters])].copy()


def coloured_records(points, idx, ground):
    """The recolour copy of reference src/processor.cpp:152-163 restated: cloud_in[idx] as 32-byte pcl::PointXYZRGBL
    records -- (x, y, z, 220, 220, 220, label 0) for the ground cloud, (x, y, z, 0, 255, 0, label 1) for the
    obstacle cloud -- i.e. the bytes convertPCLToPointCloud2 memcpy's into the message (src/conversions.cpp:164-193).
    Layout per PCL 1.12 point_types (PCL is absent from this image; third-party, version unpinned by the
    reference): float x, y, z, data[3] = 1.0f | uint8 b, g, r, a = 255 | uint32 label | 8 bytes padding (zero)."""
    a = np.ascontiguousarray(points, dtype=np.float32)
    idx = np.asarray(idx, dtype=np.int64)
    rec = np.zeros(idx.shape[0], dtype=np.dtype([("xyz", "<f4", 3), ("w", "<f4"), ("bgra", "u1", 4), ("label", "<u4"),
                                                 ("pad", "u1", 8)]))
    rec["xyz"] = a[idx, :3]
    rec["w"] = 1.0
    rec["bgra"] = (220, 220, 220, 255) if ground else (0, 255, 0, 255)
    rec["label"] = 0 if ground else 1
    return rec.view(np.uint8).reshape(idx.shape[0], 32)


# ---- compiled reference (kdtree.hpp / queue.hpp) -------------------------------------------------

def ref_kd_preorder(points):
    xyz = _xyz(points)
    out = np.zeros(xyz.shape[0], np.uint32)
    assert ref().ref_kd_preorder(_p(xyz), C.c_uint32(xyz.shape[0]), _p(out)) == 0
    return out


def ref_radius_search(points, target, r2):
    xyz = _xyz(points)
    m = xyz.shape[0]
    idx = np.zeros(max(m, 1), np.uint32)
    dist = np.zeros(max(m, 1), np.float32)
    cnt = C.c_uint32(0)
    t = np.ascontiguousarray(target, dtype=np.float32)
    ref().ref_radius_search(_p(xyz), C.c_uint32(m), _p(t), C.c_float(r2), _p(idx), _p(dist), C.byref(cnt))
    return idx[:cnt.value].copy(), dist[:cnt.value].copy()


def ref_nth_element(keys, payload, first, nth, last):
    k = np.array(keys, dtype=np.float32)
    p = np.array(payload, dtype=np.uint32)
    ref().ref_nth_element_u32(_p(k), _p(p), C.c_uint32(first), C.c_uint32(nth), C.c_uint32(last))
    return k, p


def ref_fec(points, cfg=None):
    cfg = cfg or CluCfg()
    xyz = _xyz(points)
    m = xyz.shape[0]
    labels = np.zeros(m, np.int32)
    nc = C.c_uint32(0)
    rc = ref().ref_fec(_p(xyz), C.c_uint32(m), C.c_float(cfg.distance_squared), C.c_float(cfg.cluster_quality),
                       C.c_uint32(cfg.min_cluster_size), C.c_uint32(cfg.max_cluster_size), _p(labels), C.byref(nc))
    assert rc == 0
    return labels, nc.value
