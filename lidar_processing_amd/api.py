"""Host-side mirror of the reference's operator interface for the hot path.

`Segmenter` / `Clusterer` keep the reference's method names, argument meaning, defaults and error
behaviour (reference src/segmentation.hpp:48-70, src/clustering.hpp:42-75); point clouds are
(n, k>=3) float32 arrays whose first three columns are x, y, z (k = 4 for PointXYZ's 16-byte
record, 8 for the 32-byte PointXYZI / PointXYZRGBL records).  Everything computes on the GPU
through liblpx.so; nothing here falls back to the CPU.
"""
import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import _lib

UNDEFINED = -(2 ** 31)  # Clusterer::UNDEFINED, src/clustering.hpp:53
INVALID = -1            # Clusterer::INVALID,   src/clustering.hpp:54

_ERRORS = {-1: "LPX_ERR_ARG", -2: "LPX_ERR_RANGE", -3: "LPX_ERR_HIP", -4: "LPX_ERR_CAPACITY",
           -5: "LPX_ERR_NO_DEVICE", -6: "LPX_ERR_INTERNAL"}


class LpxError(RuntimeError):
    def __init__(self, code, message=""):
        self.code = code
        super().__init__(f"{_ERRORS.get(code, code)}: {message}")


class SegmentationLabel(enum.IntEnum):
    """src/segmentation.hpp:41-46"""
    UNKNOWN = 0
    GROUND = 1
    OBSTACLE = 2


@dataclass
class SegmentationConfiguration:
    """src/segmentation.hpp:48-56 (same defaults)"""
    sensor_height_m: float = 1.73
    orthogonal_distance_threshold: float = 0.3
    initial_seed_threshold: float = 0.6
    number_of_iterations: int = 3
    number_of_planar_partitions: int = 2
    number_of_lower_point_representatives: int = 5000

    def _c(self):
        return _lib.SegCfg(self.sensor_height_m, self.orthogonal_distance_threshold, self.initial_seed_threshold,
                           self.number_of_iterations, self.number_of_planar_partitions,
                           self.number_of_lower_point_representatives)


@dataclass
class ClusteringConfiguration:
    """src/clustering.hpp:42-48 (same defaults)"""
    distance_squared: float = 0.18
    cluster_quality: float = 0.5
    min_cluster_size: int = 4
    max_cluster_size: int = 2 ** 32 - 1

    def _c(self):
        return _lib.CluCfg(self.distance_squared, self.cluster_quality, self.min_cluster_size, self.max_cluster_size)


def _points(cloud):
    a = np.asarray(cloud)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("point cloud must be an (n, k>=3) array with x, y, z in the first three columns")
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.shape[1] * 4


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP device + stream + scratch (lpx_ctx).  Not thread-safe, like the reference objects."""

    def __init__(self, device=0, stream=None, batch=1):
        self._L = _lib.lib()
        h = C.c_void_p()
        self.batch = int(batch)
        if self.batch != 1:
            rc = self._L.lpx_create_batch(int(device), self.batch, C.byref(h))
        elif stream is None:
            rc = self._L.lpx_create(int(device), C.byref(h))
        else:
            rc = self._L.lpx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        if rc != 0:
            raise LpxError(rc, "no usable GPU: the MI355X path has no CPU fallback" if rc == -5 else "lpx_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise LpxError(rc, self._L.lpx_last_error(self._h).decode())

    def reserve(self, n_points, neighbours_per_point=0):
        self.check(self._L.lpx_reserve(self._h, int(n_points), int(neighbours_per_point)))

    def reserve_single_pass(self, words_per_point):
        """lpx_reserve_single_pass: extra neighbour workspace for lists reserved by an upper bound (0 = off)"""
        self.check(self._L.lpx_reserve_single_pass(self._h, int(words_per_point)))

    def set_neighbour_mode(self, mode):
        """lpx_set_neighbour_mode: "auto" (lists for a single-frame context, search for a batch context), "lists"
        (every radius list materialised: lowest single-frame latency) or "search" (expansion-driven: highest
        throughput, no list workspace); all give the same labels"""
        self.check(self._L.lpx_set_neighbour_mode(self._h, {"auto": 0, "lists": 1, "search": 2}[mode]))

    def use_lists(self, on=True):
        self.set_neighbour_mode("lists" if on else "search")

    def set_overlap(self, on=True):
        """lpx_set_overlap: batch contexts alternate between two slot sets and run the replay + label kernels of a
        call on a second stream (throughput; results are final after synchronize())"""
        self.check(self._L.lpx_set_overlap(self._h, 1 if on else 0))

    def set_fork(self, on=True):
        """lpx_set_fork: the component search of a call runs on a side stream beside its kd build and chunk tables"""
        self.check(self._L.lpx_set_fork(self._h, 1 if on else 0))

    def set_lookahead(self, on=True):
        """lpx_set_lookahead: segment() enqueues the clustering the cluster() call that follows it will ask for"""
        self.check(self._L.lpx_set_lookahead(self._h, 1 if on else 0))

    def set_record_copy(self, on=True):
        """lpx_set_record_copy: the segmentation keeps its own copy of the coordinates, so that coloured_clouds of a
        DEVICE call no longer read the caller's input array.  Off (the default) that array has to stay allocated and
        unmodified until the coloured-cloud call has run (include/lpx.h, LIFETIME OF THE INPUT)."""
        self.check(self._L.lpx_set_record_copy(self._h, 1 if on else 0))

    def lookahead_hits(self):
        """cluster() calls of this context that found their clustering already enqueued by segment()"""
        return int(self._L.lpx_dbg_lookahead_hits(self._h))

    def wait_previous(self):
        """lpx_wait_previous: every batch call but the last one is complete (overlapped contexts)"""
        self.check(self._L.lpx_wait_previous(self._h))

    def synchronize(self):
        self.check(self._L.lpx_synchronize(self._h))

    # ---- profiling ----
    def profile_enable(self, on=True):
        self.check(self._L.lpx_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, reset=True):
        n = self._L.lpx_profile_stage_count()
        ms = np.zeros(n, np.float32)
        cnt = np.zeros(n, np.uint32)
        self.check(self._L.lpx_profile_read(self._h, _vp(ms), _vp(cnt), 1 if reset else 0))
        names = [self._L.lpx_profile_stage_name(i).decode() for i in range(n)]
        return {names[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}

    # ---- host entry points ----
    def segment(self, cloud, cfg):
        a, stride = _points(cloud)
        n = a.shape[0]
        P = cfg.number_of_planar_partitions
        labels = np.zeros(n, np.uint32)
        gi = np.zeros(max(n, 1), np.uint32)
        oi = np.zeros(max(n, 1), np.uint32)
        planes = np.zeros((max(P, 1), 4), np.float32)
        ng, no = C.c_uint32(0), C.c_uint32(0)
        c = cfg._c()
        self.check(self._L.lpx_segment(self._h, _vp(a), stride, n, C.byref(c), _vp(labels), _vp(gi), C.byref(ng),
                                       _vp(oi), C.byref(no), _vp(planes)))
        return labels, gi[:ng.value].copy(), oi[:no.value].copy(), planes[:P]

    def cluster(self, cloud, cfg):
        a, stride = _points(cloud)
        m = a.shape[0]
        labels = np.full(m, UNDEFINED, np.int32)
        nc = C.c_uint32(0)
        c = cfg._c()
        self.check(self._L.lpx_cluster(self._h, _vp(a), stride, m, C.byref(c), _vp(labels), C.byref(nc)))
        return labels, nc.value

    def segment_cluster(self, cloud, seg_cfg, clu_cfg):
        a, stride = _points(cloud)
        n = a.shape[0]
        P = seg_cfg.number_of_planar_partitions
        labels = np.zeros(n, np.uint32)
        gi = np.zeros(max(n, 1), np.uint32)
        oi = np.zeros(max(n, 1), np.uint32)
        cl = np.full(max(n, 1), UNDEFINED, np.int32)
        planes = np.zeros((max(P, 1), 4), np.float32)
        ng, no, nc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        sc, cc = seg_cfg._c(), clu_cfg._c()
        self.check(self._L.lpx_segment_cluster(self._h, _vp(a), stride, n, C.byref(sc), C.byref(cc), _vp(labels),
                                               _vp(gi), C.byref(ng), _vp(oi), C.byref(no), _vp(planes), _vp(cl),
                                               C.byref(nc)))
        return dict(labels=labels, ground_idx=gi[:ng.value].copy(), obstacle_idx=oi[:no.value].copy(),
                    planes=planes[:P], cluster_labels=cl[:no.value].copy(), n_clusters=nc.value)

    # ---- PointCloud2 wire format ----
    def segment_cluster_fields(self, data, point_step, offsets, n, seg_cfg, clu_cfg=None):
        """lpx_segment_fields / lpx_segment_cluster_fields: `data` is the byte buffer of a sensor_msgs/PointCloud2
        message (n records of point_step bytes), offsets = byte offsets of its float32 x, y, z fields"""
        buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray)
                                   else data.view(np.uint8).reshape(-1))
        assert buf.size >= n * point_step
        P = seg_cfg.number_of_planar_partitions
        labels = np.zeros(n, np.uint32)
        gi = np.zeros(max(n, 1), np.uint32)
        oi = np.zeros(max(n, 1), np.uint32)
        cl = np.full(max(n, 1), UNDEFINED, np.int32)
        planes = np.zeros((max(P, 1), 4), np.float32)
        ng, no, nc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        sc = seg_cfg._c()
        ox, oy, oz = (int(o) for o in offsets)
        if clu_cfg is None:
            self.check(self._L.lpx_segment_fields(self._h, _vp(buf), int(point_step), ox, oy, oz, int(n), C.byref(sc),
                                                  _vp(labels), _vp(gi), C.byref(ng), _vp(oi), C.byref(no),
                                                  _vp(planes)))
        else:
            cc = clu_cfg._c()
            self.check(self._L.lpx_segment_cluster_fields(self._h, _vp(buf), int(point_step), ox, oy, oz, int(n),
                                                          C.byref(sc), C.byref(cc), _vp(labels), _vp(gi), C.byref(ng),
                                                          _vp(oi), C.byref(no), _vp(planes), _vp(cl), C.byref(nc)))
        return dict(labels=labels, ground_idx=gi[:ng.value].copy(), obstacle_idx=oi[:no.value].copy(),
                    planes=planes[:P], cluster_labels=cl[:no.value].copy(), n_clusters=nc.value)

    def coloured_clouds(self, n_ground, n_obstacle):
        """lpx_coloured_clouds: the 32-byte PointXYZRGBL records of the ground and obstacle clouds of the last host
        segmentation call (reference src/processor.cpp:152-163), as two uint8 arrays of shape (count, 32)"""
        g = np.zeros((max(n_ground, 1), 32), np.uint8)
        o = np.zeros((max(n_obstacle, 1), 32), np.uint8)
        ng, no = C.c_uint32(0), C.c_uint32(0)
        self.check(self._L.lpx_coloured_clouds(self._h, _vp(g), _vp(o), C.byref(ng), C.byref(no)))
        assert ng.value == n_ground and no.value == n_obstacle
        return g[:n_ground], o[:n_obstacle]

    def cluster_groups(self, m, n_clusters):
        """offsets[n_clusters + 1], indices: the regrouping of reference src/processor.cpp:180-200 for the
        labels of the last cluster call of this context"""
        off = np.zeros(n_clusters + 1, np.uint32)
        idx = np.zeros(max(m, 1), np.uint32)
        nv = C.c_uint32(0)
        self._L.lpx_cluster_groups.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_uint32)]
        self.check(self._L.lpx_cluster_groups(self._h, int(m), int(n_clusters), _vp(off), _vp(idx), C.byref(nv)))
        return off, idx[:nv.value].copy()

    def cluster_hulls(self, m, n_clusters, max_points=20):
        """lpx_cluster_hulls: (hull_offsets[n_clusters + 1], hull_indices, hull_xy) -- counter-clockwise convex
        hulls of the clusters with fewer than max_points points (reference src/polygon_simplification.cpp:96-115)
        for the labels of the last cluster call of this context"""
        off = np.zeros(n_clusters + 1, np.uint32)
        idx = np.zeros(max(m, 1), np.uint32)
        xy = np.zeros((max(m, 1), 2), np.float32)
        nh = C.c_uint32(0)
        self.check(self._L.lpx_cluster_hulls(self._h, int(m), int(n_clusters), int(max_points), _vp(off), _vp(idx),
                                             _vp(xy), C.byref(nh)))
        return off, idx[:nh.value].copy(), xy[:nh.value].copy()

    # ---- device-resident entry point (asynchronous on the context stream) ----
    def segment_cluster_device(self, d_pts, stride_bytes, n, seg_cfg, clu_cfg, d_labels, d_ground_idx, d_obstacle_idx,
                               d_planes, d_cluster_labels, d_counts):
        """lpx_segment_cluster_device: every d_* is a raw device pointer (int); nothing synchronises."""
        sc, cc = seg_cfg._c(), clu_cfg._c()
        self.check(self._L.lpx_segment_cluster_device(self._h, d_pts, stride_bytes, n, C.byref(sc), C.byref(cc),
                                                      d_labels, d_ground_idx, d_obstacle_idx, d_planes,
                                                      d_cluster_labels, d_counts))

    def segment_cluster_batch_device(self, n_points, d_pts, stride_bytes, frame_pitch, seg_cfg, clu_cfg, d_labels,
                                     d_ground_idx, d_obstacle_idx, d_planes, d_cluster_labels, d_counts):
        """lpx_segment_cluster_batch_device: len(n_points) frames per launch chain; frame b of every array
        starts b * frame_pitch elements (records for d_pts) behind frame 0; nothing synchronises."""
        sc, cc = seg_cfg._c(), clu_cfg._c()
        n = np.ascontiguousarray(n_points, dtype=np.uint32)
        self.check(self._L.lpx_segment_cluster_batch_device(self._h, n.shape[0], d_pts, stride_bytes, frame_pitch,
                                                            _vp(n), C.byref(sc), C.byref(cc), d_labels, d_ground_idx,
                                                            d_obstacle_idx, d_planes, d_cluster_labels, d_counts))

    def workspace_bytes(self):
        """lpx_workspace_bytes: (frame-slot arenas, neighbour-list arena) in bytes"""
        o = np.zeros(2, np.uint64)
        self._L.lpx_workspace_bytes.argtypes = [C.c_void_p, C.c_void_p]
        self.check(self._L.lpx_workspace_bytes(self._h, _vp(o)))
        return int(o[0]), int(o[1])

    def frame_stats(self, slot=0):
        """counters of the last frame processed in frame slot `slot` of this context (synchronises)"""
        o = np.zeros(12, np.uint32)
        self._L.lpx_dbg_frame_stats_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        self.check(self._L.lpx_dbg_frame_stats_slot(self._h, slot, _vp(o)))
        s4 = np.zeros(4, np.uint32)
        self._L.lpx_dbg_search_stats_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        self.check(self._L.lpx_dbg_search_stats_slot(self._h, slot, _vp(s4)))
        return dict(n_ground=int(o[0]), n_obstacle=int(o[1]), n_clusters=int(o[2]), status=int(o[3]),
                    neighbour_entries=int(o[4]) | (int(o[5]) << 32), components=int(o[6]), expansions=int(o[7]),
                    replay_entries=int(o[8]) | (int(o[9]) << 32), neighbour_words=int(o[10]) | (int(o[11]) << 32),
                    candidates=int(s4[0]) | (int(s4[1]) << 32), windows=int(s4[2]), overflows=int(s4[3]))

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        """lpx_dbg_copy_bandwidth: GB/s (read + write) of a streaming device copy -- the achievable HBM bandwidth"""
        g = C.c_double(0.0)
        self._L.lpx_dbg_copy_bandwidth.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_double)]
        self.check(self._L.lpx_dbg_copy_bandwidth(self._h, int(nbytes), int(reps), C.byref(g)))
        return g.value

    # ---- stage-level entry points (parity tests) ----
    def dbg_sort_pairs(self, keys, values, bits=32):
        k = np.array(keys, dtype=np.uint32)
        v = np.array(values, dtype=np.uint32)
        self.check(self._L.lpx_dbg_sort_pairs(self._h, _vp(k), _vp(v), k.shape[0], bits))
        return k, v

    def dbg_sort_keys64(self, keys, bits=64):
        k = np.array(keys, dtype=np.uint64)
        self.check(self._L.lpx_dbg_sort_keys64(self._h, _vp(k), k.shape[0], bits))
        return k

    def dbg_scan(self, data):
        d = np.array(data, dtype=np.uint32)
        tot = C.c_uint64(0)
        self.check(self._L.lpx_dbg_scan(self._h, _vp(d), d.shape[0], C.byref(tot)))
        return d, tot.value

    def dbg_kd_layout(self, xyz):
        a = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32)[:, :3])
        out = np.zeros(a.shape[0], np.uint32)
        self.check(self._L.lpx_dbg_kd_layout(self._h, _vp(a), a.shape[0], _vp(out)))
        return out

    def dbg_neighbours(self, xyz, r2, capacity=None):
        a = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32)[:, :3])
        m = a.shape[0]
        cap = int(capacity if capacity is not None else max(1024, 512 * m))
        off = np.zeros(m + 1, np.uint64)
        idx = np.zeros(cap, np.uint32)
        dist = np.zeros(cap, np.float32)
        rc = self._L.lpx_dbg_neighbours(self._h, _vp(a), m, C.c_float(r2), _vp(off), _vp(idx), _vp(dist), cap)
        if rc == -4 and capacity is None:
            return self.dbg_neighbours(xyz, r2, capacity=int(off[m]) + 16)
        self.check(rc)
        tot = int(off[m])
        return off, idx[:tot].copy(), dist[:tot].copy()

    def dbg_components(self, xyz, r2):
        a = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32)[:, :3])
        root = np.zeros(a.shape[0], np.uint32)
        self.check(self._L.lpx_dbg_components(self._h, _vp(a), a.shape[0], C.c_float(r2), _vp(root)))
        return root

    def dbg_plane(self, xyz):
        a = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32)[:, :3])
        plane = np.zeros(4, np.float32)
        rc = self._L.lpx_dbg_plane(self._h, _vp(a), a.shape[0], _vp(plane))
        if rc < 0:
            self.check(rc)
        return plane, rc


class PinnedArray:
    """numpy view of page-locked host memory (lpx_host_alloc); free() or garbage collection releases it"""

    def __init__(self, shape, dtype):
        self._L = _lib.lib()
        self.dtype = np.dtype(dtype)
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        rc = self._L.lpx_host_alloc(C.byref(p), max(nbytes, 1))
        if rc != 0:
            raise LpxError(rc, "lpx_host_alloc")
        self._p = p
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.lpx_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PlainArray:
    """ordinary (pageable) host memory with the interface of PinnedArray"""

    def __init__(self, shape, dtype):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.array = np.zeros(self.shape, self.dtype)

    def free(self):
        self.array = None


def pcd_info(path):
    """lpx_pcd_info_read: dict(n_points, point_step, offsets=(x, y, z), n_fields) of a binary PCD v0.7 file"""
    inf = _lib.PcdInfo()
    rc = _lib.lib().lpx_pcd_info_read(str(path).encode(), C.byref(inf))
    if rc != 0:
        raise LpxError(rc, f"{path}: not a binary PCD file with float32 x, y, z")
    return dict(n_points=inf.n_points, point_step=inf.point_step, offsets=(inf.off_x, inf.off_y, inf.off_z),
                n_fields=inf.n_fields)


def load_pcd(path, pinned=True):
    """lpx_pcd_load: the records of the file as an (n, point_step / 4) float32 array (a view of pinned memory when
    pinned=True, kept alive by the array's .base chain) and the info dict.  Counterpart of pcl::io::loadPCDFile as
    the reference uses it (src/dataloader.cpp:139)."""
    info = pcd_info(path)
    n, step = info["n_points"], info["point_step"]
    if step % 4:
        raise LpxError(-1, "records are not a whole number of floats")
    if pinned:
        holder = PinnedArray((n, step // 4), np.float32)
        dst = holder.array
    else:
        holder, dst = None, np.zeros((n, step // 4), np.float32)
    inf = _lib.PcdInfo()
    rc = _lib.lib().lpx_pcd_load(str(path).encode(), _vp(dst) if n else None, dst.nbytes, C.byref(inf))
    if rc != 0:
        raise LpxError(rc, f"{path}: load failed")
    info["_pinned"] = holder
    return dst, info


class Feeder:
    """lpx_feeder: a list of PCD files preloaded into pinned memory (the reference preloads all clouds too,
    src/dataloader.cpp:128-153) and the double-buffered H2D -> launch chain -> D2H pipeline over them"""

    def __init__(self, paths, device=0):
        self._L = _lib.lib()
        arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
        h = C.c_void_p()
        rc = self._L.lpx_feeder_create(int(device), arr, len(paths), C.byref(h))
        if rc != 0:
            raise LpxError(rc, "lpx_feeder_create")
        self._h = h
        self.n_frames = self._L.lpx_feeder_frames(h)
        self.info = []
        for i in range(self.n_frames):
            inf = _lib.PcdInfo()
            self._L.lpx_feeder_frame(h, i, C.byref(inf))
            self.info.append(dict(n_points=inf.n_points, point_step=inf.point_step,
                                  offsets=(inf.off_x, inf.off_y, inf.off_z)))

    def frame(self, i):
        """(n, point_step / 4) float32 view of the pinned records of frame i"""
        inf = _lib.PcdInfo()
        p = self._L.lpx_feeder_frame(self._h, i, C.byref(inf))
        buf = (C.c_char * max(inf.n_points * inf.point_step, 1)).from_address(p)
        return np.frombuffer(buf, dtype=np.float32, count=inf.n_points * inf.point_step // 4).reshape(
            inf.n_points, inf.point_step // 4)

    def run(self, ctx, frame_ids, seg_cfg, clu_cfg, out=None, pinned=True):
        """lpx_feeder_run through the batch context `ctx` -- or lpx_feeder_run_multi through a list of batch
        contexts of equal slot count (chain k on context k % len(ctx)); returns the dict of pinned, pitched result
        arrays (labels, ground_idx, obstacle_idx, cluster_labels (F, pitch); planes (F, 4P); counts (F, 4))"""
        ids = np.ascontiguousarray(frame_ids, dtype=np.uint32)
        F = ids.shape[0]
        P = seg_cfg.number_of_planar_partitions
        if F and int(ids.max()) >= self.n_frames:
            raise LpxError(-1, f"frame id {int(ids.max())} out of range: the feeder holds {self.n_frames} frames")
        if out is None:
            pitch = max([self.info[i]["n_points"] for i in ids.tolist()] + [1])
            # pinned arrays let the copy engines write the results while the next chain computes; ordinary memory
            # (pinned=False) works too, at the speed of a staged copy
            arr = PinnedArray if pinned else _PlainArray
            out = dict(pitch=pitch,
                       labels=arr((F, pitch), np.uint32), ground_idx=arr((F, pitch), np.uint32),
                       obstacle_idx=arr((F, pitch), np.uint32), cluster_labels=arr((F, pitch), np.int32),
                       planes=arr((F, 4 * P), np.float32), counts=arr((F, 4), np.uint32))
        so = _lib.StreamOut(*[out[k].array.ctypes.data for k in ("labels", "ground_idx", "obstacle_idx", "cluster_labels",
                                                                 "planes", "counts")], out["pitch"])
        sc, cc = seg_cfg._c(), clu_cfg._c()
        if isinstance(ctx, (list, tuple)):
            hs = (C.c_void_p * len(ctx))(*[c._h for c in ctx])
            rc = self._L.lpx_feeder_run_multi(self._h, hs, len(ctx), _vp(ids), F, C.byref(sc), C.byref(cc), C.byref(so))
        else:
            rc = self._L.lpx_feeder_run(self._h, ctx._h, _vp(ids), F, C.byref(sc), C.byref(cc), C.byref(so))
        if rc != 0:
            raise LpxError(rc, self._L.lpx_feeder_last_error(self._h).decode())
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpx_feeder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def _ctx(device):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Segmenter:
    """Mirror of lidar_processing::Segmenter (src/segmentation.hpp:58-70)."""

    def __init__(self, device=0, context=None):
        self._ctx = context or _ctx(device)
        self.configuration = SegmentationConfiguration()
        self.reserve_memory()

    def update_configuration(self, configuration):
        self.configuration = configuration
        self.reserve_memory()

    def reserve_memory(self, number_of_points=200_000):
        self._ctx.reserve(number_of_points)

    def segment(self, cloud_in):
        """Returns (labels, ground_cloud, obstacle_cloud) -- the three outputs of Segmenter::segment
        (src/segmentation.cpp:311-345); the clouds keep every column of cloud_in."""
        a = np.asarray(cloud_in)
        labels, gi, oi, _ = self._ctx.segment(a, self.configuration)
        return labels, a[gi], a[oi]

    def segment_indices(self, cloud_in):
        """labels, ground indices, obstacle indices (output-cloud order) and the fitted planes"""
        return self._ctx.segment(cloud_in, self.configuration)


class Clusterer:
    """Mirror of lidar_processing::Clusterer (src/clustering.hpp:50-75)."""
    UNDEFINED = UNDEFINED
    INVALID = INVALID

    def __init__(self, device=0, context=None):
        self._ctx = context or _ctx(device)
        self.configuration = ClusteringConfiguration()
        self.reserve_memory()

    def update_configuration(self, configuration):
        self.configuration = configuration

    def reserve_memory(self, number_of_points=200_000):
        self._ctx.reserve(number_of_points)

    def cluster(self, cloud_in):
        """labels (int32, one per point) as Clusterer::cluster writes them (src/clustering.cpp:47-125)"""
        a = np.asarray(cloud_in)
        if a.shape[0] == 0:
            return np.zeros(0, np.int32)
        labels, self._n_clusters = self._ctx.cluster(a, self.configuration)
        self._m = a.shape[0]
        return labels

    def convex_outlines(self, cloud_in, max_points=20):
        """findOrderedConcaveOutlines' convex branch (src/polygon_simplification.cpp:96-115): list of (k, 2) xy
        arrays, one per cluster with fewer than max_points points (counter-clockwise), clusters in label order;
        larger clusters are skipped (concave branch: out of scope)"""
        off, idx, xy = self._ctx.cluster_hulls(self._m, self._n_clusters, max_points)
        return [xy[off[c]:off[c + 1]] for c in range(self._n_clusters) if off[c + 1] > off[c]]

    def grouped(self, cloud_in):
        """The regrouping the reference's caller does right after cluster() (src/processor.cpp:180-200):
        list of (k, 3) xyz arrays, one per valid cluster in label order, points in index order."""
        a = np.asarray(cloud_in)
        off, idx = self._ctx.cluster_groups(self._m, self._n_clusters)
        return [np.ascontiguousarray(a[idx[off[c]:off[c + 1]], :3]) for c in range(self._n_clusters)]
