"""ctypes loader of liblpx.so (the C-ABI of include/lpx.h)."""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# LPX_LIB: development switch -- another build of the same library: liblpx_dev.so (DEV_LIB_PATH: the same sources with
# -DLPX_DEV_KNOBS, the only build that reads LPX_* environment knobs; tests that force a code path or poison the
# workspace run their subprocesses on it) or an A/B variant of tools/build_variant.sh
DEV_LIB_PATH = os.path.join(_HERE, "liblpx_dev.so")
LIB_PATH = os.path.abspath(os.environ["LPX_LIB"]) if os.environ.get("LPX_LIB") else os.path.join(_HERE, "liblpx.so")


class SegCfg(C.Structure):
    _fields_ = [("sensor_height_m", C.c_float), ("orthogonal_distance_threshold", C.c_float),
                ("initial_seed_threshold", C.c_float), ("number_of_iterations", C.c_uint32),
                ("number_of_planar_partitions", C.c_uint32), ("number_of_lower_point_representatives", C.c_uint32)]


class PcdInfo(C.Structure):
    _fields_ = [("n_points", C.c_uint32), ("point_step", C.c_uint32), ("off_x", C.c_uint32), ("off_y", C.c_uint32),
                ("off_z", C.c_uint32), ("n_fields", C.c_uint32)]


class StreamOut(C.Structure):
    _fields_ = [("labels", C.c_void_p), ("ground_idx", C.c_void_p), ("obstacle_idx", C.c_void_p),
                ("cluster_labels", C.c_void_p), ("planes", C.c_void_p), ("counts", C.c_void_p),
                ("frame_pitch", C.c_uint32)]


class CluCfg(C.Structure):
    _fields_ = [("distance_squared", C.c_float), ("cluster_quality", C.c_float), ("min_cluster_size", C.c_uint32),
                ("max_cluster_size", C.c_uint32)]


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "-s"], check=True)
    return LIB_PATH


_lib = None


def lib():
    """Load liblpx.so.  Raises if it is missing: the product path never falls back to the CPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C lidar_processing_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    # torch ships its own HIP runtime under the same soname; when torch is in the process it has to be
    # loaded first so that both use one runtime.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u32, sz, i32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int32
    pu32, pi32, pf = C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.lpx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.lpx_create_on_stream.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.lpx_destroy.argtypes = [vp]
    L.lpx_destroy.restype = None
    L.lpx_reserve.argtypes = [vp, u32, u32]
    L.lpx_reserve_single_pass.argtypes = [vp, u32]
    L.lpx_last_error.argtypes = [vp]
    L.lpx_last_error.restype = C.c_char_p
    L.lpx_synchronize.argtypes = [vp]
    L.lpx_set_overlap.argtypes = [vp, C.c_int]
    if hasattr(L, "lpx_set_fork"):  # (absent from A/B variants built from an older tree)
        L.lpx_set_fork.argtypes = [vp, C.c_int]
    if hasattr(L, "lpx_set_lookahead"):
        L.lpx_set_lookahead.argtypes = [vp, C.c_int]
        L.lpx_dbg_lookahead_hits.argtypes = [vp]
        L.lpx_dbg_lookahead_hits.restype = C.c_uint64
    if hasattr(L, "lpx_set_record_copy"):
        L.lpx_set_record_copy.argtypes = [vp, C.c_int]
    L.lpx_wait_previous.argtypes = [vp]
    L.lpx_segment.argtypes = [vp, vp, sz, u32, C.POINTER(SegCfg), vp, vp, pu32, vp, pu32, vp]
    L.lpx_cluster.argtypes = [vp, vp, sz, u32, C.POINTER(CluCfg), vp, pu32]
    L.lpx_segment_cluster.argtypes = [vp, vp, sz, u32, C.POINTER(SegCfg), C.POINTER(CluCfg), vp, vp, pu32, vp, pu32,
                                      vp, vp, pu32]
    L.lpx_segment_cluster_device.argtypes = [vp, vp, sz, u32, C.POINTER(SegCfg), C.POINTER(CluCfg), vp, vp, vp, vp,
                                             vp, vp]
    L.lpx_segment_fields.argtypes = [vp, vp, u32, u32, u32, u32, u32, C.POINTER(SegCfg), vp, vp, pu32, vp, pu32, vp]
    L.lpx_segment_cluster_fields.argtypes = [vp, vp, u32, u32, u32, u32, u32, C.POINTER(SegCfg), C.POINTER(CluCfg), vp, vp,
                                             pu32, vp, pu32, vp, vp, pu32]
    L.lpx_segment_cluster_fields_device.argtypes = [vp, vp, u32, u32, u32, u32, u32, C.POINTER(SegCfg), C.POINTER(CluCfg),
                                                    vp, vp, vp, vp, vp, vp]
    L.lpx_coloured_clouds.argtypes = [vp, vp, vp, pu32, pu32]
    L.lpx_coloured_clouds_device.argtypes = [vp, vp, vp, vp, vp]
    L.lpx_coloured_clouds_batch_device.argtypes = [vp, u32, u32, vp, vp, vp, vp]
    L.lpx_cluster_hulls.argtypes = [vp, u32, u32, u32, vp, vp, vp, pu32]
    L.lpx_cluster_hulls_device.argtypes = [vp, vp, u32, vp, vp, u32, vp, vp, vp]
    L.lpx_cluster_groups_device.argtypes = [vp, vp, u32, vp, vp]
    L.lpx_host_alloc.argtypes = [C.POINTER(vp), sz]
    L.lpx_host_free.argtypes = [vp]
    L.lpx_host_free.restype = None
    L.lpx_pcd_info_read.argtypes = [C.c_char_p, C.POINTER(PcdInfo)]
    L.lpx_pcd_load.argtypes = [C.c_char_p, vp, sz, C.POINTER(PcdInfo)]
    L.lpx_feeder_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), u32, C.POINTER(vp)]
    L.lpx_feeder_destroy.argtypes = [vp]
    L.lpx_feeder_destroy.restype = None
    L.lpx_feeder_frames.argtypes = [vp]
    L.lpx_feeder_frames.restype = u32
    L.lpx_feeder_frame.argtypes = [vp, u32, C.POINTER(PcdInfo)]
    L.lpx_feeder_frame.restype = vp
    L.lpx_feeder_last_error.argtypes = [vp]
    L.lpx_feeder_last_error.restype = C.c_char_p
    L.lpx_feeder_run.argtypes = [vp, vp, vp, u32, C.POINTER(SegCfg), C.POINTER(CluCfg), C.POINTER(StreamOut)]
    L.lpx_feeder_run_multi.argtypes = [vp, C.POINTER(vp), u32, vp, u32, C.POINTER(SegCfg), C.POINTER(CluCfg),
                                       C.POINTER(StreamOut)]
    L.lpx_segment_cluster_batch_fields_device.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, vp, C.POINTER(SegCfg),
                                                          C.POINTER(CluCfg), vp, vp, vp, vp, vp, vp]
    L.lpx_create_batch.argtypes = [C.c_int, u32, C.POINTER(vp)]
    L.lpx_segment_cluster_batch_device.argtypes = [vp, u32, vp, sz, u32, vp, C.POINTER(SegCfg), C.POINTER(CluCfg), vp,
                                                   vp, vp, vp, vp, vp]
    L.lpx_segment_device.argtypes = [vp, vp, sz, u32, C.POINTER(SegCfg), vp, vp, vp, vp, vp]
    L.lpx_cluster_device.argtypes = [vp, vp, sz, u32, C.POINTER(CluCfg), vp, vp]
    L.lpx_profile_enable.argtypes = [vp, C.c_int]
    L.lpx_profile_stage_count.argtypes = []
    L.lpx_profile_stage_name.argtypes = [C.c_int]
    L.lpx_profile_stage_name.restype = C.c_char_p
    L.lpx_profile_read.argtypes = [vp, vp, vp, C.c_int]
    L.lpx_set_neighbour_mode.argtypes = [vp, C.c_int]
    L.lpx_dbg_sort_pairs.argtypes = [vp, vp, vp, u32, u32]
    L.lpx_dbg_sort_keys64.argtypes = [vp, vp, u32, u32]
    L.lpx_dbg_scan.argtypes = [vp, vp, u32, C.POINTER(C.c_uint64)]
    L.lpx_dbg_kd_layout.argtypes = [vp, vp, u32, vp]
    L.lpx_dbg_neighbours.argtypes = [vp, vp, u32, C.c_float, vp, vp, vp, C.c_uint64]
    L.lpx_dbg_components.argtypes = [vp, vp, u32, C.c_float, vp]
    L.lpx_dbg_plane.argtypes = [vp, vp, u32, vp]
    if hasattr(L, "lpx_build_info"):  # (an A/B variant built from an older tree, LPX_LIB, may lack it)
        L.lpx_build_info.argtypes = []
        L.lpx_build_info.restype = C.c_char_p
    _lib = L
    return L
