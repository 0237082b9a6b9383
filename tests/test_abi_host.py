"""CPU tests: the C-ABI library loads and exports every symbol include/lpx.h declares, the product
path fails loudly without a GPU, the drop-in C++ headers compile against a processor.cpp-like caller,
host-side helpers (PCD reader, argument checks), and the N>1 bench path with gloo, world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="module")
def liblpx():
    from lidar_processing_amd import _lib
    _lib.build()
    return C.CDLL(_lib.LIB_PATH)


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(lpx_[a-z0-9_]+)\s*\(", hdr))


def test_every_declared_symbol_is_exported(liblpx):
    names = _declared("lpx.h")
    assert len(names) >= 20
    missing = [n for n in sorted(names) if not hasattr(liblpx, n)]
    assert not missing, f"declared in lpx.h but not exported: {missing}"
    assert not [n for n in names if n.startswith("lpx_dbg_")], "test hooks belong in lpx_debug.h"
    dbg = _declared("lpx_debug.h")
    assert dbg and all(n.startswith("lpx_dbg_") for n in dbg)
    assert not [n for n in sorted(dbg) if not hasattr(liblpx, n)]


def test_every_exported_symbol_is_declared():
    """nothing is exported that neither header declares (the boundary is what the headers say)"""
    from lidar_processing_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("lpx_")}
    assert exported <= (_declared("lpx.h") | _declared("lpx_debug.h")), exported - _declared("lpx.h") - _declared("lpx_debug.h")


def test_release_library_reads_no_environment_knobs():
    """the product library holds no LPX_* knob at all (LPX_SKIP and friends can make results wrong: they exist only in
    liblpx_dev.so, -DLPX_DEV_KNOBS) and never calls setenv; the development build is a separate file that nothing loads
    by default"""
    from lidar_processing_amd import _lib
    _lib.build()
    strs = subprocess.run(["strings", _lib.LIB_PATH], capture_output=True, text=True).stdout
    knobs = sorted(set(re.findall(r"\bLPX_[A-Z0-9_]+", strs)))
    assert knobs == [], knobs
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "setenv" not in und and "putenv" not in und
    assert os.path.basename(_lib.LIB_PATH) == "liblpx.so" or os.environ.get("LPX_LIB")
    L = C.CDLL(_lib.LIB_PATH)
    L.lpx_build_info.restype = C.c_char_p
    if not os.environ.get("LPX_LIB"):
        assert b"release build" in L.lpx_build_info()
    dev = subprocess.run(["strings", _lib.DEV_LIB_PATH], capture_output=True, text=True).stdout
    assert "LPX_SKIP" in dev and "LPX_POISON" in dev  # the knobs the tools and four GPU tests use live here


def test_no_oracle_symbols_in_product_library():
    """the product never links the oracle"""
    from lidar_processing_amd import _lib
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out and "ref_fec" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "lidar_processing_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "lidar_oracle" not in src, f


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_fails_loudly_without_gpu(liblpx):
    h = C.c_void_p()
    assert liblpx.lpx_create(0, C.byref(h)) == -5  # LPX_ERR_NO_DEVICE: no CPU fallback
    from lidar_processing_amd import Context, LpxError, Segmenter
    with pytest.raises(LpxError):
        Context(0)
    with pytest.raises(LpxError):
        Segmenter()


def test_profile_stage_names(liblpx):
    liblpx.lpx_profile_stage_name.restype = C.c_char_p
    n = liblpx.lpx_profile_stage_count()
    names = [liblpx.lpx_profile_stage_name(i).decode() for i in range(n)]
    assert n == 14 and len(set(names)) == n and "replay" in names and "plane_passes" in names
    sys.path.insert(0, ROOT)
    import bench
    for s in names:  # every stage has an algorithmic-bytes formula
        if s == "groups":
            continue  # not part of the timed path
        assert bench.algorithmic_bytes(s, 120000, 50000, 6e6, 5, 6) > 0


def test_dropin_headers_compile_like_processor(tmp_path):
    """include/lidar_processing/{segmentation,clustering}.hpp against the call sequence of
    reference src/processor.cpp:150-200 (tests/cxx/dropin_main.cpp)"""
    from lidar_processing_amd import _lib
    _lib.build()
    exe = tmp_path / "dropin_main"
    cmd = ["g++", "-std=c++17", "-O1", f"-I{ROOT}/include", f"-I{ROOT}/include/lidar_processing",
           f"-I{ROOT}/tests/cxx", f"{ROOT}/tests/cxx/dropin_main.cpp", "-o", str(exe),
           f"-L{ROOT}/lidar_processing_amd", "-llpx", f"-Wl,-rpath,{ROOT}/lidar_processing_amd",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_latency_harness_compiles_against_the_headers(tmp_path):
    """tests/cxx/dropin_latency.cpp -- what bench.py builds on the GPU box for latency.dropin_cxx (three set-ups: shared
    context with the look-ahead, without it, a context each) -- compiles and links here"""
    from lidar_processing_amd import _lib
    _lib.build()
    exe = tmp_path / "dropin_latency"
    cmd = ["g++", "-std=c++17", "-O1", f"-I{ROOT}/include", f"-I{ROOT}/include/lidar_processing",
           f"-I{ROOT}/tests/cxx", f"{ROOT}/tests/cxx/dropin_latency.cpp", "-o", str(exe),
           f"-L{ROOT}/lidar_processing_amd", "-llpx", f"-Wl,-rpath,{ROOT}/lidar_processing_amd",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_degrade_harness_compiles_against_the_headers(tmp_path):
    """tests/cxx/dropin_degrade.cpp (Clusterer::cluster's no-throw degrade path and the ownership guard of regroup /
    convex_outlines, run on the GPU by tests/test_gpu_pipeline.py) compiles and links here"""
    from lidar_processing_amd import _lib
    _lib.build()
    exe = tmp_path / "dropin_degrade"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT}/include", f"-I{ROOT}/include/lidar_processing",
           f"-I{ROOT}/tests/cxx", f"{ROOT}/tests/cxx/dropin_degrade.cpp", "-o", str(exe),
           f"{ROOT}/lidar_processing_amd/liblpx_dev.so", f"-Wl,-rpath,{ROOT}/lidar_processing_amd",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_binding_recipe_survives_quoted_include_resolution(tmp_path):
    """src/processor.cpp includes "segmentation.hpp" / "clustering.hpp" with quotes, and a quoted include searches
    the including file's own directory before any -I path.  A caller that sits NEXT TO same-named CPU headers (as
    processor.cpp does in the reference's src/) therefore never sees the drop-in headers through an include path
    alone; integration/apply_to_reference.sh replaces the two headers by forwarding shims.  Decoy headers that
    refuse to compile stand in for the reference's."""
    from lidar_processing_amd import _lib
    _lib.build()
    ref = tmp_path / "reference"
    (ref / "src").mkdir(parents=True)
    for h in ("segmentation", "clustering"):
        (ref / "src" / f"{h}.hpp").write_text('#error "the CPU header of the reference was picked"\n')
        (ref / "src" / f"{h}.cpp").write_text("// cpu implementation\n")
    tu = ref / "src" / "processor_like.cpp"
    tu.write_text(open(os.path.join(ROOT, "tests", "cxx", "dropin_main.cpp")).read())

    def compile_tu():
        return subprocess.run(["g++", "-std=c++17", "-O0", "-fsyntax-only", f"-I{ROOT}/include/lidar_processing",
                               f"-I{ROOT}/include", f"-I{ROOT}/tests/cxx", str(tu)], capture_output=True, text=True)

    r = compile_tu()
    assert r.returncode != 0 and "CPU header of the reference was picked" in r.stderr  # the pitfall is real
    r = subprocess.run([os.path.join(ROOT, "integration", "apply_to_reference.sh"), str(ref), ROOT],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (ref / "src" / "segmentation_cpu.hpp").exists() and (ref / "src" / "clustering_cpu.cpp").exists()
    r = compile_tu()
    assert r.returncode == 0, r.stderr
    r = subprocess.run([os.path.join(ROOT, "integration", "apply_to_reference.sh"), str(ref), ROOT],
                       capture_output=True, text=True)  # idempotent: the originals are not overwritten by a shim
    assert r.returncode == 0 and "#error" in (ref / "src" / "segmentation_cpu.hpp").read_text()


def test_config_structs_mirror_reference_defaults():
    from lidar_processing_amd import ClusteringConfiguration, Clusterer, SegmentationConfiguration, SegmentationLabel
    s = SegmentationConfiguration()
    assert (s.sensor_height_m, s.orthogonal_distance_threshold, s.initial_seed_threshold) == (1.73, 0.3, 0.6)
    assert (s.number_of_iterations, s.number_of_planar_partitions, s.number_of_lower_point_representatives) == \
        (3, 2, 5000)
    c = ClusteringConfiguration()
    assert (c.distance_squared, c.cluster_quality, c.min_cluster_size, c.max_cluster_size) == \
        (0.18, 0.5, 4, 2 ** 32 - 1)
    assert Clusterer.UNDEFINED == -2 ** 31 and Clusterer.INVALID == -1
    assert [int(x) for x in SegmentationLabel] == [0, 1, 2]
    from lidar_processing_amd import _lib
    assert C.sizeof(_lib.SegCfg) == 24 and C.sizeof(_lib.CluCfg) == 16


def test_pcd_roundtrip(tmp_path):
    from lidar_processing_amd import read_pcd, write_pcd
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(1234, 4)).astype(np.float32)
    p = tmp_path / "a.pcd"
    write_pcd(p, pts)
    with open(p, "ab") as f:
        f.write(b"\0" * 3900)  # the reference's files carry trailing bytes after the payload
    back, fields = read_pcd(p)
    assert fields == ["x", "y", "z", "intensity"] and np.array_equal(back, pts)


def test_c_pcd_loader_on_host_memory(tmp_path):
    """lpx_pcd_info_read / lpx_pcd_load are plain file I/O: they run without a GPU into ordinary memory"""
    from lidar_processing_amd import load_pcd, pcd_info, write_pcd, LpxError
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(4321, 4)).astype(np.float32)
    p = tmp_path / "a.pcd"
    write_pcd(p, pts)
    with open(p, "ab") as f:
        f.write(b"\0" * 3900)
    assert pcd_info(p) == dict(n_points=4321, point_step=16, offsets=(0, 4, 8), n_fields=4)
    got, _ = load_pcd(p, pinned=False)
    assert np.array_equal(got.view(np.uint32), pts.view(np.uint32))
    with pytest.raises(LpxError):
        pcd_info(tmp_path / "missing.pcd")
    ref = "/root/reference/data/0000000077.pcd"
    if os.path.exists(ref):  # build container: the reference's own file, against the committed fixture
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import load_frame
        got, info = load_pcd(ref, pinned=False)
        assert info["n_points"] == 124049 and np.array_equal(got.view(np.uint32), load_frame("0000000077").view(np.uint32))


def test_frame_sharding_is_round_robin():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.frame_ids_for_rank(0, 1, 6, 3) == [0, 1, 2, 0, 1, 2]
    got = [bench.frame_ids_for_rank(r, 4, 3, 154) for r in range(4)]
    assert got == [[0, 4, 8], [1, 5, 9], [2, 6, 10], [3, 7, 11]]
    # over all ranks the first world*F stream positions are covered exactly once
    assert sorted(sum(got, [])) == list(range(12))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t, pts = bench.aggregate(0.5 + rank, 1000.0 * (rank + 1), torch.device("cpu"), world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, t, pts))


def test_multi_rank_aggregation_gloo():
    """the only cross-rank exchange of the bench: MAX of the times, SUM of the points (world_size 2, gloo)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, 1.5, 3000.0), (1, 1.5, 3000.0)]


def _worker_world1(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=0, world_size=1)
    calls = []
    real_ar, real_ag = dist.all_reduce, dist.all_gather
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), real_ar(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), real_ag(*a, **k))[1]
    agg = bench.aggregate(0.125, 1000.0, torch.device("cpu"), 1, 64)
    rows = bench.gather_per_rank([1.5, -2.25], torch.device("cpu"), 1)
    dist.destroy_process_group()
    q.put((agg, rows, calls))


def test_world_1_group_still_runs_the_collectives():
    """the RCCL self-test (`bench.py --dist-selftest`, world 1) must EXECUTE the all-reduces and the all-gather of the
    N > 1 line, not skip them because world == 1: with a process group alive aggregate() / gather_per_rank() go through
    torch.distributed (gloo here; the RCCL run is tests/test_gpu_stream.py::test_rccl_world_1_selftest)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_world1, args=(31500 + os.getpid() % 2000, q))
    p.start()
    agg, rows, calls = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert agg == (0.125, 1000.0, 64.0) and rows == [[1.5, -2.25]]
    assert calls == ["all_reduce", "all_reduce", "all_gather"]
    # and without a group nothing distributed is touched at world 1
    import torch
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(0.5, 10.0, torch.device("cpu"), 1) == (0.5, 10.0)
    assert bench.gather_per_rank([3.0], torch.device("cpu"), 1) == [[3.0]]
    assert bench.pci_id_str(-1.0) is None and bench.pci_id_str(float((1 << 16) | (0x2f << 8) | 3)) == "0001:2f:03.0"


def test_bench_gpus_2_launches_two_ranks_by_itself_dry_run():
    """`python bench.py --gpus 2` without a launcher starts its two ranks itself (frame i -> GPU i mod N,
    no data-path collective) and rank 0 reports the world it ran in.  --dry-run --backend gloo walks main() end to
    end without a GPU: spawn, rendezvous on 127.0.0.1, sharding, MAX / SUM aggregation, the JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo",
                        "--workload", "kitti", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] == 0.0
    assert line["config"]["distributed_world_size"] == 2 and line["config"]["distributed_backend"] == "gloo"
    # both ranks' frames are in the aggregate: 2 x 512 frames, rank 0 holds frames 0, 2, 4, ... of the stream
    assert line["config"]["frames_per_step"] == 2 * line["config"]["frames_per_step_per_gpu"]
    assert line["config"]["frame_ids_rank0_head"] == [0, 2, 1, 0]  # i mod 3 over the three committed frames
    assert line["scaling"] == "weak" and line["steps"] == 2
    # what every rank measured by itself is gathered (not only the MAX), and at N > 1 rank 0 runs no side legs while the
    # other ranks would wait in the final barrier holding their GPUs
    assert [p["rank"] for p in line["per_rank"]] == [0, 1] and all(p["frames_per_s"] > 0 for p in line["per_rank"])
    assert not {"latency", "throughput_vs_inflight", "stream", "kitti_3_frames_cycled", "beyond_latency_budget"} & set(line)


def test_bench_under_a_launcher_does_not_spawn():
    """with WORLD_SIZE set (torch.distributed.run) bench.py is one rank of an existing job"""
    import json
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--workload", "kitti",
                        "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1


def test_library_leaves_the_environment_alone():
    """loading liblpx.so changes nothing in the process environment (round 3's library set GPU_MAX_HW_QUEUES from a
    constructor: not thread-safe against a running host, invisible to the user, dependent on load order); the launcher
    sets it -- bench.py, tests/conftest.py, INTEGRATION.md -- and lpx_build_info() reports what the process has.
    Checked in child processes: the variable is process state."""
    # (os.environ is a snapshot of the start of the interpreter: ask the C library)
    code = ("import ctypes; L = ctypes.CDLL(%r); g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; "
            "L.lpx_build_info.restype = ctypes.c_char_p; v = g(b'GPU_MAX_HW_QUEUES'); "
            "print(v.decode() if v else 'unset', '|', L.lpx_build_info().decode())"
            % os.path.join(ROOT, "lidar_processing_amd", "liblpx.so"))
    base = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("unset |") and "GPU_MAX_HW_QUEUES=unset" in r.stdout, (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code], env=dict(base, GPU_MAX_HW_QUEUES="8"), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("8 |") and "GPU_MAX_HW_QUEUES=8" in r.stdout, (r.stdout, r.stderr[-500:])


def test_tools_index_names_every_script():
    """tools/README.md is the index of the development scripts: every script is named there"""
    text = open(os.path.join(ROOT, "tools", "README.md")).read()
    missing = [f for f in sorted(os.listdir(os.path.join(ROOT, "tools")))
               if f.endswith((".py", ".sh")) and f not in text]
    assert not missing, missing
