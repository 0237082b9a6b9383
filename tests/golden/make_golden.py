"""Regenerates tests/golden/*.npz.  Runs only in the build container (needs /root/reference).

  frames.npz  three of the reference's data/*.pcd frames repacked losslessly (coordinates are exact
              1 mm multiples, intensity exact 0.01 multiples; float32(int / 1000) reproduces every
              value, negative zeros are listed separately).  These are DATA the reference ships.
  golden.npz  expected outputs:
              clu_*  int32 labels from the REFERENCE's own kdtree.hpp + queue.hpp build
                     (oracle/_ref/libkdref.so, see oracle/ref_driver.cpp) -- the pinned half
              seg_*  labels / planes from the oracle restatement (segmentation cannot be built from
                     the reference here: Eigen is absent) -- regression lock, "parity unpinned"
              kd_*   kd-tree pre-order of the reference build for a real obstacle cloud
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from lidar_processing_amd.pcd import read_pcd  # noqa: E402

FRAMES = ["0000000000", "0000000077", "0000000153"]
SEG_CFGS = {"p2i3": dict(number_of_planar_partitions=2, number_of_iterations=3),
            "p3i3": dict(number_of_planar_partitions=3, number_of_iterations=3),
            "p6i5": dict(number_of_planar_partitions=6, number_of_iterations=5)}
CLU_CFGS = {"d018q05": (0.18, 0.5), "d025q05": (0.25, 0.5), "d018q10": (0.18, 1.0)}


def pack(pts):
    q = np.rint(pts[:, :3].astype(np.float64) * 1000.0).astype(np.int32)
    qi = np.rint(pts[:, 3].astype(np.float64) * 100.0).astype(np.int16)
    rec = np.empty_like(pts)
    rec[:, :3] = (q / 1000.0).astype(np.float32)
    rec[:, 3] = (qi / 100.0).astype(np.float32)
    assert np.array_equal(rec, pts), "frame is not 1 mm / 0.01 quantised"
    negzero = np.argwhere(np.signbit(pts) & (pts == 0))
    return q, qi, negzero.astype(np.int32)


def main():
    assert oracle.ref() is not None, "oracle/_ref/libkdref.so missing"
    frames, gold = {}, {}
    for f in FRAMES:
        pts, fields = read_pcd(f"/root/reference/data/{f}.pcd")
        assert fields == ["x", "y", "z", "intensity"]
        q, qi, nz = pack(pts)
        frames[f"{f}_xyz_mm"] = q
        frames[f"{f}_intensity_c"] = qi
        frames[f"{f}_negzero"] = nz
        for sname, skw in SEG_CFGS.items():
            r = oracle.segment(pts, oracle.SegCfg(**skw))
            assert r["rc"] == 0
            gold[f"seg_{f}_{sname}_labels"] = r["labels"].astype(np.uint8)
            gold[f"seg_{f}_{sname}_planes"] = r["planes"]
            gold[f"seg_{f}_{sname}_counts"] = np.array([len(r["ground_idx"]), len(r["obstacle_idx"])], np.uint32)
            # obstacle_idx is fully determined by labels + x order; keep a checksum of the order
            gold[f"seg_{f}_{sname}_oidx_crc"] = np.array(
                [np.bitwise_xor.reduce((r["obstacle_idx"].astype(np.uint64) + 1) *
                                       (np.arange(len(r["obstacle_idx"]), dtype=np.uint64) * 2654435761 + 1)
                                       % (2 ** 61 - 1))], np.uint64)
            if sname == "p3i3":
                continue
            obs = pts[r["obstacle_idx"]]
            for cname, (d2, qual) in CLU_CFGS.items():
                if f != "0000000000" and cname == "d018q10":
                    continue  # exact EC replay of the larger frames is slow on the CPU; one is enough
                lab, nc = oracle.ref_fec(obs, oracle.CluCfg(d2, qual))
                lab2, nc2 = oracle.cluster(obs, oracle.CluCfg(d2, qual))
                assert nc == nc2 and np.array_equal(lab, lab2), "oracle restatement != reference build"
                gold[f"clu_{f}_{sname}_{cname}_labels"] = lab
                gold[f"clu_{f}_{sname}_{cname}_n"] = np.array([nc], np.uint32)
            if f == "0000000000" and sname == "p2i3":
                gold["kd_0000000000_p2i3_preorder"] = oracle.ref_kd_preorder(obs)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames.npz"), **frames)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden.npz"), **gold)
    for n in ("frames.npz", "golden.npz"):
        print(n, os.path.getsize(os.path.join(ROOT, "tests", "golden", n)) // 1024, "KiB")


if __name__ == "__main__":
    main()
