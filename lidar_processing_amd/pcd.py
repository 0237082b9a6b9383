"""Binary PCD v0.7 reader/writer for float32 `x y z intensity` clouds.

Counterpart of the reference's input harness (`pcl::io::loadPCDFile` in src/dataloader.cpp:128-153);
only what the hot path needs: header fields as in data/0000000000.pcd:1-11, `DATA binary`, exactly
POINTS records are read (the reference files carry trailing bytes after the payload).
"""
import numpy as np


def read_pcd(path):
    with open(path, "rb") as f:
        raw = f.read()
    marker = b"DATA binary\n"
    pos = raw.find(marker)
    if pos < 0:
        raise ValueError("only `DATA binary` PCD files are supported")
    header = {}
    for line in raw[:pos].decode("ascii", "replace").splitlines():
        if line and not line.startswith("#"):
            k, *v = line.split()
            header[k] = v
    fields = header.get("FIELDS", [])
    if header.get("SIZE") != ["4"] * len(fields) or header.get("TYPE") != ["F"] * len(fields) or \
            header.get("COUNT", ["1"] * len(fields)) != ["1"] * len(fields):
        raise ValueError("only float32 scalar fields are supported")
    n = int(header["POINTS"][0])
    k = len(fields)
    data = np.frombuffer(raw, dtype="<f4", count=n * k, offset=pos + len(marker)).reshape(n, k)
    return np.ascontiguousarray(data), fields


def write_pcd(path, points, fields=("x", "y", "z", "intensity")):
    pts = np.ascontiguousarray(points, dtype="<f4")
    n, k = pts.shape
    assert k == len(fields)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
           f"FIELDS {' '.join(fields)}\nSIZE {' '.join(['4'] * k)}\nTYPE {' '.join(['F'] * k)}\n"
           f"COUNT {' '.join(['1'] * k)}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(pts.tobytes())
