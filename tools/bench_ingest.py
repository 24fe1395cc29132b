#!/usr/bin/env python
"""tools/bench_ingest.py -- host-side timing of the native point-cloud readers (bx_io_read_xyz); no GPU needed.
One JSON line per format."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best(fn, rep=5):
    t = []
    for _ in range(rep):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return min(t) * 1e3


def main():
    from bufferx_amd import ingest
    rng = np.random.default_rng(0)
    d = tempfile.mkdtemp()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    pts = (rng.normal(size=(n, 3)) * 3).astype(np.float32)
    cases = []
    # binary PLY with two extra float properties interleaved (x y nx z ny)
    f = os.path.join(d, "a.ply")
    rec = np.zeros(n, np.dtype([("x", "<f4"), ("y", "<f4"), ("nx", "<f4"), ("z", "<f4"), ("ny", "<f4")]))
    rec["x"], rec["y"], rec["z"], rec["nx"], rec["ny"] = pts[:, 0], pts[:, 1], pts[:, 2], pts[:, 0], pts[:, 1]
    with open(f, "wb") as h:
        h.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join("property float %s\n" % k for k in rec.dtype.names)
                 + "end_header\n").encode())
        h.write(rec.tobytes())
    cases.append(("ply binary (x y nx z ny)", f))
    # binary PCD x y z intensity
    f = os.path.join(d, "b.pcd")
    with open(f, "wb") as h:
        h.write(("VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                 "POINTS %d\nDATA binary\n" % (n, n)).encode())
        h.write(np.concatenate([pts, pts[:, :1]], 1).astype("<f4").tobytes())
    cases.append(("pcd binary (x y z intensity)", f))
    f = os.path.join(d, "c.bin"); np.concatenate([pts, pts[:, :1]], 1).tofile(f); cases.append(("kitti bin", f))
    m = min(n, 50000)
    f = os.path.join(d, "d.ply")
    with open(f, "w") as h:
        h.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % m)
        h.write("".join("%r %r %r\n" % (float(a), float(b), float(c)) for a, b, c in pts[:m]))
    cases.append(("ply ascii (%d pts)" % m, f))
    for name, f in cases:
        a = ingest.read_point_cloud(f)
        assert np.array_equal(a, pts[:len(a)])
        row = dict(format=name, points=len(a), file_MB=round(os.path.getsize(f) / 1e6, 2), native_ms=round(best(lambda: ingest.read_point_cloud(f)), 3))
        if name == "kitti bin":       # the reference's own expression (a view: no copy is made)
            row["numpy_fromfile_ms"] = round(best(lambda: np.fromfile(f, dtype=np.float32).reshape(-1, 4)[:, :3]), 3)
        print(json.dumps(row))


if __name__ == "__main__":
    main()
