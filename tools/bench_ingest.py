#!/usr/bin/env python
"""tools/bench_ingest.py -- host-side timing of the native point-cloud readers (bx_io_read_xyz) beside the numpy restatement
(oracle/io_oracle.py); no GPU needed.  One JSON line per format."""
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
    from oracle import io_oracle as IO
    rng = np.random.default_rng(0)
    d = tempfile.mkdtemp()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    pts = (rng.normal(size=(n, 3)) * 3).astype(np.float32)
    cases = []
    f = os.path.join(d, "a.ply"); IO.write_ply(f, pts, "binary_little_endian", extra=[("nx", "float", pts[:, 0]), ("ny", "float", pts[:, 1])]); cases.append(("ply binary (x y nx z ny)", f, IO.read_ply))
    f = os.path.join(d, "b.pcd"); IO.write_pcd(f, pts, "binary", extra=[("intensity", "F4", pts[:, 0])]); cases.append(("pcd binary (x y intensity z)", f, IO.read_pcd))
    f = os.path.join(d, "c.bin"); np.concatenate([pts, pts[:, :1]], 1).tofile(f); cases.append(("kitti bin", f, IO.read_kitti_bin))
    m = min(n, 50000)
    f = os.path.join(d, "d.ply"); IO.write_ply(f, pts[:m], "ascii"); cases.append(("ply ascii (%d pts)" % m, f, IO.read_ply))
    for name, f, ref in cases:
        a = ingest.read_point_cloud(f)
        assert np.array_equal(a, ref(f))
        print(json.dumps(dict(format=name, points=len(a), file_MB=round(os.path.getsize(f) / 1e6, 2), native_ms=round(best(lambda: ingest.read_point_cloud(f)), 3),
                              numpy_ms=round(best(lambda: ref(f), 2), 3))))


if __name__ == "__main__":
    main()
