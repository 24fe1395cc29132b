"""Data ingest: the native readers (bx_io_*) against the format restatement in oracle/io_oracle.py and,
for KITTI .bin, against the reference's own expression np.fromfile(...).reshape(-1, 4)[:, :3] (dataset/kitti.py:76-80).
CPU only: no kernel is launched (the prefetcher has its own GPU test)."""
import os

import numpy as np
import pytest

from oracle import io_oracle as IO


@pytest.fixture(scope="module")
def ing():
    from bufferx_amd import ingest
    return ingest


def _cloud(n, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(n, 3)) * [3.0, 2.0, 0.5] + [1.0, -2.0, 0.25]).astype(dtype)


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
@pytest.mark.parametrize("coord", ["float", "double"])
def test_ply_plain(tmp_path, ing, fmt, coord):
    pts = _cloud(2500, 1, np.float64 if coord == "double" else np.float32)
    f = str(tmp_path / "c.ply")
    IO.write_ply(f, pts, fmt, coord)
    got = ing.read_point_cloud(f)
    assert ing.probe(f) == 2500
    assert got.dtype == np.float32 and np.array_equal(got, IO.read_ply(f)) and np.array_equal(got, pts.astype(np.float32))


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_extra_properties_and_faces(tmp_path, ing, fmt):
    """3DMatch fragments carry normals / colours; meshes carry a face element with a list property (before or after the vertices)."""
    pts = _cloud(700, 2)
    rng = np.random.default_rng(3)
    extra = [("nx", "float", rng.normal(size=700).astype(np.float32)), ("red", "uchar", rng.integers(0, 255, 700)),
             ("label", "short", rng.integers(-5, 5, 700))]
    faces = rng.integers(0, 700, (40, 3))
    for face_first in (False, True):
        f = str(tmp_path / f"m{int(face_first)}.ply")
        IO.write_ply(f, pts, fmt, "float", extra, faces, face_first, crlf=(fmt == "ascii"))
        got = ing.read_point_cloud(f)
        assert np.array_equal(got, pts) and np.array_equal(got, IO.read_ply(f))


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
@pytest.mark.parametrize("coord", ["F4", "F8"])
def test_pcd(tmp_path, ing, mode, coord):
    """TIERS .pcd: x y z intensity (+ ring); binary_compressed is LZF over a field-major layout."""
    pts = _cloud(3100, 4, np.float64 if coord == "F8" else np.float32)
    pts[100:400] = pts[100]            # constant stretches: back references in the LZF stream
    f = str(tmp_path / "c.pcd")
    IO.write_pcd(f, pts, mode, coord, extra=[("intensity", "F4", np.zeros(3100)), ("ring", "U2", np.arange(3100) % 64)])
    got = ing.read_point_cloud(f)
    assert ing.probe(f) == 3100
    assert np.array_equal(got, IO.read_pcd(f)) and np.array_equal(got, pts.astype(np.float32))


def test_kitti_bin_is_the_reference_expression(tmp_path, ing):
    rng = np.random.default_rng(5)
    xyzr = rng.normal(size=(12345, 4)).astype(np.float32)
    f = str(tmp_path / "000000.bin")
    xyzr.tofile(f)
    ref = np.fromfile(f, dtype=np.float32).reshape(-1, 4)[:, :3]          # dataset/kitti.py:76-80
    got = ing.read_point_cloud(f)
    assert np.array_equal(got, ref) and np.array_equal(got, IO.read_kitti_bin(f))


def test_empty_and_errors(tmp_path, ing):
    from bufferx_amd import lib
    f = str(tmp_path / "e.ply")
    IO.write_ply(f, np.zeros((0, 3), np.float32))
    assert ing.read_point_cloud(f).shape == (0, 3)
    with pytest.raises(lib.BxError):
        ing.read_point_cloud(str(tmp_path / "missing.ply"))
    bad = str(tmp_path / "bad.bin")
    open(bad, "wb").write(b"\0" * 20)                                     # not a multiple of 16 bytes
    with pytest.raises(lib.BxError):
        ing.read_point_cloud(bad)
    trunc = str(tmp_path / "t.ply")
    IO.write_ply(trunc, _cloud(100), "binary_little_endian")
    raw = open(trunc, "rb").read()
    open(trunc, "wb").write(raw[:-50])
    with pytest.raises(lib.BxError):
        ing.read_point_cloud(trunc)
    with pytest.raises(lib.BxError):
        ing.read_point_cloud(str(tmp_path / "x.xyz"))
    corrupt = str(tmp_path / "c.pcd")
    IO.write_pcd(corrupt, _cloud(500), "binary_compressed")
    raw = bytearray(open(corrupt, "rb").read())
    raw[-40] ^= 0xE0
    open(corrupt, "wb").write(bytes(raw[:-7]))
    with pytest.raises(lib.BxError):
        ing.read_point_cloud(corrupt)


def test_lzf_reference_vector():
    """The LZF token format, pinned on a hand-assembled stream: literal 'abc', then a back reference (length 9, distance 3)."""
    stream = bytes([2]) + b"abc" + bytes([(7 << 5) | 0, 0, 2])
    assert IO.lzf_decompress(stream, 12) == b"abcabcabcabc"
    data = bytes(np.random.default_rng(0).integers(0, 3, 4000, dtype=np.uint8)) + b"\0" * 1000
    assert IO.lzf_decompress(IO.lzf_compress(data), len(data)) == data


def test_corrupt_files_never_crash(tmp_path, ing):
    """Byte flips, truncations, inserted bytes and altered header digits: the readers either return an [n,3] float32 array or raise
    BxError -- no exception crosses the C boundary, no out-of-bounds access (400 mutated files over all seven storage modes)."""
    from bufferx_amd import lib
    rng = np.random.default_rng(99)
    pts = rng.normal(size=(200, 3)).astype(np.float32)
    seeds = []
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        f = str(tmp_path / f"a_{fmt}.ply")
        IO.write_ply(f, pts, fmt, extra=[("nx", "float", pts[:, 0])], faces=rng.integers(0, 200, (4, 3)), face_first=(fmt != "ascii"))
        seeds.append(f)
    for mode in ("ascii", "binary", "binary_compressed"):
        f = str(tmp_path / f"b_{mode}.pcd")
        IO.write_pcd(f, pts, mode, extra=[("i", "F4", pts[:, 0])])
        seeds.append(f)
    f = str(tmp_path / "c.bin")
    np.concatenate([pts, pts[:, :1]], 1).tofile(f)
    seeds.append(f)
    ok = err = 0
    for it in range(400):
        src = seeds[it % len(seeds)]
        raw = bytearray(open(src, "rb").read())
        mode = int(rng.integers(0, 4))
        if mode == 0:
            for _ in range(int(rng.integers(1, 6))):
                raw[int(rng.integers(0, len(raw)))] = int(rng.integers(0, 256))
        elif mode == 1:
            raw = raw[:int(rng.integers(0, len(raw)))]
        elif mode == 2:
            for _ in range(4):
                i = int(rng.integers(0, min(len(raw), 300)))
                if 48 <= raw[i] <= 57:
                    raw[i] = int(rng.integers(48, 58))
        else:
            i = int(rng.integers(0, len(raw)))
            raw[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))
        out = str(tmp_path / ("f" + os.path.splitext(src)[1]))
        open(out, "wb").write(bytes(raw))
        try:
            a = ing.read_point_cloud(out)
            assert a.ndim == 2 and a.shape[1] == 3 and a.dtype == np.float32
            ok += 1
        except (lib.BxError, MemoryError, ValueError):
            err += 1
    assert ok + err == 400 and err > 50


# ---------------------------------------------------------------------------------------------------- hand-assembled known answers
# tests/golden/io_kat/*: files assembled byte by byte from the PLY 1.0 / PCD 0.7 format descriptions by tests/golden/io_kat/make_kat.py,
# which uses neither oracle/io_oracle.py's writers nor the product -- the expected coordinates are the literals below.
KAT_XYZ = np.array([(0.5, -1.25, 2.0), (1.0, 0.0, -0.75), (-3.5, 4.25, 0.125), (100.0, -0.0625, 7.0), (-8.0, 16.0, -32.0)], np.float32)
KAT_FILES = ["kat_ascii.ply", "kat_le_faces_last.ply", "kat_be_faces_first.ply", "kat_double_be_crlf.ply",
             "kat_ascii.pcd", "kat_binary.pcd", "kat_compressed.pcd"]


@pytest.mark.parametrize("name", KAT_FILES)
def test_known_answer_files(ing, name):
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_kat", name)
    assert ing.probe(f) == 5
    got = ing.read_point_cloud(f)
    assert got.dtype == np.float32 and np.array_equal(got, KAT_XYZ)
    ref = IO.read_ply(f) if name.endswith(".ply") else IO.read_pcd(f)      # the format restatement is pinned by the same files
    assert np.array_equal(ref, KAT_XYZ)


def test_known_answer_files_are_what_the_recipe_makes(tmp_path):
    """the committed bytes == the output of the byte-level recipe (so the files cannot drift from their documented construction)"""
    import importlib.util
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_kat")
    spec = importlib.util.spec_from_file_location("make_kat", os.path.join(d, "make_kat.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    made = {"kat_ascii.ply": mk.ply_ascii(), "kat_le_faces_last.ply": mk.ply_binary("little", False),
            "kat_be_faces_first.ply": mk.ply_binary("big", True), "kat_double_be_crlf.ply": mk.ply_double_be_crlf(),
            "kat_ascii.pcd": mk.pcd_ascii(), "kat_binary.pcd": mk.pcd_binary(), "kat_compressed.pcd": mk.pcd_compressed()}
    for name, data in made.items():
        assert open(os.path.join(d, name), "rb").read() == data, name


@pytest.mark.parametrize("count", [-1, 200, 2**31 - 1])
def test_ply_corrupt_list_length_is_rejected(tmp_path, ing, count):
    """a list length that the file cannot hold (negative, or larger than the rest of the body) is an error, not a loop or a backwards seek"""
    import struct
    from bufferx_amd import lib
    hdr = (b"ply\nformat binary_little_endian 1.0\nelement face 1\nproperty list int int vertex_indices\n"
           b"element vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
    f = str(tmp_path / "bad_list.ply")
    open(f, "wb").write(hdr + struct.pack("<i3i", count, 0, 1, 2) + struct.pack("<3f", 1.0, 2.0, 3.0))
    with pytest.raises(lib.BxError):
        ing.read_point_cloud(f)
    fa = str(tmp_path / "bad_list_ascii.ply")
    open(fa, "wb").write(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty list uchar int junk\nproperty float x\nproperty float y\n"
                         b"property float z\nend_header\n" + str(count).encode() + b" 1 2 3 1.0 2.0 3.0\n")
    with pytest.raises(lib.BxError):
        ing.read_point_cloud(fa)
