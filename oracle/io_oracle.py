"""oracle/io_oracle.py -- TEST INFRASTRUCTURE: CPU restatement of the point-cloud file formats the reference's loaders read
(SURVEY.md §8f rank 2).  Only tests/ may import it.

    read_kitti_bin   the reference's own code: np.fromfile(path, dtype=np.float32).reshape(-1, 4)[:, :3]   (dataset/kitti.py:76-80)
    read_ply / read_pcd
                     what open3d.io.read_point_cloud(path).points holds (dataset/threedmatch.py:75-79, dataset/tiers.py:72-75),
                     restated from the PLY 1.0 and PCD 0.7 specifications with numpy (open3d==0.18.0 is pinned by
                     requirements/base.txt:7 but absent from this image: PARITY UNPINNED for these two readers -- they are
                     checked against files produced by the independent writers below, byte layouts per the specifications).
    write_ply / write_pcd / lzf_compress
                     writers used by the tests to mint files in every storage mode.

All readers return float32 [n,3] (float64 coordinates rounded once, as `np.asarray(points, dtype=np.float32)` does)."""
import struct

import numpy as np

_PLY_T = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
          "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_kitti_bin(path):
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)[:, :3]


def read_ply(path):
    raw = open(path, "rb").read()
    end = raw.index(b"end_header")
    end = raw.index(b"\n", end) + 1
    header = raw[:end].decode("ascii").splitlines()
    assert header[0].strip() == "ply"
    fmt, elems = None, []
    for ln in header[1:]:
        w = ln.split()
        if not w:
            continue
        if w[0] == "format":
            fmt = w[1]
        elif w[0] == "element":
            elems.append([w[1], int(w[2]), []])
        elif w[0] == "property":
            elems[-1][2].append((w[4], "list", w[2], w[3]) if w[1] == "list" else (w[2], w[1]))
    body = raw[end:]
    if fmt == "ascii":
        lines = body.decode("ascii").splitlines()
        li = 0
        for name, count, props in elems:
            rows = lines[li:li + count]
            li += count
            if name != "vertex":
                continue
            out = np.zeros((count, 3))
            for r, ln in enumerate(rows):
                tok = ln.split()
                ti = 0
                for p in props:
                    if p[1] == "list":
                        c = int(tok[ti]); ti += 1 + c
                        continue
                    if p[0] in "xyz":
                        out[r, "xyz".index(p[0])] = float(tok[ti])
                    ti += 1
            return out.astype(np.float32)
        return np.zeros((0, 3), np.float32)
    bo = "<" if fmt == "binary_little_endian" else ">"
    pos = 0
    for name, count, props in elems:
        if all(p[1] != "list" for p in props):
            dt = np.dtype([(p[0], bo + _PLY_T[p[1]]) for p in props])
            arr = np.frombuffer(body, dt, count, pos)
            pos += dt.itemsize * count
            if name == "vertex":
                return np.stack([arr["x"], arr["y"], arr["z"]], 1).astype(np.float32)
        else:
            out = np.zeros((count, 3))
            for r in range(count):
                for p in props:
                    if p[1] == "list":
                        ct = np.dtype(bo + _PLY_T[p[2]]); it = np.dtype(bo + _PLY_T[p[3]])
                        c = int(np.frombuffer(body, ct, 1, pos)[0]); pos += ct.itemsize + c * it.itemsize
                    else:
                        t = np.dtype(bo + _PLY_T[p[1]])
                        v = np.frombuffer(body, t, 1, pos)[0]; pos += t.itemsize
                        if name == "vertex" and p[0] in "xyz":
                            out[r, "xyz".index(p[0])] = v
            if name == "vertex":
                return out.astype(np.float32)
    return np.zeros((0, 3), np.float32)


def write_ply(path, pts, fmt="binary_little_endian", coord="float", extra=(), faces=None, face_first=False, crlf=False):
    """extra: [(name, ply type, array)] additional vertex properties (interleaved around x/y/z); faces: int array [m,3] written as
    an element with a list property (before the vertices when face_first)."""
    pts = np.asarray(pts)
    n = len(pts)
    props = [("x", coord, pts[:, 0]), ("y", coord, pts[:, 1])] + list(extra[:1]) + [("z", coord, pts[:, 2])] + list(extra[1:])
    nl = "\r\n" if crlf else "\n"
    h = ["ply", f"format {fmt} 1.0", "comment written by oracle/io_oracle.py"]
    vert_h = [f"element vertex {n}"] + [f"property {t} {name}" for name, t, _ in props]
    face_h = [] if faces is None else [f"element face {len(faces)}", "property list uchar int vertex_indices"]
    h += (face_h + vert_h) if face_first else (vert_h + face_h)
    h.append("end_header")
    bo = ">" if fmt == "binary_big_endian" else "<"

    def vert_bytes():
        if fmt == "ascii":
            return "".join(" ".join(repr(float(a[r])) if _PLY_T[t][0] == "f" else str(int(a[r])) for _, t, a in props) + nl for r in range(n)).encode()
        rec = np.zeros(n, np.dtype([(name, bo + _PLY_T[t]) for name, t, _ in props]))
        for name, t, a in props:
            rec[name] = a
        return rec.tobytes()

    def face_bytes():
        if faces is None:
            return b""
        if fmt == "ascii":
            return "".join("3 " + " ".join(str(int(v)) for v in f) + nl for f in faces).encode()
        return b"".join(struct.pack(bo + "B3i", 3, *[int(v) for v in f]) for f in faces)

    with open(path, "wb") as f:
        f.write((nl.join(h) + nl).encode("ascii"))
        f.write((face_bytes() + vert_bytes()) if face_first else (vert_bytes() + face_bytes()))


def lzf_compress(data):
    """A valid (not a good) LZF stream: literal runs of up to 32 bytes, and a distance-1 back reference for every run of a
    repeated byte (3..264 bytes) -- exercises both token kinds, the extended length byte and overlapping copies."""
    out = bytearray()
    lit = bytearray()

    def flush():
        for k in range(0, len(lit), 32):
            chunk = lit[k:k + 32]
            out.append(len(chunk) - 1)
            out.extend(chunk)
        lit.clear()
    i, n = 0, len(data)
    while i < n:
        j = i
        while j < n and data[j] == data[i]:
            j += 1
        run = j - i
        if run >= 4:
            lit.append(data[i])              # the byte the reference copies from
            flush()
            rest = run - 1
            while rest >= 3:
                ln = min(rest, 264)          # token length field = ln - 2: 1..6 direct, 7 + one extension byte (0..255)
                if ln - 2 < 7:
                    out += bytes([(ln - 2) << 5, 0])
                else:
                    out += bytes([7 << 5, ln - 2 - 7, 0])
                rest -= ln
            lit.extend(data[j - rest:j])
            i = j
        else:
            lit.extend(data[i:j])
            i = j
    flush()
    return bytes(out)


def lzf_decompress(comp, out_len):
    out = bytearray()
    ip = 0
    while ip < len(comp):
        ctrl = comp[ip]; ip += 1
        if ctrl < 32:
            out.extend(comp[ip:ip + ctrl + 1]); ip += ctrl + 1
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += comp[ip]; ip += 1
            dist = ((ctrl & 31) << 8 | comp[ip]) + 1; ip += 1
            for _ in range(ln + 2):
                out.append(out[-dist])
    assert len(out) == out_len
    return bytes(out)


def write_pcd(path, pts, mode="binary", coord="F4", extra=()):
    """extra: [(name, 'F4'|'U4'|'U1'|'F8'..., array)] fields placed after x y z (one before z when two are given)."""
    pts = np.asarray(pts)
    n = len(pts)
    fields = [("x", coord, pts[:, 0]), ("y", coord, pts[:, 1])] + list(extra[:1]) + [("z", coord, pts[:, 2])] + list(extra[1:])
    npdt = {"F4": "<f4", "F8": "<f8", "U1": "u1", "U2": "<u2", "U4": "<u4", "I1": "i1", "I2": "<i2", "I4": "<i4"}
    h = ["# .PCD v0.7 - Point Cloud Data file format", "VERSION 0.7",
         "FIELDS " + " ".join(f[0] for f in fields), "SIZE " + " ".join(f[1][1] for f in fields),
         "TYPE " + " ".join(f[1][0] for f in fields), "COUNT " + " ".join("1" for _ in fields),
         f"WIDTH {n}", "HEIGHT 1", "VIEWPOINT 0 0 0 1 0 0 0", f"POINTS {n}", f"DATA {mode}"]
    with open(path, "wb") as f:
        f.write(("\n".join(h) + "\n").encode("ascii"))
        if mode == "ascii":
            for r in range(n):
                f.write((" ".join(repr(float(a[r])) if t[0] == "F" else str(int(a[r])) for _, t, a in fields) + "\n").encode())
        elif mode == "binary":
            rec = np.zeros(n, np.dtype([(name, npdt[t]) for name, t, _ in fields]))
            for name, t, a in fields:
                rec[name] = a
            f.write(rec.tobytes())
        else:
            raw = b"".join(np.asarray(a).astype(npdt[t]).tobytes() for _, t, a in fields)
            comp = lzf_compress(raw)
            f.write(struct.pack("<II", len(comp), len(raw)))
            f.write(comp)


def read_pcd(path):
    raw = open(path, "rb").read()
    pos = 0
    meta = {}
    while True:
        e = raw.index(b"\n", pos)
        ln = raw[pos:e].decode("ascii").strip()
        pos = e + 1
        if not ln or ln.startswith("#"):
            continue
        w = ln.split()
        meta[w[0]] = w[1:]
        if w[0] == "DATA":
            break
    fields, sizes, types = meta["FIELDS"], [int(s) for s in meta["SIZE"]], meta["TYPE"]
    n = int(meta["POINTS"][0]) if "POINTS" in meta else int(meta["WIDTH"][0]) * int(meta["HEIGHT"][0])
    dts = ["<" + {"F": "f", "I": "i", "U": "u"}[t] + str(s) for t, s in zip(types, sizes)]
    mode = meta["DATA"][0]
    if mode == "ascii":
        rows = [ln.split() for ln in raw[pos:].decode("ascii").splitlines()[:n]]
        cols = {f: np.array([float(r[i]) for r in rows]) for i, f in enumerate(fields)}
    elif mode == "binary":
        arr = np.frombuffer(raw, np.dtype(list(zip(fields, dts))), n, pos)
        cols = {f: arr[f] for f in fields}
    else:
        csz, usz = struct.unpack_from("<II", raw, pos)
        un = lzf_decompress(raw[pos + 8:pos + 8 + csz], usz)
        cols, o = {}, 0
        for f, d in zip(fields, dts):
            cols[f] = np.frombuffer(un, d, n, o)
            o += np.dtype(d).itemsize * n
    return np.stack([cols["x"], cols["y"], cols["z"]], 1).astype(np.float32)
