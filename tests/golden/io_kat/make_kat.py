"""Known-answer point-cloud files assembled BYTE BY BYTE from the format descriptions -- PLY 1.0 (Turk, "The PLY polygon file
format": header grammar, `format ascii|binary_little_endian|binary_big_endian 1.0`, `property list <count type> <item type>`) and
PCD 0.7 (PCL "The PCD file format": FIELDS / SIZE / TYPE / COUNT / POINTS / DATA ascii|binary|binary_compressed; the compressed
body is  uint32 compressed size, uint32 uncompressed size, LZF stream  over the FIELD-MAJOR layout) -- and NOT with the writers of
oracle/io_oracle.py, so that the native readers (buffer-x_amd/csrc/k_io.hip) are pinned by something other than this repository's
own restatement.  The expected coordinates are literals in tests/test_ingest.py::KAT_XYZ.  Re-run: python make_kat.py
(the committed files are its output; nothing here imports the oracle or the product).

LZF stream format (liblzf, as PCL uses it): a control byte c < 32 starts a literal run of c + 1 bytes; otherwise a back reference of
length (c >> 5) + 2 -- when (c >> 5) == 7 the next byte is added to the length -- at distance (((c & 31) << 8) | next byte) + 1."""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))

# five vertices; every value is exactly representable in binary32 (and prints exactly in the ascii files)
XYZ = [(0.5, -1.25, 2.0), (1.0, 0.0, -0.75), (-3.5, 4.25, 0.125), (100.0, -0.0625, 7.0), (-8.0, 16.0, -32.0)]
NRM = [(0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, -1.0), (0.5, 0.5, 0.0)]
RGB = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (10, 20, 30), (200, 100, 50)]
FACES = [(0, 1, 2), (2, 3, 4, 0)]          # a triangle and a quad: list lengths 3 and 4


def w(name, data):
    with open(os.path.join(HERE, name), "wb") as f:
        f.write(data)


# ---------------------------------------------------------------------------------------------------- PLY
def ply_header(fmt, face_first):
    vert = ("element vertex 5\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\n"
            "property float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n")
    face = "element face 2\nproperty list uchar int vertex_indices\n"
    body = (face + vert) if face_first else (vert + face)
    return ("ply\nformat %s 1.0\ncomment hand-assembled known-answer file\n" % fmt + body + "end_header\n").encode("ascii")


def ply_ascii():
    out = ply_header("ascii", False)
    for p, n, c in zip(XYZ, NRM, RGB):
        out += ("%r %r %r %r %r %r %d %d %d\n" % (p + n + c)).encode("ascii")
    for f in FACES:
        out += (" ".join(str(v) for v in (len(f),) + f) + "\n").encode("ascii")
    return out


def ply_binary(endian, face_first):
    e = "<" if endian == "little" else ">"
    verts = b"".join(struct.pack(e + "6f3B", *(p + n + c)) for p, n, c in zip(XYZ, NRM, RGB))
    faces = b"".join(struct.pack(e + "B%di" % len(f), len(f), *f) for f in FACES)
    return ply_header("binary_%s_endian" % endian, face_first) + ((faces + verts) if face_first else (verts + faces))


# double-precision coordinates, big endian, CRLF header line ends (a Windows writer), no extra properties
def ply_double_be_crlf():
    hdr = "ply\r\nformat binary_big_endian 1.0\r\nelement vertex 5\r\nproperty double x\r\nproperty double y\r\nproperty double z\r\nend_header\r\n"
    return hdr.encode("ascii") + b"".join(struct.pack(">3d", *p) for p in XYZ)


# ---------------------------------------------------------------------------------------------------- PCD
def pcd_header(data, fields="x y z intensity", size="4 4 4 4", typ="F F F F", count="1 1 1 1"):
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH 5\nHEIGHT 1\n"
            "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 5\nDATA %s\n" % (fields, size, typ, count, data)).encode("ascii")


INTENSITY = [0.0, 1.0, 2.0, 3.0, 4.0]


def pcd_ascii():
    out = pcd_header("ascii")
    for p, i in zip(XYZ, INTENSITY):
        out += ("%r %r %r %r\n" % (p + (i,))).encode("ascii")
    return out


def pcd_binary():
    return pcd_header("binary") + b"".join(struct.pack("<4f", *(p + (i,))) for p, i in zip(XYZ, INTENSITY))


def pcd_compressed():
    """field-major payload: all x, all y, all z, all intensity (5 x 4 bytes each = 80 bytes), LZF-coded by hand:
    literal run of the 60 coordinate bytes in two runs (32 + 28), then the intensity column 0,1,2,3,4 as a literal run of 20 bytes
    -- and, to exercise a back reference, the file carries a SECOND copy of the x column as a fifth field `x2`, coded as a back
    reference of length 20 at distance 80."""
    cols = [struct.pack("<5f", *[p[k] for p in XYZ]) for k in range(3)] + [struct.pack("<5f", *INTENSITY)]
    raw = b"".join(cols) + cols[0]                     # 100 bytes uncompressed
    coord = b"".join(cols[:3])                         # 60 bytes
    lzf = bytes([31]) + coord[:32] + bytes([27]) + coord[32:] + bytes([19]) + cols[3]
    # back reference: length 20 -> (20 - 2) = 18 >= 7: control = (7 << 5) | (off >> 8), extra length byte 18 - 7 = 11, low offset byte
    off = 80 - 1
    lzf += bytes([(7 << 5) | (off >> 8), 11, off & 0xff])
    hdr = pcd_header("binary_compressed", "x y z intensity x2", "4 4 4 4 4", "F F F F F", "1 1 1 1 1")
    return hdr + struct.pack("<II", len(lzf), len(raw)) + lzf


if __name__ == "__main__":
    w("kat_ascii.ply", ply_ascii())
    w("kat_le_faces_last.ply", ply_binary("little", False))
    w("kat_be_faces_first.ply", ply_binary("big", True))
    w("kat_double_be_crlf.ply", ply_double_be_crlf())
    w("kat_ascii.pcd", pcd_ascii())
    w("kat_binary.pcd", pcd_binary())
    w("kat_compressed.pcd", pcd_compressed())
