"""The C-ABI shared library loads and exports every symbol include/bufferx.h declares (no compute: CPU-only)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bufferx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|int64_t|const char \*|void)\s*\*?\s*(bx_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(n for n in names if n != "bx_chunk_slot"))


def test_header_symbols_exported():
    from bufferx_amd import lib
    lib.build()
    so = C.CDLL(lib._SO)
    decl = _declared()
    assert len(decl) >= 19, decl
    for name in decl:
        assert hasattr(so, name), f"{name} declared in include/bufferx.h but not exported"
    assert sorted(lib.EXPORTS) == decl


def test_struct_layouts_match_header():
    """ctypes mirrors == the C structs: sizes and the offset of EVERY field, taken from a C program compiled against include/bufferx.h."""
    import subprocess
    import tempfile
    from bufferx_amd import lib
    structs = {"bx_params": lib.BxParams, "bx_result": lib.BxResult, "bx_capture": lib.BxCapture, "bx_weights": lib.BxWeights}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "bufferx.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ["return 0; }"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "o.c"), os.path.join(d, "o")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, f)]) == getattr(cls, f).offset, (cname, f)
    # the arithmetic forms sit behind keypoint_tiles; bx_result echoes them in the former reserved word
    assert lib.BxParams.keypoint_tiles.offset == 168 and lib.BxParams.desc_conv_form.offset == 172 and lib.BxParams.cost_l0_form.offset == 180
    assert C.sizeof(lib.BxParams) == 184 and lib.BxResult.arith_forms.offset == 156


def test_arith_forms_table():
    """cfg.arith names <-> bx_params values: defaults are value 0, unknown names are refused, the header's constants agree."""
    import bufferx_amd
    from bufferx_amd import lib, config
    hdr = open(os.path.join(ROOT, "include", "bufferx.h")).read()
    for macro, key, name in (("BX_DESC_CONV_WINOGRAD43", "desc_conv", "winograd43"), ("BX_DESC_CONV_WINOGRAD22", "desc_conv", "winograd22"),
                             ("BX_DESC_CONV_DIRECT", "desc_conv", "direct"), ("BX_DESC_CONV_WINOGRAD43M", "desc_conv", "winograd43m"), ("BX_POSE_CONV_WINOGRAD22", "pose_conv", "winograd22"),
                             ("BX_POSE_CONV_DIRECT", "pose_conv", "direct"), ("BX_POSE_CONV_WINOGRAD43", "pose_conv", "winograd43"), ("BX_COST_L0_COLLAPSED", "cost_l0", "collapsed"),
                             ("BX_COST_L0_DIRECT", "cost_l0", "direct")):
        v = int(re.search(r"#define %s (\d+)" % macro, hdr).group(1))
        assert config.ARITH_FORMS[key][v] == name
    cfg = bufferx_amd.make_cfg("3DMatch")
    saved = dict(config.ARITH_DEFAULT)
    try:
        config.ARITH_DEFAULT.update({k: v[0] for k, v in config.ARITH_FORMS.items()})
        cfg = bufferx_amd.make_cfg("3DMatch")
        p = lib.params_from_cfg(cfg, 1000)
        assert (p.desc_conv_form, p.pose_conv_form, p.cost_l0_form) == (0, 0, 0)
        cfg.arith.desc_conv, cfg.arith.pose_conv, cfg.arith.cost_l0 = "direct", "winograd22", "direct"
        p = lib.params_from_cfg(cfg, 1000)
        assert (p.desc_conv_form, p.pose_conv_form, p.cost_l0_form) == (2, 1, 1)
        cfg.arith.pose_conv = "fast"
        try:
            lib.params_from_cfg(cfg, 1000)
            assert False, "unknown form accepted"
        except ValueError:
            pass
    finally:
        config.ARITH_DEFAULT.update(saved)


def test_keypoint_tile_bounds():
    """bx_keypoint_tile_bounds (pure host arithmetic of the latency form): tile 0 ends at the radius-estimation prefix rounded up
    to 4, the rest is split evenly in multiples of 4, empty tiles collapse, K <= nk + 4 is not tiled."""
    import bufferx_amd
    from bufferx_amd import lib
    so = lib.load()
    so.bx_keypoint_tile_bounds.restype = C.c_int
    b = (C.c_int32 * 9)()

    def bounds(K, nk, tiles):
        cfg = bufferx_amd.make_cfg("3DMatch")
        cfg.patch.num_fps, cfg.patch.num_points_radius_estimate = K, nk
        cfg.test.keypoint_tiles = tiles
        p = lib.params_from_cfg(cfg, 1000)
        T = so.bx_keypoint_tile_bounds(C.byref(p), b)
        return T, list(b[:T + 1])

    assert bounds(5000, 2000, 0) == (1, [0, 5000])
    assert bounds(5000, 2000, 2) == (2, [0, 2000, 5000])
    assert bounds(5000, 2000, 3) == (3, [0, 2000, 3500, 5000])
    assert bounds(5000, 2000, 5) == (5, [0, 2000, 2748, 3500, 4248, 5000])
    assert bounds(400, 96, 3) == (3, [0, 96, 248, 400])
    assert bounds(1500, 2000, 4) == (1, [0, 1500])          # reference default: nothing to tile
    assert bounds(2004, 2000, 4) == (1, [0, 2004])
    T, bb = bounds(2010, 1998, 8)
    assert bb[0] == 0 and bb[-1] == 2010 and all(x < y for x, y in zip(bb, bb[1:])) and bb[1] == 2000
    T, bb = bounds(4999, 2001, 4)
    assert bb[1] == 2004 and bb[-1] == 4999 and all((x % 4) == 0 for x in bb[:-1])


def test_error_path_without_gpu():
    """bx_create on a box without a GPU must return an error code and a message, never abort."""
    import torch
    from bufferx_amd import lib
    import bufferx_amd
    if torch.cuda.is_available():
        return
    so = lib.load()
    p = lib.params_from_cfg(bufferx_amd.make_cfg("3DMatch"), 1000)
    h = C.c_void_p()
    rc = so.bx_create(0, C.byref(p), C.byref(h))
    assert rc != 0 and len(so.bx_last_error()) > 0
    p.rad_n = 4
    assert so.bx_create(0, C.byref(p), C.byref(h)) == 1  # BX_ERR_ARG: geometry other than 3/7/20 is rejected
    p.rad_n, p.desc_conv_form = 3, 7
    assert so.bx_create(0, C.byref(p), C.byref(h)) == 1 and b"arithmetic form" in so.bx_last_error()


def test_slot_permutation_is_involution_free_bijection():
    import numpy as np
    from bufferx_amd import lib
    inv = lib.slot_perm()
    assert sorted(inv.tolist()) == list(range(16))
    x = np.arange(32, dtype=np.float32).reshape(2, 16)
    assert np.array_equal(lib.chunked_to_logical(lib.logical_to_chunked(x)), x)
    # slot 4*(c%4)+c/4 holds channel c
    ch = lib.logical_to_chunked(np.arange(16, dtype=np.float32)[None])[0]
    for c in range(16):
        assert ch[4 * (c % 4) + c // 4] == c


def test_header_is_plain_c_and_cxx():
    """include/bufferx.h is the drop-in boundary: it must compile on its own as C99 and as C++11 (no torch / HIP types in the
    signatures), warnings as errors."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "bufferx.h")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)          # declarations only (the comments cite the reference)
    assert "torch" not in code.lower() and "#include <hip" not in code and "at::" not in code
