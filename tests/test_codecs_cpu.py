"""Host codecs behind the C-ABI (csrc/io_native.cpp) against their Python / numpy statements, and the file conventions of the
libraries the reference reads its assets with (cv2, Open3D, pyredner) on hand-assembled files -- no encoder of this repo writes the inputs."""
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import ROOT


HOSTILE_OBJ = (
    "# exported by some scanner\r\n"
    "mtllib out1.mtl\r\n"
    "o room\r\n"
    "v 0 0 0\r\n"
    "v 1.5 0 0\r\n"
    "v +1.5 1e0 0\r\n"
    "v 0 1 -0.000000000000000000000000001\r\n"
    "  v 0.1234567890123456789 2.5E-3 -7.25e+2\r\n"          # leading blanks, long mantissa (slow path), exponents
    "v\t3\t4\t5 0.5 0.5 0.5\r\n"                              # tab separated, trailing vertex colours ignored
    "vt 0 0\r\n"
    "vt 1 0 0\r\n"                                            # 3-component vt: w ignored
    "vt 1 1\r\n"
    "vt 0.333333343 0.999999999999\r\n"
    "vn 0 0 1\r\n"
    "vn 0 0.6 0.8\r\n"
    "g wall\r\n"
    "usemtl material_0\r\n"
    "s off\r\n"
    "f 1/1/1 2/2/1 3/3/1 4/4/2\r\n"                           # quad -> fan
    "f -6//-1 -5//-2 -4//-1\r\n"                              # negative indices, v//vn
    "f 1 2 3 4 5\r\n"                                         # bare indices, pentagon -> 3 triangles
    "f 1/2 3/4 6/1\r\n"                                       # v/vt
    "v 9 9 9\r\n"                                             # vertices after faces: negative indices below count these too
    "f -1/-1 -2/-2 1/1\r\n"
    "l 1 2\r\n"
    "mtllib trailing.mtl"                                     # no newline at the end of the file
)


def _same_obj(a, b):
    assert set(a) == set(b)
    for k in a:
        if a[k] is None:
            assert b[k] is None, k
        else:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


def test_native_obj_parser_matches_python_on_a_hostile_file(tmp_path):
    from texir_code_amd import io_formats as IO
    p = tmp_path / "hostile.obj"
    p.write_bytes(HOSTILE_OBJ.encode())
    ref = IO.load_obj_py(str(p))
    out = IO.load_obj(str(p), cache=False)
    _same_obj(ref, out)
    assert out["indices"].shape == (8, 3)                                          # 2 + 1 + 3 + 1 + 1 triangles
    assert out["indices"][:2].tolist() == [[0, 1, 2], [0, 2, 3]]                     # quad fan
    assert out["indices"][2].tolist() == [0, 1, 2] and out["normal_indices"][2].tolist() == [1, 0, 1] and out["uv_indices"][2].tolist() == [-1, -1, -1]
    assert out["indices"][-1].tolist() == [6, 5, 0] and out["uv_indices"][-1].tolist() == [3, 2, 0]       # negatives count the late vertex
    assert out["vertices"][2].tolist() == [1.5, 1.0, 0.0] and out["vertices"][5].tolist() == [3.0, 4.0, 5.0]
    assert out["uvs"][1].tolist() == [1.0, 0.0]
    # LF and lone-CR line ends give the same arrays
    for nl in ("\n", "\r"):
        q = tmp_path / ("nl%d.obj" % ord(nl))
        q.write_bytes(HOSTILE_OBJ.replace("\r\n", nl).encode())
        _same_obj(ref, IO.load_obj(str(q), cache=False))
    # malformed lines are errors, not silently dropped geometry
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        IO.load_obj(str(bad), cache=False)


def test_native_obj_parser_matches_python_across_thread_chunks(tmp_path, monkeypatch):
    """a file large enough to be cut into several chunks: relative (negative) indices must resolve against the running counts of the
    whole file, whichever chunk a face lands in"""
    from texir_code_amd import io_formats as IO
    rng = np.random.default_rng(3)
    lines = []
    nv = nvt = 0
    for blk in range(6000):
        k = int(rng.integers(3, 7))
        for _ in range(k):
            lines.append("v %.9g %.9g %.9g" % tuple(rng.normal(size=3) * 10.0 ** int(rng.integers(-3, 4))))
            lines.append("vt %.9g %.9g" % tuple(rng.random(2)))
        nv += k
        nvt += k
        if blk % 2:
            lines.append("f " + " ".join("%d/%d" % (-(j + 1), -(j + 1)) for j in range(k)))
        else:
            lines.append("f " + " ".join("%d/%d" % (nv - j, nvt - j) for j in range(k)))
        lines.append("# pad " + "x" * int(rng.integers(0, 400)))
    p = tmp_path / "big.obj"
    p.write_text("\n".join(lines) + "\n")
    assert os.path.getsize(p) > 3 << 20                        # > 3 chunks of 1 MiB when threads are available
    ref = IO.load_obj_py(str(p))
    _same_obj(ref, IO.load_obj(str(p), cache=False))


def test_obj_cache_hands_out_one_parse_per_file(tmp_path):
    from texir_code_amd import io_formats as IO
    p = tmp_path / "m.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\n")
    a = IO.load_obj(str(p))
    b = IO.load_obj(str(p))
    assert a["vertices"] is b["vertices"] and not a["vertices"].flags.writeable
    p.write_text("v 0 0 0\nv 2 0 0\nv 0 2 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\n")
    os.utime(p, ns=(1, 10 ** 18))                             # (a rewritten file is a different cache key even within one clock tick)
    c = IO.load_obj(str(p))
    assert c["vertices"][1, 0] == 2.0


def test_native_rgbe_codec_is_byte_identical_to_numpy():
    from texir_code_amd import io_formats as IO
    rng = np.random.default_rng(5)
    rgb = (np.exp(rng.normal(size=(257, 131, 3)) * 4).astype(np.float32))
    rgb[0, 0] = 0
    rgb[0, 1] = (1e-33, 0, 0)                                  # below the 1e-32 cut: black
    rgb[0, 2] = (-1.0, 2.0, 0.5)                               # negative channel clips to 0
    rgb[0, 3] = (255.99999, 1.0, 1.0)
    rgb[0, 4] = (1.0, 1.0, 1.0)
    rgb[0, 5] = (3.4e38, 1.0, 0.0)
    rgb[0, 6] = (2.0 ** -100, 2.0 ** -101, 0.0)
    enc = IO.rgbe_encode(rgb)
    assert np.array_equal(enc, IO.rgbe_encode_py(rgb))
    allb = rng.integers(0, 256, size=(4096, 64, 4)).astype(np.uint8)
    allb[:256, 0, 3] = np.arange(256)                          # every exponent byte, incl. 0 (black) and 255
    allb[:256, 0, :3] = 255
    dec = IO.rgbe_decode(allb)
    assert dec.dtype == np.float32 and np.array_equal(dec, IO.rgbe_decode_py(allb))
    assert np.array_equal(IO.rgbe_decode(enc), IO.rgbe_decode_py(enc))


def _opencv_rle_bytes(data):
    """RGBE_WriteBytes_RLE of Radiance's rgbe.c as OpenCV's HdrEncoder uses it (modules/imgcodecs/src/rgbe.cpp): restated here so that the
    expected file bytes do not come from this repo's encoder"""
    out = bytearray()
    n, cur = len(data), 0
    while cur < n:
        beg = cur
        run = old = 0
        while run < 4 and beg < n:
            beg += run
            old = run
            run = 1
            while beg + run < n and run < 127 and data[beg] == data[beg + run]:
                run += 1
        if old > 1 and old == beg - cur:
            out += bytes([128 + old, data[cur]])
            cur = beg
        while cur < beg:
            k = min(beg - cur, 128)
            out += bytes([k]) + bytes(data[cur:cur + k])
            cur += k
        if run >= 4:
            out += bytes([128 + run, data[beg]])
            cur += run
    return bytes(out)


def _opencv_hdr_file(rgb):
    """the bytes cv2.imwrite(path, rgb[..., ::-1]) produces for a float32 image: "#?RADIANCE", FORMAT line, blank line, "-Y H +X W", then per
    scanline 2 2 hi lo + four RLE planes; pixel arithmetic float2rgbe (mantissa truncated, shared exponent of the largest channel)"""
    H, W, _ = rgb.shape
    body = bytearray()
    for y in range(H):
        planes = np.zeros((4, W), np.uint8)
        for x in range(W):
            r, g, b = (float(v) for v in rgb[y, x])
            v = max(r, g, b)
            if v >= 1e-32:
                m, e = np.frexp(np.float32(v))
                s = np.float32(np.float64(m) * 256.0 / np.float64(np.float32(v)))
                planes[:, x] = (int(np.float32(r) * s) if r > 0 else 0, int(np.float32(g) * s) if g > 0 else 0, int(np.float32(b) * s) if b > 0 else 0, int(e) + 128)
        body += bytes([2, 2, W >> 8, W & 255])
        for c in range(4):
            body += _opencv_rle_bytes(planes[c].tobytes())
    return b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + ("-Y %d +X %d\n" % (H, W)).encode() + bytes(body)


def test_hdr_file_is_laid_out_as_opencv_writes_it(tmp_path):
    from texir_code_amd import io_formats as IO
    rng = np.random.default_rng(11)
    H, W = 9, 300                                              # > 127 and > 128: run and literal length caps both bite
    rgb = np.exp(rng.normal(size=(H, W, 3))).astype(np.float32)
    rgb[1] = 0.75                                              # a whole scanline of one value: runs of 127 + 127 + 46
    rgb[2, :150] = (0.5, 0.25, 2.0)
    rgb[3, 10:13] = 0.0                                        # a 3-long run (below the minimum run length) inside noise
    rgb[4, ::2] = 1.0
    rgb[5] = np.repeat(rng.random((W // 2, 3)).astype(np.float32), 2, axis=0)            # runs of two everywhere
    expected = _opencv_hdr_file(rgb)
    p = str(tmp_path / "cv.hdr")
    with open(p, "wb") as f:
        f.write(expected)
    # reading the cv2-layout file: cv2.imread(path, -1)[:, :, ::-1] = mantissa * 2^(e - 136)
    back = IO.read_hdr(p)
    assert back.shape == (H, W, 3) and np.array_equal(back, IO.rgbe_decode_py(IO.rgbe_encode_py(rgb)))
    # writing: byte for byte the same file
    q = str(tmp_path / "ours.hdr")
    IO.write_hdr(q, rgb)
    assert open(q, "rb").read() == expected
    # narrow images are written flat by that writer (scanline width < 8)
    small = rgb[:3, :5].copy()
    IO.write_hdr(q, small)
    raw = open(q, "rb").read()
    assert raw.endswith(IO.rgbe_encode_py(small).tobytes()) and np.array_equal(IO.read_hdr(q), IO.rgbe_decode_py(IO.rgbe_encode_py(small)))


def _png_chunk(tag, payload):
    return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)


def test_index_texture_png_in_cv2_channel_order_feeds_the_reference_codes(tmp_path):
    """"0.png" is a 16-bit RGB PNG; cv2.imread(path, -1) hands the reference BGR, and tracer_o3d_irt.py:119-135 reads channel 2 as the panorama id,
    channel 1 as the column code, channel 0 as the row code -- i.e. the FILE stores (R, G, B) = (panorama id, column code, row code).  The PNG is
    assembled here byte by byte (big-endian samples, Sub-filtered rows, several IDAT chunks)."""
    from texir_code_amd import io_formats as IO
    from texir_code_amd.models import seam_texels
    H, W = 6, 5
    rng = np.random.default_rng(7)
    row_code = rng.integers(0, 50001, (H, W)).astype(np.uint16)
    col_code = rng.integers(0, 50001, (H, W)).astype(np.uint16)
    pano = rng.integers(0, 3, (H, W)).astype(np.uint16)
    row_code[0, 0] = col_code[0, 0] = pano[0, 0] = 0                                # a seam
    row_code[0, 1], col_code[0, 1], pano[0, 1] = 30000, 35530, 6                      # sums to 65536: wraps to 0 in the reference's uint16 sum
    row_code[0, 2], col_code[0, 2], pano[0, 2] = 50000, 50000, 1                      # the largest codes: clip to the last row / column
    file_rgb = np.stack([pano, col_code, row_code], -1)                              # what the asset pipeline stores
    raw = bytearray()
    for y in range(H):
        line = file_rgb[y].astype(">u2").tobytes()
        bpp = 6
        filt = bytes((line[i] - (line[i - bpp] if i >= bpp else 0)) & 255 for i in range(len(line)))      # filter type 1 (Sub)
        raw += b"\x01" + filt
    z = zlib.compress(bytes(raw), 9)
    png = (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 16, 2, 0, 0, 0)) + _png_chunk(b"tEXt", b"Software\x00hand")
           + _png_chunk(b"IDAT", z[:7]) + _png_chunk(b"IDAT", z[7:]) + _png_chunk(b"IEND", b""))
    p = str(tmp_path / "0.png")
    with open(p, "wb") as f:
        f.write(png)
    idx = IO.read_index_texture(p)
    assert idx.dtype == np.uint16 and idx.shape == (H, W, 3)
    assert np.array_equal(idx[..., 0], row_code) and np.array_equal(idx[..., 1], col_code) and np.array_equal(idx[..., 2], pano)
    # the reference's arithmetic on those channels (tracer_o3d_irt.py:130-138) for a 512 x 1024 panorama
    h, w = 512, 1024
    hdr_col = np.clip((idx[..., 1] / 50000 * w).astype(int), 0, w - 1)
    hdr_row = np.clip((idx[..., 0] / 50000 * h).astype(int), 0, h - 1)
    assert hdr_col[0, 2] == w - 1 and hdr_row[0, 2] == h - 1
    seam_ref = (idx[:, :, 0] + idx[:, :, 1] + idx[:, :, 2]) == 0                      # uint16 arithmetic, as the reference evaluates it
    assert seam_ref[0, 0] and seam_ref[0, 1] and seam_ref.sum() == 2
    assert np.array_equal(seam_texels(idx), seam_ref)


def test_scalar_log_and_phase_timer(tmp_path):
    from texir_code_amd.runlog import PhaseTimer, ScalarLog, read_scalars
    w = ScalarLog(str(tmp_path))
    for it in range(5):
        w.add_scalar("img_loss_L1_stage0", 1.0 / (it + 1), it)
        w.add_scalar("seg_loss_L1_stage0", 0.5, it)
    w.close()
    s = read_scalars(str(tmp_path / "scalars.jsonl"))
    assert [x[0] for x in s["img_loss_L1_stage0"]] == [0, 1, 2, 3, 4] and s["seg_loss_L1_stage0"][0][1] == 0.5
    ScalarLog(None).add_scalar("x", 1.0, 0)                                           # no directory: a no-op, not an error
    t = PhaseTimer()
    t.reset(True)
    with t.phase("a", sync=False):
        pass
    with t.phase("a", sync=False):
        pass
    with t.phase("b", sync=False):
        pass
    assert list(t.report()) == ["a", "b"]
    t.reset(False)
    with t.phase("c"):
        pass
    assert t.report() == {}


def test_four_byte_texels_hold_every_rgbe_born_value_exactly():
    """texture layouts 3 / 4 (include/texir_hip.h texir_texel_pack): a texel of an RGBE file times 2^hdr_exposure (tracer_o3d_irt.py:77-81) is three 8-bit
    integers times one power of two; the 4-byte word decodes to the IDENTICAL float32 triple.  Anything else is flagged and keeps float32 texels."""
    from texir_code_amd import _lib
    from texir_code_amd.io_formats import rgbe_decode_py
    L = _lib.lib()
    rng = np.random.default_rng(3)
    rgbe = rng.integers(0, 256, (200000, 4), dtype=np.uint8)
    rgbe[:, 3] = rng.integers(40, 220, 200000)          # exponents whose values times 2^+-5 stay normal floats
    rgbe[:1000, :3] = 0                                 # black texels
    rgbe[1000:2000, 3] = 0                              # e = 0: cv2 decodes them to 0
    rgbe[2000:2100] = [255, 255, 255, 219]
    rgbe[2100:2200] = [1, 0, 0, 40]
    for exposure in (5.0, 3.0, 0.0, -2.0):
        rgb = np.ascontiguousarray(rgbe_decode_py(rgbe) * np.float32(2.0 ** exposure), np.float32)
        words, ok, back = np.zeros(len(rgb), np.uint32), np.zeros(len(rgb), np.uint8), np.zeros_like(rgb)
        _lib.check(L.texir_texel_pack(_lib.ptr(rgb), len(rgb), _lib.ptr(words), _lib.ptr(ok)))
        assert ok.all()
        _lib.check(L.texir_texel_unpack(_lib.ptr(words), len(rgb), _lib.ptr(back)))
        assert np.array_equal(back.view(np.uint32), rgb.view(np.uint32))
        assert (words[:2000] == 0).all()
    # what does not have the form: a ninth mantissa bit, two exponents too far apart, negative, -0.0, subnormal, inf, nan, a non-power-of-two exposure
    bad = np.array([[257.0, 1.0, 1.0], [1.0, 2.0 ** -9, 0.0], [-1.0, 0.0, 0.0], [-0.0, 1.0, 1.0], [1e-45, 0.0, 0.0], [np.inf, 0.0, 0.0], [np.nan, 0.0, 0.0],
                    [0.3, 0.3, 0.3], [2.0 ** 128 * 0 + 3.0e38, 1.0, 0.0]], np.float32)
    ok = np.ones(len(bad), np.uint8)
    words = np.full(len(bad), 7, np.uint32)
    _lib.check(L.texir_texel_pack(_lib.ptr(bad), len(bad), _lib.ptr(words), _lib.ptr(ok)))
    assert not ok.any() and (words == 0).all()
    frac = np.ascontiguousarray(rgbe_decode_py(rgbe[3000:4000]) * np.float32(2.0 ** 2.5), np.float32)
    ok = np.ones(len(frac), np.uint8)
    _lib.check(L.texir_texel_pack(_lib.ptr(frac), len(frac), _lib.ptr(np.zeros(len(frac), np.uint32)), _lib.ptr(ok)))
    assert ok.mean() < 0.05
    # values that are exact but not RGBE-born: (255, 1, 0) * 2^-20, a lone 2^100
    good = np.array([[255 * 2.0 ** -20, 2.0 ** -20, 0.0], [2.0 ** 100, 0.0, 0.0], [0.0, 0.0, 0.0], [96.0, 0.0, 160.0]], np.float32)
    words, ok, back = np.zeros(4, np.uint32), np.zeros(4, np.uint8), np.zeros_like(good)
    _lib.check(L.texir_texel_pack(_lib.ptr(good), 4, _lib.ptr(words), _lib.ptr(ok)))
    _lib.check(L.texir_texel_unpack(_lib.ptr(words), 4, _lib.ptr(back)))
    assert ok.all() and np.array_equal(back, good)
    # the synthetic scenes' texture, passed through the reference's file format, has the form everywhere
    from texir_code_amd import synth
    sc0 = synth.make_scene(2000, tex_res=128)
    born = np.ascontiguousarray(synth.rgbe_born(sc0["hdr"]).reshape(-1, 3))
    ok = np.zeros(len(born), np.uint8)
    _lib.check(L.texir_texel_pack(_lib.ptr(born), len(born), _lib.ptr(np.zeros(len(born), np.uint32)), _lib.ptr(ok)))
    assert ok.all() and np.abs(born.reshape(sc0["hdr"].shape) - sc0["hdr"]).max() <= sc0["hdr"].max() / 128


def test_obj_numbers_on_the_slow_path_follow_python_float(tmp_path):
    """the OBJ parser's slow path (mantissas past 19 digits, |exponent| > 22) goes through strtod -- in the "C" locale, and without the hexadecimal
    spellings strtod takes but Python's float() (the stated reference, io_formats.load_obj_py) refuses"""
    import locale
    from texir_code_amd import io_formats as IO
    long_m = "0.1234567890123456789012345"          # 25 digits
    txt = "v %s 1e-30 -2.5E+25\nv 1 0 0\nv 0 1 0\nf 1 2 3\n" % long_m
    p = tmp_path / "slow.obj"
    p.write_text(txt)
    out = IO.load_obj(str(p), cache=False)
    assert out["vertices"][0].tolist() == [np.float32(float(long_m)), np.float32(1e-30), np.float32(-2.5e25)]
    for spelling in ("0x1p3", "0X10"):
        bad = tmp_path / "hex.obj"
        bad.write_text("v %s 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n" % spelling)
        with pytest.raises(ValueError):
            IO.load_obj(str(bad), cache=False)
        with pytest.raises(ValueError):
            float(spelling)
    # a comma-decimal LC_NUMERIC of the embedding process must not change the parse (only testable where such a locale is installed)
    old = locale.setlocale(locale.LC_NUMERIC)
    try:
        for name in ("de_DE.UTF-8", "de_DE.utf8", "fr_FR.UTF-8"):
            try:
                locale.setlocale(locale.LC_NUMERIC, name)
            except locale.Error:
                continue
            assert IO.load_obj(str(p), cache=False)["vertices"][0].tolist() == out["vertices"][0].tolist()
            break
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)
