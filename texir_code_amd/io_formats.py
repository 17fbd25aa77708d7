"""On-disk formats around the hot path, replacing cv2 / pyredner / Open3D loaders:
Radiance .hdr (RGBE, flat + new-style RLE), 8/16-bit PNG (index texture "0.png", tracer_o3d_irt.py:91), Wavefront OBJ with
the conventions the reference relies on (Open3D per-corner triangle_uvs un-flipped, tracer_o3d_irt.py:85; pyredner uvs
with V flipped)."""
import struct
import zlib

import numpy as np


# ------------------------------------------------------------------------------------------------- Radiance RGBE
def rgbe_encode_py(rgb):
    """numpy statement of the RGBE pixel encode (the test reference of texir_rgbe_encode)"""
    rgb = np.asarray(rgb, np.float32)
    mx = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    ok = mx > 1e-32
    m, e = np.frexp(mx[ok])
    scale = (m * 256.0 / mx[ok]).astype(np.float32)
    out[ok, 0:3] = np.clip(rgb[ok] * scale[:, None], 0, 255).astype(np.uint8)
    out[ok, 3] = (e + 128).astype(np.uint8)
    return out


def rgbe_decode_py(rgbe):
    """numpy statement of the RGBE pixel decode (the test reference of texir_rgbe_decode)"""
    e = rgbe[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(1.0, e - (128 + 8)), 0.0).astype(np.float32)
    return rgbe[..., 0:3].astype(np.float32) * f[..., None]


def rgbe_encode(rgb):
    """float RGB [...,3] -> RGBE bytes [...,4] through the library's threaded loop (texir_rgbe_encode, csrc/io_native.cpp)"""
    from . import _lib
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.empty(rgb.shape[:-1] + (4,), np.uint8)
    _lib.check(_lib.lib().texir_rgbe_encode(_lib.ptr(rgb), rgb.size // 3, _lib.ptr(out)))
    return out


def rgbe_decode(rgbe):
    """RGBE bytes [...,4] -> float32 RGB [...,3] (texir_rgbe_decode)"""
    from . import _lib
    rgbe = np.ascontiguousarray(rgbe, np.uint8)
    out = np.empty(rgbe.shape[:-1] + (3,), np.float32)
    _lib.check(_lib.lib().texir_rgbe_decode(_lib.ptr(rgbe), rgbe.size // 4, _lib.ptr(out)))
    return out


def write_hdr(path, rgb, rle=True, scratch=None):
    """rgb [H,W,3] float32, RGB order, row 0 = top.  The file cv2.imwrite(path, rgb[..., ::-1]) writes (trainer/generate_ir_texture.py:82):
    header "#?RADIANCE / FORMAT=32-bit_rle_rgbe / -Y H +X W", new-style RLE scanlines (cv2's default); rle=False writes flat scanlines.
    scratch: a dict a repeated caller passes to keep the two encode buffers between calls (plot_writer: allocating and releasing 2 x 67 MB per
    file means page faults and munmaps of that size under the interpreter lock, which the training loop on the main thread then waits for)."""
    from . import _lib
    rgb = np.ascontiguousarray(rgb, np.float32)
    H, W, _ = rgb.shape
    body = None if scratch is None else scratch.get(("rgbe", H, W))
    if body is None:
        body = np.empty((H, W, 4), np.uint8)
        if scratch is not None:
            scratch[("rgbe", H, W)] = body
    _lib.check(_lib.lib().texir_rgbe_encode(_lib.ptr(rgb), rgb.size // 3, _lib.ptr(body)))
    if rle:
        cap = H * (4 + 4 * (W + W // 64 + 4)) + 16
        buf = None if scratch is None else scratch.get(("rle", cap))
        if buf is None:
            buf = np.empty(cap, np.uint8)
            if scratch is not None:
                scratch[("rle", cap)] = buf
        n = _lib.lib().texir_hdr_encode_rle(_lib.ptr(body), W, H, _lib.ptr(buf), cap)
        if n < 0:
            raise ValueError("%s: RLE encode failed" % path)
        body = buf[:n]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(("-Y %d +X %d\n" % (H, W)).encode())
        f.write(memoryview(body).cast("B") if body.flags.c_contiguous else body.tobytes())


def read_hdr(path):
    """-> [H,W,3] float32 in RGB order (cv2.imread(...,-1)[:,:,::-1] of the reference), row 0 = top"""
    with open(path, "rb") as f:
        data = f.read()
    pos = data.index(b"\n\n") + 2 if b"\n\n" in data[:4096] else None
    if pos is None or not data.startswith(b"#?"):
        raise ValueError("%s: not a Radiance HDR file" % path)
    end = data.index(b"\n", pos)
    dims = data[pos:end].split()
    if len(dims) != 4 or dims[0] != b"-Y" or dims[2] != b"+X":
        raise ValueError("%s: unsupported resolution line %r" % (path, data[pos:end]))
    H, W = int(dims[1]), int(dims[3])
    buf = np.frombuffer(data, np.uint8, offset=end + 1)
    if buf.size == H * W * 4 and not (W >= 8 and W < 32768 and buf[0] == 2 and buf[1] == 2):
        return rgbe_decode(buf.reshape(H, W, 4))
    # RLE (or mixed) scanlines: a byte-serial recurrence -> the library's C++ loop (texir_hdr_decode_scanlines)
    from . import _lib
    out = np.empty((H, W, 4), np.uint8)
    src = np.ascontiguousarray(buf)
    used = _lib.lib().texir_hdr_decode_scanlines(_lib.ptr(src), int(src.size), W, H, _lib.ptr(out))
    if used < 0:
        raise ValueError("%s: corrupt Radiance scanline data" % path)
    return rgbe_decode(out)


# ------------------------------------------------------------------------------------------------- PNG (8/16 bit)
def write_png(path, img):
    """img [H,W,C] uint8 or uint16, C in {1,3,4}, channel order as stored (RGB)"""
    img = np.ascontiguousarray(img)
    if img.ndim == 2:
        img = img[..., None]
    H, W, C = img.shape
    depth = 16 if img.dtype == np.uint16 else 8
    ctype = {1: 0, 3: 2, 4: 6}[C]
    raw = img.astype(">u2" if depth == 16 else np.uint8).tobytes()
    stride = W * C * depth // 8
    lines = b"".join(b"\x00" + raw[y * stride:(y + 1) * stride] for y in range(H))

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(lines, 3)) + chunk(b"IEND", b""))


def read_png(path):
    """-> [H,W,C] uint8/uint16 in the file's channel order (RGB[A]); non-interlaced, colour types 0/2/6 at depth 8/16 and 8-bit palette
    images (3, expanded to RGB); all five scanline filters"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: not a PNG" % path)
    p, idat, hdr, plte = 8, [], None, None
    while p < len(data):
        n, tag = struct.unpack(">I4s", data[p:p + 8])
        body = data[p + 8:p + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            plte = body
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        p += 12 + n
    W, H, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 3, 6) or (ctype == 3 and depth != 8):
        raise ValueError("%s: unsupported PNG flavour (depth %d, colour type %d, interlace %d)" % (path, depth, ctype, interlace))
    C = {0: 1, 2: 3, 3: 1, 6: 4}[ctype]
    bpp = C * depth // 8
    stride = W * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    if raw.size != H * (stride + 1):
        raise ValueError("%s: truncated PNG data" % path)
    if not raw.reshape(H, stride + 1)[:, 0].any():
        out = np.ascontiguousarray(raw.reshape(H, stride + 1)[:, 1:])        # every scanline unfiltered (our own writer)
    else:
        # Sub / Up / Average / Paeth are byte-serial recurrences: the library's C++ loop (texir_png_unfilter)
        from . import _lib
        out = np.empty((H, stride), np.uint8)
        if _lib.lib().texir_png_unfilter(_lib.ptr(np.ascontiguousarray(raw)), H, stride, bpp, _lib.ptr(out)) != 0:
            raise ValueError("%s: corrupt PNG scanline data" % path)
    if ctype == 3:
        # palette image -> RGB, as cv2.imread's colour conversion delivers it
        if plte is None:
            raise ValueError("%s: palette PNG without PLTE chunk" % path)
        pal = np.frombuffer(plte, np.uint8).reshape(-1, 3)
        return pal[out.reshape(H, W)]
    if depth == 16:
        return out.view(">u2").astype(np.uint16).reshape(H, W, C)
    return out.reshape(H, W, C)


def read_index_texture(path):
    """"0.png" as cv2.imread(path, -1) returns it (BGR channel order): ch0 = row code, ch1 = col code, ch2 = panorama id
    (models/tracer_o3d_irt.py:91,119-135)."""
    img = read_png(path)
    return np.ascontiguousarray(img[..., :3][..., ::-1])


# ------------------------------------------------------------------------------------------------- Wavefront OBJ
def load_obj_py(path):
    """pure-Python statement of the OBJ reader (the test reference of texir_obj_parse): lines are classified by their first token."""
    v, vt, vn, fi, ft, fn = [], [], [], [], [], []
    with open(path, "r", newline=None) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                v.append([float(x) for x in tok[1:4]])
            elif tok[0] == "vt":
                vt.append([float(x) for x in tok[1:3]])
            elif tok[0] == "vn":
                vn.append([float(x) for x in tok[1:4]])
            elif tok[0] == "f":
                corners = []
                for t in tok[1:]:
                    parts = t.split("/")
                    a = int(parts[0]); a = a - 1 if a > 0 else len(v) + a
                    b = -1
                    if len(parts) > 1 and parts[1]:
                        b = int(parts[1]); b = b - 1 if b > 0 else len(vt) + b
                    c = -1
                    if len(parts) > 2 and parts[2]:
                        c = int(parts[2]); c = c - 1 if c > 0 else len(vn) + c
                    corners.append((a, b, c))
                for k in range(1, len(corners) - 1):
                    tri = (corners[0], corners[k], corners[k + 1])
                    fi.append([t[0] for t in tri]); ft.append([t[1] for t in tri]); fn.append([t[2] for t in tri])
    return {"vertices": np.asarray(v, np.float32).reshape(-1, 3), "indices": np.asarray(fi, np.int32).reshape(-1, 3),
            "uvs": np.asarray(vt, np.float32).reshape(-1, 2), "uv_indices": np.asarray(ft, np.int32).reshape(-1, 3),
            "normals": np.asarray(vn, np.float32).reshape(-1, 3) if vn else None,
            "normal_indices": np.asarray(fn, np.int32).reshape(-1, 3) if vn else None}


_OBJ_CACHE = {}


def load_obj(path, cache=True):
    """-> dict(vertices [V,3], indices [T,3], uvs [Vt,2] (as written in the file), uv_indices [T,3], normals [Vn,3] | None,
    normal_indices [T,3] | None).  Faces with more than 3 corners are fan-triangulated; negative indices resolved.
    Parsed by the library (texir_obj_parse: threaded C++, 1 M triangles in well under a second).  One run of a runner opens the same
    mesh from the dataset AND from the model (datasets.py, models.py -- as the reference does, datasets/dataset.py:385 and
    models/tracer_o3d_irt.py:75): the parsed arrays are kept per (path, size, mtime) and handed out read-only."""
    import ctypes as C
    import os
    from . import _lib
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_size, st.st_mtime_ns)
    if cache and key in _OBJ_CACHE:
        return dict(_OBJ_CACHE[key])
    with open(path, "rb") as f:
        text = f.read()
    L = _lib.lib()
    h = C.c_void_p()
    counts = np.zeros(4, np.int64)
    if L.texir_obj_parse(text, len(text), C.byref(h), _lib.ptr(counts)) != 0:
        raise ValueError("%s: malformed OBJ (a v / vt / vn / f line could not be parsed)" % path)
    nv, nvt, nvn, nt = (int(x) for x in counts)
    v, vt = np.empty((nv, 3), np.float32), np.empty((nvt, 2), np.float32)
    vn = np.empty((nvn, 3), np.float32) if nvn else None
    fi, ft = np.empty((nt, 3), np.int32), np.empty((nt, 3), np.int32)
    fn = np.empty((nt, 3), np.int32) if nvn else None
    _lib.check(L.texir_obj_take(h, _lib.ptr(v), _lib.ptr(vt), _lib.ptr(vn), _lib.ptr(fi), _lib.ptr(ft), _lib.ptr(fn)))
    out = {"vertices": v, "indices": fi, "uvs": vt, "uv_indices": ft, "normals": vn, "normal_indices": fn}
    for a in out.values():
        if a is not None:
            a.setflags(write=False)
    if cache:
        _OBJ_CACHE.clear()                      # one mesh per run: keep the latest only
        _OBJ_CACHE[key] = out
    return dict(out)


def triangle_uvs_open3d(obj):
    """np.asarray(o3d.io.read_triangle_mesh(path).triangle_uvs): per-corner [3T,2], V NOT flipped (tracer_o3d_irt.py:85)"""
    if obj["uvs"].shape[0] == 0:
        raise ValueError("mesh has no texture coordinates")
    return obj["uvs"][obj["uv_indices"].reshape(-1)].astype(np.float32)


def corner_normals(obj):
    """[3T,3] shading normals per corner; geometric normals when the file has no vn"""
    if obj["normals"] is not None and (obj["normal_indices"] >= 0).all():
        return obj["normals"][obj["normal_indices"].reshape(-1)].astype(np.float32)
    vtx, idx = obj["vertices"], obj["indices"]
    n = np.cross(vtx[idx[:, 1]] - vtx[idx[:, 0]], vtx[idx[:, 2]] - vtx[idx[:, 0]])
    n = n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-20)
    return np.repeat(n, 3, axis=0).astype(np.float32)


def write_obj(path, vertices, indices, tri_uvs, tri_normals=None):
    """per-corner uvs [3T,2] (and optional per-corner normals [3T,3]) written un-indexed"""
    T = indices.shape[0]
    with open(path, "w") as f:
        f.write("mtllib out1.mtl\nusemtl material_0\n")
        for p in vertices:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        for t in tri_uvs:
            f.write("vt %.9g %.9g\n" % tuple(t))
        if tri_normals is not None:
            for n in tri_normals:
                f.write("vn %.9g %.9g %.9g\n" % tuple(n))
        for i in range(T):
            a, b, c = indices[i] + 1
            k = 3 * i + 1
            if tri_normals is not None:
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, k, k, b, k + 1, k + 1, c, k + 2, k + 2))
            else:
                f.write("f %d/%d %d/%d %d/%d\n" % (a, k, b, k + 1, c, k + 2))


def obj_texture_path(path_obj):
    """the diffuse texture map an OBJ refers to: last `map_Kd` of its `mtllib` (Open3D: trianglemesh.textures, used by
    models/tracer_o3d_irrf.py:59 when the scene has no HDR texture).  None if the OBJ names no material library/map."""
    import os
    mtl = None
    with open(path_obj, "r", errors="replace") as f:
        for line in f:
            if line.startswith("mtllib"):
                mtl = line.split(None, 1)[1].strip()
                break
    if mtl is None:
        return None
    mtl = os.path.join(os.path.dirname(path_obj), mtl)
    if not os.path.exists(mtl):
        return None
    tex = None
    with open(mtl, "r", errors="replace") as f:
        for line in f:
            t = line.strip()
            if t.startswith("map_Kd"):
                tex = t.split()[-1]
    return os.path.join(os.path.dirname(mtl), tex) if tex else None


def read_ldr(path):
    """8/16-bit image -> [H,W,3] uint array, RGB.  PNG through the in-tree decoder; other formats need Pillow."""
    if path.lower().endswith(".png"):
        return read_png(path)[..., :3]
    try:
        from PIL import Image
    except ImportError as e:
        raise ValueError("%s: only PNG textures can be decoded without Pillow" % path) from e
    import numpy as np
    return np.asarray(Image.open(path).convert("RGB"))
