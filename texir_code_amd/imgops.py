"""The handful of OpenCV image operations the reference's dataset adapters use (datasets/dataset.py:471-541, 789-884),
restated in numpy from OpenCV's documented semantics (cv2 is not available on ROCm images; parity unpinned)."""
import numpy as np


def gray(img, order="RGB"):
    """cv2.cvtColor(., COLOR_RGB2GRAY / COLOR_BGR2GRAY) for float images: 0.299 R + 0.587 G + 0.114 B"""
    img = np.asarray(img, np.float32)
    r, g, b = (img[..., 0], img[..., 1], img[..., 2]) if order == "RGB" else (img[..., 2], img[..., 1], img[..., 0])
    return (np.float32(0.299) * r + np.float32(0.587) * g + np.float32(0.114) * b).astype(np.float32)


def sobel3(a):
    """(cv2.Sobel(a, CV_32F, 1, 0, ksize=3), cv2.Sobel(a, CV_32F, 0, 1, ksize=3)), border BORDER_REFLECT_101"""
    p = np.pad(np.asarray(a, np.float32), 1, mode="reflect")
    sm_y = p[:-2, :] + 2 * p[1:-1, :] + p[2:, :]          # smooth along y
    sm_x = p[:, :-2] + 2 * p[:, 1:-1] + p[:, 2:]          # smooth along x
    gx = sm_y[:, 2:] - sm_y[:, :-2]
    gy = sm_x[2:, :] - sm_x[:-2, :]
    return gx.astype(np.float32), gy.astype(np.float32)


def magnitude(x, y):
    return np.sqrt(x * x + y * y).astype(np.float32)


def erode(a, k=5):
    """cv2.erode(a, np.ones((k,k))): minimum over the k x k window, pixels outside the image ignored"""
    a = np.asarray(a)
    squeeze = a.ndim == 3 and a.shape[2] == 1
    if squeeze:
        a = a[..., 0]                                     # (cv2 drops a trailing singleton channel, dataset.py:499 comment)
    r = k // 2
    top = np.iinfo(a.dtype).max if np.issubdtype(a.dtype, np.integer) else np.inf
    p = np.pad(a, r, mode="constant", constant_values=top)
    out = a.copy()
    h, w = a.shape[:2]
    for dy in range(k):
        for dx in range(k):
            out = np.minimum(out, p[dy:dy + h, dx:dx + w])
    return out


def resize_nearest(a, size):
    """cv2.resize(a, (w, h), interpolation=INTER_NEAREST): src index = floor(dst * src/dst)"""
    w, h = size
    a = np.asarray(a)
    ys = np.minimum((np.arange(h) * (a.shape[0] / h)).astype(np.int64), a.shape[0] - 1)
    xs = np.minimum((np.arange(w) * (a.shape[1] / w)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


def resize_linear(a, size):
    """cv2.resize(a, (w, h)) (INTER_LINEAR, half-pixel centres, edge replicate); 8-bit inputs are rounded back to 8 bits
    (OpenCV's fixed-point path can differ from this float evaluation by one code value)"""
    w, h = size
    a = np.asarray(a)
    src = a.astype(np.float32)

    def axis(n_dst, n_src):
        x = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        x0 = np.floor(x).astype(np.int64)
        f = (x - x0).astype(np.float32)
        return np.clip(x0, 0, n_src - 1), np.clip(x0 + 1, 0, n_src - 1), f

    y0, y1, fy = axis(h, a.shape[0])
    x0, x1, fx = axis(w, a.shape[1])
    fy = fy.reshape(-1, 1, *([1] * (a.ndim - 2)))
    fx = fx.reshape(1, -1, *([1] * (a.ndim - 2)))
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    out = top * (1 - fy) + bot * fy
    if np.issubdtype(a.dtype, np.integer):
        out = np.clip(np.rint(out), np.iinfo(a.dtype).min, np.iinfo(a.dtype).max).astype(a.dtype)
    return out


def resize_u16_as_cv2_default(a, size):
    """what `cv2.resize(a, (w, h), cv2.INTER_NEAREST)` actually computes for a uint16 image: the third POSITIONAL argument of cv2.resize is `dst`, so the
    flag is ignored and the interpolation is the default INTER_LINEAR (/root/reference/models/tracer_o3d_irt.py:95 resizes the index texture's row / column /
    panorama codes this way).  Restated from OpenCV 4.x modules/imgproc/src/resize.cpp (not executable in this image: parity with cv2 itself is unpinned,
    the arithmetic below is pinned by a hand-computed vector, tests/test_host_cpu.py):
      * same size: a copy;
      * an exact 2 x 2 reduction (src = 2 x dst on both axes) is re-routed to INTER_AREA's fast path (`if (interpolation == INTER_LINEAR && is_area_fast &&
        iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA`): (a + b + c + d + 2) >> 2 in integers (ResizeAreaFastVec<ushort>);
      * everything else: CV_16U has no fixed-point path (the 11-bit INTER_RESIZE_COEF table is the 8-bit case only) -- float32 taps,
        fx = float((dx + 0.5) * scale - 0.5) with scale = 1 / (dst / src) in double, sx = floor(fx), weights (1 - fx, fx) as float32; sx < 0 -> (sx, fx) = (0, 0);
        sx >= src - 1 -> (src - 1, 0); rows are clamped (border replicate) with the weights kept; horizontal pass first into float32 rows, then the vertical
        pass, then saturate_cast<ushort> = round half to even (cvRound) and clamp to [0, 65535]."""
    w, h = int(size[0]), int(size[1])
    a = np.asarray(a)
    if a.dtype != np.uint16:
        raise TypeError("resize_u16_as_cv2_default: uint16 input expected, got %s" % a.dtype)
    H, W = a.shape[:2]
    if (H, W) == (h, w):
        return a.copy()
    if H == 2 * h and W == 2 * w:
        s = a.astype(np.uint32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint16)

    def taps(n_dst, n_src, clamp_weights):
        scale = 1.0 / (float(n_dst) / float(n_src))
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = (f - s0.astype(np.float32)).astype(np.float32)
        if clamp_weights:                       # the x axis: out-of-range taps get weight 0
            lo, hi = s0 < 0, s0 >= n_src - 1
            f = np.where(lo | hi, np.float32(0), f)
            s0 = np.where(lo, 0, np.where(hi, n_src - 1, s0))
        return np.clip(s0, 0, n_src - 1), np.clip(s0 + 1, 0, n_src - 1), (np.float32(1) - f).astype(np.float32), f

    x0, x1, ax0, ax1 = taps(w, W, True)
    y0, y1, by0, by1 = taps(h, H, False)
    src = a.astype(np.float32)
    tail = (1,) * (a.ndim - 2)
    rows = src[:, x0] * ax0.reshape(1, -1, *tail) + src[:, x1] * ax1.reshape(1, -1, *tail)             # [H, w, ...] float32
    out = rows[y0] * by0.reshape(-1, 1, *tail) + rows[y1] * by1.reshape(-1, 1, *tail)
    return np.clip(np.rint(out), 0, 65535).astype(np.uint16)
