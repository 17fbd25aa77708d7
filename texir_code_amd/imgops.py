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
