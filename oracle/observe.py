"""CPU restatement of the observation stage between pyflex.render and prepare_image.  TEST
INFRASTRUCTURE ONLY: imported by tests/ and never by the product.

Reference:  get_image (environment/flex_utils.py:418-427): flip rows, drop alpha, cv2.resize (default INTER_LINEAR) of the
            720 x 720 render to image_dim;
            SimEnv.get_cloth_mask (environment/simEnv.py:699-708): cv2.cvtColor(RGB2HSV) -> cv2.inRange((0,0,0),(100,100,100))
            -> mask == 0 -> get_largest_component (environment/utils.py:585-601, skimage.measure.label, full connectivity);
            SimEnv.get_obs (simEnv.py:710-737): minimum centred square crop around the mask, x 1.5, / image_dim;
            preprocess_obs (environment/utils.py:579-582).

PARITY UNPINNED: cv2 and skimage are absent from this image, so no vector of the reference's own calls exists.  The
functions below restate the documented scalar algorithms of OpenCV (imgproc/src/resize.cpp: HResizeLinear / VResizeLinear
with INTER_RESIZE_COEF_BITS = 11 for 8-bit images, plain float arithmetic for 32-bit ones; imgproc/src/color_hsv.cpp
RGB2HSV_b with hsv_shift = 12 and the sdiv / hdiv tables) and skimage's labelling semantics (8-connectivity; labels in raster
order of each component's first pixel), which OpenCV's vector paths are built to reproduce bit for bit."""
import numpy as np
from scipy import ndimage

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _linear_taps(dst, src):
    """sx[dst], fx[dst] of cv::resize INTER_LINEAR: fx = (float)((d + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx, clamped."""
    scale = np.float64(src) / np.float64(dst)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    return s, f


def _round_half_even(x):
    return np.rint(x)  # cvRound: lrint in the default rounding mode


def resize_linear_u8(img, dim):
    """img uint8 [H, W, C] -> [dim, dim, C]."""
    h, w = img.shape[:2]
    if (h, w) == (dim, dim):
        return img.copy()
    sx, fx = _linear_taps(dim, w)
    sy, fy = _linear_taps(dim, h)
    ax0 = _round_half_even((np.float32(1.0) - fx) * np.float32(COEF_SCALE)).astype(np.int64)
    ax1 = _round_half_even(fx * np.float32(COEF_SCALE)).astype(np.int64)
    by0 = _round_half_even((np.float32(1.0) - fy) * np.float32(COEF_SCALE)).astype(np.int64)
    by1 = _round_half_even(fy * np.float32(COEF_SCALE)).astype(np.int64)
    src = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, w - 1)
    sy1 = np.minimum(sy + 1, h - 1)
    rows = src[:, sx] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]      # [H, dim, C], scale 2^11
    s0, s1 = rows[sy], rows[sy1]
    out = (((by0[:, None, None] * (s0 >> 4)) >> 16) + ((by1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_linear_f32(img, dim):
    """img float32 [H, W] -> [dim, dim] (float taps, row pass then column pass, no fused multiply-add)."""
    h, w = img.shape
    if (h, w) == (dim, dim):
        return img.copy()
    sx, fx = _linear_taps(dim, w)
    sy, fy = _linear_taps(dim, h)
    sx1 = np.minimum(sx + 1, w - 1)
    sy1 = np.minimum(sy + 1, h - 1)
    a0, a1 = (np.float32(1.0) - fx), fx
    b0, b1 = (np.float32(1.0) - fy), fy
    rows = (img[:, sx] * a0[None, :]).astype(np.float32) + (img[:, sx1] * a1[None, :]).astype(np.float32)
    rows = rows.astype(np.float32)
    out = (rows[sy] * b0[:, None]).astype(np.float32) + (rows[sy1] * b1[:, None]).astype(np.float32)
    return out.astype(np.float32)


def _div_table(num, den_mul, n=256, shift=12):
    t = np.zeros(n, np.int64)
    i = np.arange(1, n)
    t[1:] = np.rint((num << shift) / (den_mul * i.astype(np.float64))).astype(np.int64)
    return t


SDIV = _div_table(255, 1.0)
HDIV180 = _div_table(180, 6.0)


def rgb2hsv_u8(rgb):
    """cv2.cvtColor(rgb, COLOR_RGB2HSV) for uint8 (hue range 180)."""
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * SDIV[v] + (1 << 11)) >> 12
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + (~vg & (r - g + 4 * diff))))
    h = (h * HDIV180[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], -1).astype(np.uint8)


def cloth_mask_raw(rgb):
    hsv = rgb2hsv_u8(rgb)
    inrange = (hsv <= 100).all(-1)
    return (~inrange).astype(np.uint8)


def largest_component(arr):
    """get_largest_component: the foreground component with the most pixels (ties: the one labelled first, i.e. whose first
    pixel comes first in raster order), 8-connectivity; None when the mask is empty."""
    lab, n = ndimage.label(arr, structure=np.ones((3, 3), int))
    if n == 0:
        return None
    counts = np.bincount(lab.ravel(), minlength=n + 1)[1:]
    best = int(np.argmax(counts))  # first maximum = lowest label
    return (lab == best + 1).astype(np.uint8)


def adaptive_crop(mask):
    """simEnv.py:722-731: crop side (before the `crop < dimx` test) from the mask's bounding box, or None."""
    if mask is None or not mask.any():
        return None
    x, y = np.where(mask)
    dimx, dimy = mask.shape
    cropx = max(dimx - 2 * x.min(), dimx - 2 * (dimx - x.max()))
    cropy = max(dimy - 2 * y.min(), dimy - 2 * (dimy - y.max()))
    return int(max(cropx, cropy) * 1.5)


def get_obs(rgba, depth, render_dim, image_dim):
    """pyflex.render() output (flat uint8 RGBA bottom-up, flat float32 depth) -> (obs float32 [4, S, S], rgb uint8 [S, S, 3],
    depth [S, S], mask of the largest cloth component or None, crop or None)."""
    rgb = np.flip(np.asarray(rgba).reshape(render_dim, render_dim, 4), 0)[:, :, :3].astype(np.uint8)
    d = np.flip(np.asarray(depth, np.float32).reshape(render_dim, render_dim), 0)
    rgb = resize_linear_u8(np.ascontiguousarray(rgb), image_dim)
    d = resize_linear_f32(np.ascontiguousarray(d), image_dim)
    mask = largest_component(cloth_mask_raw(rgb))
    obs = np.concatenate([rgb.astype(np.float32) / np.float32(255), d[:, :, None]], 2).transpose(2, 0, 1)
    return np.ascontiguousarray(obs, np.float32), rgb, d, mask, adaptive_crop(mask)
