"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- NumPy restatement of the reference's input pipeline for one image:
  letterbox (tf.image.resize_image_with_pad, NEAREST) + convert_image_dtype + RGB->BGR   /root/reference/dataset/file_util.py:47-59
  label transform                                                                         /root/reference/dataset/file_util.py:48-53
  augmentation menu (noise, then brightness / saturation / contrast in one of 3 orders)   /root/reference/dataset/dataset_util.py:29-104
The TensorFlow ops the reference calls are not in /root/reference (TensorFlow is a dependency, requirements.txt: tensorflow-gpu 1.13-1.15)
and TensorFlow is not installable here, so their published algorithms are restated:
  resize_image_with_pad: ratio = max(w/W, h/H) (float64), resized = floor(dim / ratio), pad = max(0, floor((target - dim/ratio) / 2))
  ResizeNearestNeighbor (align_corners=False, no half-pixel centres): src = min(floorf(dst * float32(in/out)), in - 1) in float32
  convert_image_dtype(uint8 -> float32): x * float32(1/255)
  adjust_brightness: x + delta; adjust_contrast: (x - mean_hw) * f + mean_hw per channel; adjust_saturation: RGB->HSV, s = clip(s*f, 0, 1), HSV->RGB
PARITY UNPINNED against TensorFlow itself (no TF here, and TF's random streams are not reproducible outside TF anyway): the random draws
come from a counter-based Philox4x32-10 generator shared bit for bit with the HIP kernel, so the GPU path is checked against THIS file
on identical draws.  Only tests/ may import this module."""
import numpy as np

PHILOX_M0, PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
SALT_PEPPER_P = np.float32(0.01)     # dataset_util.py:27 (_random_normal doubles as the salt-and-pepper rate, :41)
GAUSS_STD = np.float32(0.01)
BRIGHTNESS = 30.0 / 255.0            # dataset_util.py:22
CONTRAST = (0.9, 1.1)                # :23-24
SATURATION = (0.9, 1.1)              # :25-26


def philox4x32(counter, key, rounds=10):
    """counter: (..., 4) uint32, key: (2,) uint32 -> (..., 4) uint32 (Salmon et al., SC'11)"""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]), int(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0, p1 = PHILOX_M0 * c[0], PHILOX_M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & mask]
        k0, k1 = (k0 + PHILOX_W0) & 0xFFFFFFFF, (k1 + PHILOX_W1) & 0xFFFFFFFF
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def letterbox_geometry(h, w, H, W):
    ratio = max(float(w) / float(W), float(h) / float(H))
    rh_f, rw_f = float(h) / ratio, float(w) / ratio
    nh, nw = int(np.floor(rh_f)), int(np.floor(rw_f))
    top, left = max(0, int(np.floor((H - rh_f) / 2))), max(0, int(np.floor((W - rw_f) / 2)))
    return nh, nw, top, left


def letterbox(img_rgb_u8, image_size):
    """-> uint8 (H, W, 3) RGB, zero bars"""
    H, W = int(image_size[0]), int(image_size[1])
    h, w = img_rgb_u8.shape[:2]
    nh, nw, top, left = letterbox_geometry(h, w, H, W)
    hs, ws = np.float32(h) / np.float32(nh), np.float32(w) / np.float32(nw)
    ys = np.minimum(np.floor(np.arange(nh, dtype=np.float32) * hs).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(nw, dtype=np.float32) * ws).astype(np.int64), w - 1)
    out = np.zeros((H, W, 3), np.uint8)
    out[top:top + nh, left:left + nw] = img_rgb_u8[ys][:, xs]
    return out


def transform_label(label, h, w, image_size):
    """file_util.py:48-53 in float32"""
    lab = np.asarray(label, dtype=np.float32).reshape(-1, 5).copy()
    src = np.asarray([h, w], np.float32) / np.asarray(image_size, np.float32)
    r = src[::-1] / np.max(src)
    lab[:, 0:2] = lab[:, 0:2] * r + (np.float32(1) - r) / np.float32(2.0)
    lab[:, 2:4] = lab[:, 2:4] * r
    return lab


def to_float_bgr(img_rgb_u8):
    return (img_rgb_u8.astype(np.float32) * np.float32(1.0 / 255))[..., ::-1]


def _uniform24(r):
    return (r >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def pixel_noise(image, noise, seed, n):
    """noise 0: salt and pepper (dataset_util.py:37-44), 1: gaussian (:29-34), else none.  One Philox block per pixel:
    counter (pixel index, image index, 0, 0), key = seed."""
    H, W, _ = image.shape
    if noise not in (0, 1):
        return image
    ctr = np.zeros((H * W, 4), np.uint32)
    ctr[:, 0] = np.arange(H * W, dtype=np.uint32)
    ctr[:, 1] = n
    r = philox4x32(ctr, np.asarray(seed, np.uint32)).reshape(H, W, 4)
    if noise == 0:
        sel = (_uniform24(r[..., 0]) < SALT_PEPPER_P).astype(np.float32)[..., None]
        val = (_uniform24(r[..., 1]) < np.float32(0.5)).astype(np.float32)[..., None]
        return image * (np.float32(1) - sel) + val * sel
    u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)       # (0, 1]
    rad0 = np.sqrt(np.float32(-2.0) * np.log(u[..., 0]))
    rad1 = np.sqrt(np.float32(-2.0) * np.log(u[..., 2]))
    two_pi = np.float32(6.283185307179586)
    z = np.stack([rad0 * np.cos(two_pi * u[..., 1]), rad0 * np.sin(two_pi * u[..., 1]), rad1 * np.cos(two_pi * u[..., 3])], axis=-1)
    return image + (z * GAUSS_STD).astype(np.float32)


def adjust_saturation(image, factor):
    """tf.image.adjust_saturation: RGB -> HSV, scale and clip S, HSV -> RGB (channel order is taken as stored)"""
    r, g, b = image[..., 0], image[..., 1], image[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    rng = v - np.minimum(np.minimum(r, g), b)
    with np.errstate(divide='ignore', invalid='ignore'):
        s = np.where(v > 0, rng / v, np.float32(0)).astype(np.float32)
        norm = (np.float32(1.0) / (np.float32(6.0) * rng)).astype(np.float32)
        hh = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + np.float32(2.0 / 6.0), norm * (r - g) + np.float32(4.0 / 6.0)))
    hh = np.where(rng > 0, hh, np.float32(0)).astype(np.float32)
    hh = np.where(hh < 0, hh + np.float32(1), hh).astype(np.float32)
    s = np.clip(s * np.float32(factor), np.float32(0), np.float32(1)).astype(np.float32)
    c = s * v
    m = v - c
    dh = hh * np.float32(6)
    fm = dh - np.float32(2) * np.floor(dh / np.float32(2))
    x = c * (np.float32(1) - np.abs(fm - np.float32(1)))
    cat = dh.astype(np.int32)
    z = np.zeros_like(c)
    cats = [cat == k for k in range(6)]                      # category 6 (h rounded up to exactly 1) yields (0, 0, 0) + m, as the TF kernel
    rr = np.select(cats, [c, x, z, z, x, c], z)
    gg = np.select(cats, [x, c, c, x, z, z], z)
    bb = np.select(cats, [z, z, x, c, c, x], z)
    return np.stack([rr + m, gg + m, bb + m], axis=-1).astype(np.float32)


def augment(image, noise, color_order, brightness_delta, saturation_factor, contrast_factor, seed, n):
    """DatasetUtil._augment (dataset_util.py:81-99) with the scalar draws given: noise in {0,1,2}, color_order in {0,1,2,3}"""
    x = pixel_noise(np.asarray(image, np.float32), noise, seed, n)
    bd, sf, cf = np.float32(brightness_delta), np.float32(saturation_factor), np.float32(contrast_factor)

    def contrast(t):
        mean = t.astype(np.float64).mean(axis=(0, 1)).astype(np.float32)
        return (t - mean) * cf + mean
    if color_order == 0:
        x = contrast(adjust_saturation(x + bd, sf))
    elif color_order == 1:
        x = contrast(adjust_saturation(x, sf) + bd)
    elif color_order == 2:
        x = contrast(adjust_saturation(x, sf)) + bd
    return np.clip(x, np.float32(0), np.float32(1)).astype(np.float32)
