"""Host helpers of the drop-in API (reference utils.py:24-71).

normalize / denormalize / psnr keep the reference's exact numpy semantics (PSNR is the
parity metric of BASELINE.json).  ssim restates what the reference's call computes (cv2 gray
conversion + scikit-image 0.14 ``compare_ssim`` defaults) with numpy / scipy, because neither
package is installed here: PARITY UNPINNED against the real packages.  Video decoding (cv2) and
the VGG / YOLO metrics are third-party and out of scope of this path:
clips are numpy arrays here.
"""
import math
import os

import numpy as np


def get_videos(configs):
    """Files of the comma-separated directories in configs['data']['videos'] (reference utils.py:24-31)."""
    found = []
    for d in str(configs['data']['videos']).replace(' ', '').split(','):
        if not d:
            continue
        for f in sorted(os.listdir(d)):
            full = os.path.join(d, f)
            if os.path.isfile(full):
                found.append(full)
    return found


def load_clip(path):
    """A clip file is a .npy array [T, H, W, 3], uint8 (0..255) or float in [0,1]."""
    if not path.lower().endswith('.npy'):
        raise ValueError(f"{path}: only .npy clips are supported (video decoding via OpenCV is outside "
                         "the scope of this path; convert frames to a [T,H,W,3] array)")
    clip = np.load(path, allow_pickle=False)
    if clip.ndim != 4 or clip.shape[-1] != 3:
        raise ValueError(f"{path}: expected an array of shape [T,H,W,3], got {clip.shape}")
    return clip


def get_frames(video, configs, rng=np.random):
    """c consecutive HR frames of a clip (reference utils.py:34-54), from a .npy clip instead of
    cv2.VideoCapture: a random start frame with ``train.c_frames`` frames after it.  The reference
    squeezes every frame to high_res x high_res with an arbitrary-ratio cv2 cubic resize; without OpenCV a
    random high_res crop (one position for the whole sequence) is taken instead -- frames must be at least
    that large.  Returns [c, high_res, high_res_w, 3] in the clip's dtype."""
    clip = load_clip(video) if isinstance(video, str) else video
    c = int(configs['train']['c_frames'])
    hh = int(configs['data']['high_res'])
    ww = int(configs['data'].get('high_res_w', hh))
    T, H, W, _ = clip.shape
    if T < c or H < hh or W < ww:
        raise ValueError(f'clip of shape {clip.shape} is too small for {c} frames of {hh}x{ww}')
    t0 = int(rng.randint(0, T - c + 1))
    y0 = int(rng.randint(0, H - hh + 1))
    x0 = int(rng.randint(0, W - ww + 1))
    return clip[t0:t0 + c, y0:y0 + hh, x0:x0 + ww]


def normalize(input_data):
    return input_data.astype(np.float32) / 255.0


def denormalize(input_data):
    input_data = input_data * 255.0
    return input_data.astype(np.uint8)          # truncation, like the reference


def psnr(img1, img2):
    mse = np.mean((img1 - img2) ** 2)
    if mse == 0:
        return 100
    pixel_max = 1.0
    return 20 * math.log10(pixel_max / math.sqrt(mse))


def ssim(img1, img2):
    """Mean structural similarity as reference utils.py:74-77 computes it:
    ``cv2.cvtColor(img.astype('Float32'), COLOR_BGR2GRAY)`` (Y = 0.114 B + 0.587 G + 0.299 R) on both
    images, then ``skimage.measure.compare_ssim(a, b, full=True)[0]`` with its defaults: 7x7 uniform
    window (scipy ``uniform_filter``, reflect borders), sample covariance (N/(N-1)), K1 = 0.01,
    K2 = 0.03, data_range = 2 (skimage takes the dtype range (-1, 1) of a float image when none is
    given), mean over the image cropped by 3 pixels per side."""
    from scipy.ndimage import uniform_filter
    if img1.shape != img2.shape or img1.ndim != 3 or img1.shape[2] != 3:
        raise ValueError('ssim expects two [H,W,3] BGR images of the same shape')
    if min(img1.shape[:2]) < 7:
        raise ValueError('win_size exceeds image extent')       # skimage's own error
    coef = np.array([0.114, 0.587, 0.299], np.float32)
    x = (img1.astype(np.float32) @ coef).astype(np.float64)
    y = (img2.astype(np.float32) @ coef).astype(np.float64)
    win, data_range = 7, 2.0
    cov_norm = win * win / (win * win - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def blur_cubic_downscale(hr):
    """HR -> LR as the reference synthesises its inputs (FRVSR.py:274-277): ``cv2.GaussianBlur(hr,
    (5, 5), 0)`` followed by ``cv2.resize(..., INTER_CUBIC)`` to a quarter of the size, restated with
    numpy because OpenCV is not installed (PARITY UNPINNED against cv2):
    * sigma 0 with a 5-tap kernel selects OpenCV's fixed table [1, 4, 6, 4, 1] / 16, borders
      BORDER_REFLECT_101;
    * an exact x4 reduction samples at source coordinate 4d + 1.5, i.e. the four pixels of each block
      of 4 with the cubic (a = -0.75) weights of phase 0.5: [-3, 19, 19, -3] / 32, no border cases.
    hr: [T, H, W, 3] float32 with H, W multiples of 4."""
    T, H, W, C = hr.shape
    if H % 4 or W % 4:
        raise ValueError('HR height and width must be multiples of 4')
    g = np.array([1, 4, 6, 4, 1], np.float32) / 16
    k = np.array([-3, 19, 19, -3], np.float32) / 32
    x = hr.astype(np.float32)
    for axis, n in ((1, H), (2, W)):
        pad = [(0, 0)] * 4
        pad[axis] = (2, 2)
        xp = np.pad(x, pad, mode='reflect')                     # reflect without repeating the edge = REFLECT_101
        x = sum(g[i] * np.take(xp, np.arange(i, i + n), axis=axis) for i in range(5))
    x = np.tensordot(x.reshape(T, H // 4, 4, W, C), k, axes=([2], [0]))            # rows
    x = np.tensordot(x.reshape(T, H // 4, W // 4, 4, C), k, axes=([3], [0]))       # columns
    return x.astype(np.float32)


def cubic_upscale4(lr):
    """LR -> HR bicubic baseline of the reference's metrics (FRVSR.py:304-305): ``cv2.resize(lr, 4x,
    INTER_CUBIC)`` restated with numpy (OpenCV absent: PARITY UNPINNED).  Destination pixel d samples
    the source at (d + 0.5) / 4 - 0.5 with the 4-tap cubic kernel, a = -0.75; taps outside the image
    are clamped to the border pixel (OpenCV's resize replicates edges).  lr: [T, h, w, 3] float."""
    a = -0.75

    def weights(t):
        return np.array([((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a,
                         ((a + 2) * t - (a + 3)) * t * t + 1,
                         ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1,
                         0.0])

    x = lr.astype(np.float32)
    for axis in (1, 2):
        n = x.shape[axis]
        src = (np.arange(4 * n) + 0.5) / 4 - 0.5
        i0 = np.floor(src).astype(np.int64)
        w = np.stack([weights(t) for t in (src - i0)], 0)
        w[:, 3] = 1.0 - w[:, :3].sum(1)                        # OpenCV computes the last tap as the remainder
        out = 0
        for k in range(4):
            idx = np.clip(i0 - 1 + k, 0, n - 1)
            shape = [1, 1, 1, 1]
            shape[axis] = 4 * n
            out = out + np.take(x, idx, axis=axis) * w[:, k].astype(np.float32).reshape(shape)
        x = out
    return x.astype(np.float32)


def box_downscale(hr, scale=4):
    """HR -> LR stand-in for the reference's GaussianBlur + cubic resize (cv2, FRVSR.py:274-277):
    a plain scale x scale box average.  Used only to synthesise LR inputs from HR clips."""
    T, H, W, C = hr.shape
    return hr.reshape(T, H // scale, scale, W // scale, scale, C).mean(axis=(2, 4)).astype(np.float32)


def split_tiles(frame, rows, cols, th, tw):
    """rows*cols independent tiles, row-major i*cols+j, no halo (reference FRVSR.py:280-289)."""
    return np.stack([frame[i * th:(i + 1) * th, j * tw:(j + 1) * tw] for i in range(rows) for j in range(cols)])


def merge_tiles(tiles, rows, cols):
    """Inverse of split_tiles for the estimated tiles (reference FRVSR.py:299-301)."""
    _, th, tw, c = tiles.shape
    out = np.empty((rows * th, cols * tw, c), tiles.dtype)
    for i in range(rows):
        for j in range(cols):
            out[i * th:(i + 1) * th, j * tw:(j + 1) * tw] = tiles[i * cols + j]
    return out


def shard_range(n_items, world_size, rank):
    """Contiguous block of items for `rank` (clip c -> GPU c // ceil(C/N))."""
    per = -(-n_items // world_size)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
