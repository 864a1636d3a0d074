"""Seeded synthetic stereo pairs used by the tests and by bench.py (no dataset access on the GPU box).

S1 "noise"      : uniform-noise pair, numpy default_rng(seed) -- BASELINE.json config
                  "1920x1080 D=128 synthetic random pair" (arms ~0, most pixels fail the LR check).
S2 "structured" : textured scene with piecewise-smooth ground-truth disparity, rectangles and sensor
                  noise (SURVEY.md 8d recipe): natural-image-like arm lengths / support sizes /
                  invalid-pixel fraction, so the aggregation and voting kernels do real work.
All return (left, right) uint8 [H][W][3] in BGR order.
"""
import numpy as np


def noise_pair(width=1920, height=1080, seed=12345):
    rng = np.random.default_rng(seed)
    left = rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
    right = rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
    return left, right


def _bilinear_up(grid, out_h, out_w, cell):
    """grid [gh][gw][c] -> [out_h][out_w][c], grid node (i,j) sits at pixel (i*cell, j*cell)."""
    ys = np.arange(out_h, dtype=np.float64) / cell
    xs = np.arange(out_w, dtype=np.float64) / cell
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    g00 = grid[y0][:, x0]
    g01 = grid[y0][:, x0 + 1]
    g10 = grid[y0 + 1][:, x0]
    g11 = grid[y0 + 1][:, x0 + 1]
    return (g00 * (1 - fy) * (1 - fx) + g01 * (1 - fy) * fx + g10 * fy * (1 - fx) + g11 * fy * fx)


def structured_pair(width=1920, height=1080, disp_range=128, seed=777, return_gt=False):
    rng = np.random.default_rng(seed)
    W, H, D = int(width), int(height), int(disp_range)
    TW = W + D
    tex = np.full((H, TW, 3), 128.0)
    for cell, amp in ((64, 70.0), (16, 40.0), (4, 20.0), (1, 6.0)):
        g = rng.uniform(-1.0, 1.0, (H // cell + 3, TW // cell + 3, 3))
        tex += amp * _bilinear_up(g, H, TW, cell)
    nrect = max(1, (W * H) // 20000)
    for _ in range(nrect):
        rw, rh = int(rng.integers(8, 97)), int(rng.integers(8, 97))
        x0, y0 = int(rng.integers(0, max(1, TW - rw))), int(rng.integers(0, max(1, H - rh)))
        tex[y0:y0 + rh, x0:x0 + rw] = 0.25 * tex[y0:y0 + rh, x0:x0 + rw] + rng.uniform(-90.0, 90.0, 3)
    tex = np.clip(tex, 0.0, 255.0)
    # ground-truth left disparity: smooth background + constant-disparity foreground rectangles
    g = rng.uniform(0.0, 1.0, (H // 128 + 3, W // 128 + 3, 1))
    span = max(1.0, 0.6 * (D - 16))
    gt = np.rint(min(8, D // 4) + span * _bilinear_up(g, H, W, 128)[:, :, 0])
    for _ in range(max(4, nrect // 8)):
        rw = int(rng.integers(max(2, W // 16), max(3, W // 4)))
        rh = int(rng.integers(max(2, H // 16), max(3, H // 4)))
        x0, y0 = int(rng.integers(0, max(1, W - rw))), int(rng.integers(0, max(1, H - rh)))
        gt[y0:y0 + rh, x0:x0 + rw] = np.rint(rng.uniform(0.5 * D, max(0.5 * D + 1, D - 8)))
    gt = np.clip(gt, 0, D - 1).astype(np.int64)
    right = tex[:, D:D + W]
    cols = np.arange(W)[None, :] + D - gt          # left(x) = right(x - d)
    left = np.take_along_axis(tex, np.broadcast_to(cols[:, :, None], (H, W, 3)), axis=1)
    left = np.clip(left + rng.normal(0.0, 2.0, left.shape), 0, 255).astype(np.uint8)
    right = np.clip(right + rng.normal(0.0, 2.0, right.shape), 0, 255).astype(np.uint8)
    left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
    if return_gt:
        return left, right, gt.astype(np.float32)
    return left, right


def quantized_noise_pair(width, height, disp_range, seed, levels=64):
    """Small test pattern with long-ish arms: quantised noise, left = right shifted by D/2."""
    rng = np.random.default_rng(seed)
    a = (rng.integers(0, 256, (height, width + disp_range, 3), dtype=np.uint8) // levels * levels).astype(np.uint8)
    s = disp_range // 2
    return np.ascontiguousarray(a[:, s:s + width]), np.ascontiguousarray(a[:, :width])
