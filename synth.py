"""Seeded synthetic stereo scenes for bench.py, smoke() and the parity tests.

No dataset or network is available, so frames are generated: left = smooth value-noise
texture + pixel noise; GT disparity = slanted background plane in [0.1D,0.5D] plus
fronto-parallel foreground rectangles in [0.5D,0.9D]; right = forward warp of left by the GT
(foreground wins collisions, holes take the left pixel); hints = Bernoulli(p) samples of the
GT (float32, fractional).  numpy only; deterministic for a given (seed, frame index).
"""
import numpy as np


def _value_noise(rng, H, W, cell):
    gh, gw = H // cell + 2, W // cell + 2
    grid = rng.random((gh, gw, 3)).astype(np.float32)
    ys = np.arange(H, dtype=np.float32) / cell
    xs = np.arange(W, dtype=np.float32) / cell
    y0, x0 = ys.astype(np.int32), xs.astype(np.int32)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    a = grid[y0][:, x0]
    b = grid[y0][:, x0 + 1]
    c = grid[y0 + 1][:, x0]
    d = grid[y0 + 1][:, x0 + 1]
    return (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx)


def make_frame(H, W, D, p_hints, seed=1234, frame=0, channels=3):
    """Returns dict(left u8[H,W,C], right u8[H,W,C], gt f32[H,W], hints f32[H,W])."""
    rng = np.random.default_rng([seed, frame])
    tex = (0.55 * _value_noise(rng, H, W, 32) + 0.30 * _value_noise(rng, H, W, 8) + 0.15 * _value_noise(rng, H, W, 3))
    left = np.clip(tex * 255.0 + rng.uniform(-8, 8, (H, W, 3)), 0, 255).astype(np.uint8)
    # background plane + rectangles
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    a, b = rng.uniform(-0.2, 0.2) * D / max(W, 1), rng.uniform(0.0, 0.3) * D / max(H, 1)
    gt = np.clip(0.1 * D + a * (xx - W / 2) + b * yy + 0.1 * D, 0.1 * D, 0.5 * D).astype(np.float32)
    for _ in range(6):
        h, w = int(rng.integers(H // 8 + 1, H // 3 + 2)), int(rng.integers(W // 10 + 1, W // 4 + 2))
        y0, x0 = int(rng.integers(0, max(1, H - h))), int(rng.integers(0, max(1, W - w)))
        gt[y0:y0 + h, x0:x0 + w] = np.float32(rng.uniform(0.5 * D, 0.9 * D))
    gt = np.minimum(gt, np.float32(D - 2))
    # forward warp left -> right, nearer (larger d) wins
    right = left.copy()
    dq = np.rint(gt).astype(np.int64)
    xr = xx.astype(np.int64) - dq
    ok = xr >= 0
    ys_i = yy.astype(np.int64)[ok]
    xs_i = xx.astype(np.int64)[ok]
    tgt = ys_i * W + xr[ok]
    order = np.argsort(dq[ok], kind="stable")          # ascending disparity: nearest written last
    flat = right.reshape(-1, 3)
    src = left.reshape(-1, 3)[(ys_i * W + xs_i)[order]]
    flat[tgt[order]] = src
    right = flat.reshape(H, W, 3)
    mask = rng.random((H, W)) < p_hints
    hints = np.where(mask, gt, 0).astype(np.float32)
    if channels == 1:
        left = left[..., :1].copy()
        right = right[..., :1].copy()
    return dict(left=np.ascontiguousarray(left), right=np.ascontiguousarray(right), gt=gt, hints=hints)


def make_batch(B, H, W, D, p_hints, seed=1234, frame0=0, channels=3):
    fr = [make_frame(H, W, D, p_hints, seed, frame0 + i, channels) for i in range(B)]
    return {k: np.stack([f[k] for f in fr]) for k in fr[0]}


def uniform_random_pair(H, W, D, p_hints, seed=0, channels=3):
    """The uniform-random-u8 variant used for the survey's CPU numbers."""
    rng = np.random.default_rng(seed)
    l = rng.integers(0, 256, (H, W, channels), dtype=np.uint8)
    r = rng.integers(0, 256, (H, W, channels), dtype=np.uint8)
    g = np.zeros((H, W), np.float32)
    m = rng.random((H, W)) < p_hints
    g[m] = rng.uniform(1, D - 1, size=int(m.sum())).astype(np.float32)
    return l, r, g


# ---------------------------------------------------------------------------------------------------------
# splitmix64: the repository's own generator (Steele, Lea, Flood 2014) for fixtures whose inputs must be the
# same on every machine and numpy version (tests/golden/vpp_anchors_splitmix.json)
# ---------------------------------------------------------------------------------------------------------
def splitmix64(seed, n):
    """The first n outputs of splitmix64 seeded with `seed`, as uint64."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _sm_unit(seed, n):
    """n doubles in [0, 1) from the top 53 bits of the splitmix64 outputs."""
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def anchor_inputs_splitmix(H=540, W=960, D=192, p=0.03):
    """Uniform-random pair, p hints uniform in [1, D-1), a 25 % occlusion mask: the SURVEY App. D recipe with
    splitmix64 in place of numpy's Generator.  Returns l, r, g, occ0, occ1."""
    n = H * W
    l = splitmix64(101, (n * 3 + 7) // 8).view(np.uint8)[: n * 3].reshape(H, W, 3).copy()
    r = splitmix64(202, (n * 3 + 7) // 8).view(np.uint8)[: n * 3].reshape(H, W, 3).copy()
    m = _sm_unit(303, n) < p
    vals = (1.0 + _sm_unit(404, n) * (D - 2)).astype(np.float32)
    g = np.where(m, vals, np.float32(0)).astype(np.float32).reshape(H, W)
    occ1 = (_sm_unit(505, n) < 0.25).astype(np.uint8).reshape(H, W)
    return l, r, g, np.zeros((H, W), np.uint8), occ1
