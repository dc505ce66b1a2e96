"""Post-processing stage of compute_rsgm (rsgm.py:275-292) through vppx_rsgm_post_dev against the oracle's own pieces
(left/right check, cv2.filterSpeckles(0, 200, 10) restated in oracle/rsgm_oracle.c, _interpolate_background) on disparity
maps built to stress the connected-component labelling: components around the 200-pixel limit, components that cross many
64 x 32 labelling tiles, one-pixel components, one giant component."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    from vppstereo_amd.engine import Engine
    assert torch.cuda.is_available()
    return Engine()


def _pad16(n):
    return -(-n // 16) * 16


def _oracle_post(dl_pad, dr_pad, h, w, subpixel):
    Hp, Wp = dl_pad.shape
    pt, pl = (Hp - h) // 2, (Wp - w) // 2
    fd = np.ascontiguousarray(dl_pad[pt:pt + h, pl:pl + w])
    fdr = np.ascontiguousarray(dr_pad[pt:pt + h, pl:pl + w])
    fdc = fd.copy()
    mask = oracle._left_right_check(fd, fdr, 1)
    fd[mask == 128] = 0
    fd8 = np.ascontiguousarray(fd.astype(np.uint8))
    oracle.filterSpeckles(fd8, 0, 200, 10)
    out = fd8.astype(np.float32)
    if subpixel:
        keep = out != 0
        out[keep] = fdc[keep]
    out = np.ascontiguousarray(out)
    oracle._interpolate_background(out)
    return out


def _consistent_right(dl, rng, flip=0.1):
    """A right map that passes the left/right check almost everywhere (dr[x - d] = dl[x]); `flip` of it is noise."""
    h, w = dl.shape
    dr = np.zeros_like(dl)
    ys, xs = np.nonzero(dl > 0)
    xd = xs - np.rint(dl[ys, xs]).astype(np.int64)
    ok = (xd >= 0) & (xd < w)
    dr[ys[ok], xd[ok]] = dl[ys[ok], xs[ok]]
    noise = rng.random(dl.shape) < flip
    dr[noise] = rng.uniform(0, 190, size=int(noise.sum())).astype(np.float32)
    return dr


def _run(eng, maps, h, w, rng, subpixel=True, flip=0.05):
    import torch
    B = len(maps)
    Hp, Wp = _pad16(h), _pad16(w)
    pt, pl = (Hp - h) // 2, (Wp - w) // 2
    dl = np.zeros((B, Hp, Wp), np.float32)
    dr = np.zeros((B, Hp, Wp), np.float32)
    for f, m in enumerate(maps):
        dl[f] = rng.uniform(1, 100, size=(Hp, Wp)).astype(np.float32)   # the border is cropped away: must not matter
        dl[f, pt:pt + h, pl:pl + w] = m
        dr[f, pt:pt + h, pl:pl + w] = _consistent_right(m, rng, flip)
    got = eng.rsgm_post(torch.from_numpy(dl).to(eng.device), torch.from_numpy(dr).to(eng.device), h, w, subpixel=subpixel).cpu().numpy()
    for f in range(B):
        ref = _oracle_post(dl[f], dr[f], h, w, subpixel)
        assert np.array_equal(got[f], ref), (f, h, w, int((got[f] != ref).sum()))


def _blocks(h, w, rng, lo, hi):
    """Rectangles of random size, neighbours more than 10 apart, +-2 of noise inside, some holes."""
    m = np.zeros((h, w), np.float32)
    y = 0
    while y < h:
        bh = int(rng.integers(lo, hi))
        x = 0
        k = int(rng.integers(0, 2))
        while x < w:
            bw = int(rng.integers(lo, hi))
            base = 20.0 + 30.0 * ((k + (y // 7)) % 5) + float(rng.integers(0, 3)) * 0.25
            m[y:y + bh, x:x + bw] = base + rng.uniform(-2, 2, size=m[y:y + bh, x:x + bw].shape)
            k += 1
            x += bw
        y += bh
    m[rng.random((h, w)) < 0.02] = 0
    return m


@pytest.mark.parametrize("h,w", [(70, 130), (97, 203), (33, 65), (64, 64), (5, 5)])
@pytest.mark.parametrize("subpixel", [True, False])
def test_block_maps_with_components_around_the_size_limit(eng, h, w, subpixel):
    rng = np.random.default_rng(h * 1000 + w)
    maps = [_blocks(h, w, rng, 3, 22), _blocks(h, w, rng, 10, 18), _blocks(h, w, rng, 1, 5), _blocks(h, w, rng, 12, 40)]
    _run(eng, maps, h, w, rng, subpixel)


def test_thin_components_across_many_tiles(eng):
    """One-pixel-wide vertical and horizontal lines of 150..260 pixels (around the limit), separated by zeros: every one of
    them crosses several tile borders, none has more than one pixel per tile row / column."""
    rng = np.random.default_rng(5)
    h, w = 300, 420
    v = np.zeros((h, w), np.float32)
    for i, x in enumerate(range(1, w, 2)):
        n = 150 + (i * 7) % 111
        y0 = (i * 13) % (h - n)
        v[y0:y0 + n, x] = 40 + (i % 3)
    hm = np.zeros((h, w), np.float32)
    for i, y in enumerate(range(1, h, 2)):
        n = 150 + (i * 11) % 111
        x0 = (i * 17) % (w - n)
        hm[y, x0:x0 + n] = 60 + (i % 4)
    _run(eng, [v, hm], h, w, rng, flip=0.0)


def _snake(h, w, length, val):
    """A serpentine one-pixel path of exactly `length` pixels, rows 3 apart, covering the frame left-right-left."""
    m = np.zeros((h, w), np.float32)
    y, x, dx, n = 1, 1, 1, 0
    while n < length and y < h - 1:
        m[y, x] = val
        n += 1
        if (dx > 0 and x == w - 2) or (dx < 0 and x == 1):
            for _ in range(3):       # go down three rows, then turn
                if n < length and y + 1 < h - 1:
                    y += 1
                    m[y, x] = val
                    n += 1
            dx = -dx
            x += dx
        else:
            x += dx
    assert n == length, (n, length)
    return m


@pytest.mark.parametrize("length", [199, 200, 201, 1500])
def test_snake_of_exact_length(eng, length):
    rng = np.random.default_rng(length)
    h, w = 100, 150
    _run(eng, [_snake(h, w, length, 77.0)], h, w, rng, flip=0.0)


def test_checkerboard_and_giant_component(eng):
    rng = np.random.default_rng(11)
    h, w = 135, 250
    yy, xx = np.mgrid[0:h, 0:w]
    checker = np.where((yy + xx) % 2 == 0, 30.0, 90.0).astype(np.float32)        # every pixel its own component
    giant = (50 + 5 * np.sin(yy / 9.0) + 4 * np.cos(xx / 7.0)).astype(np.float32)  # one component
    giant[40:48, 60:80] = 120.0                                                     # a 160-pixel island: removed
    giant[90:105, 100:115] = 120.0                                                  # a 225-pixel island: kept
    ramp = (xx * 0.9 + 1).astype(np.float32)                                        # links everywhere horizontally (diff < 10)
    ramp[:, ::37] = 0
    _run(eng, [checker, giant, ramp], h, w, rng, flip=0.0)


def test_headline_size_random_regions(eng):
    rng = np.random.default_rng(77)
    h, w = 540, 960
    _run(eng, [_blocks(h, w, rng, 4, 30), _blocks(h, w, rng, 8, 24)], h, w, rng)


@pytest.mark.parametrize("h,w", [(70, 130), (40, 300)])
def test_background_interpolation_with_empty_rows_and_frames(eng, h, w):
    """_interpolate_background (rsgm.py:185-227): rows without any valid disparity at the top and at the bottom take the
    first / last non-empty row (the column pass works from per-row flags: after the row pass a row is valid everywhere or
    nowhere), empty rows in between stay empty, an entirely empty frame stays empty, one valid pixel fills its row and,
    through the column pass, the whole frame; large regions so that the speckle filter keeps them."""
    rng = np.random.default_rng(h + w)
    maps = []
    m = _blocks(h, w, rng, 20, 40)
    m[:5] = 0
    m[h - 3:] = 0
    m[h // 2: h // 2 + 2] = 0          # empty rows in the middle: unchanged by both passes
    maps.append(m)
    maps.append(np.zeros((h, w), np.float32))
    m = np.zeros((h, w), np.float32)
    m[h // 3: h // 3 + 18, w // 4: w // 4 + 18] = 37.5   # one block of 324 pixels in the middle of nothing
    maps.append(m)
    m = _blocks(h, w, rng, 20, 40)
    m[1:] = np.where(rng.random((h - 1, w)) < 0.5, 0, m[1:])  # noise (mostly removed as speckles) under a full first row
    maps.append(m)
    _run(eng, maps, h, w, rng, True, flip=0.0)
