"""GPU: PNG files decoded on the device (vppx_png_decode_dev: chunk walk + inflate + unfiltering) against an
independent decoder (PIL) and the reference's formulas (frame_utils.readDispKITTI :66-69: img / 256.0, valid = disp > 0;
readDispMidd :71-74).  Files come from PIL's encoder (adaptive filters, dynamic Huffman blocks, several compression
levels incl. stored blocks) and from a hand-rolled writer that forces every filter type, fixed-Huffman blocks, split
IDAT chunks and ancillary chunks."""
import io
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    from vppstereo_amd.engine import Engine
    assert torch.cuda.is_available()
    return Engine()


def _pil_png(arr, **kw):
    from PIL import Image
    buf = io.BytesIO()
    mode = "I;16" if arr.dtype == np.uint16 else ("RGB" if arr.ndim == 3 else "L")
    img = Image.fromarray(arr) if arr.dtype == np.uint16 else Image.fromarray(arr, mode)
    assert img.mode == mode
    img.save(buf, format="PNG", **kw)
    return buf.getvalue()


def _pil_decode(b):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(b)))


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _filter_rows(img_bytes, bpp, types):
    """img_bytes: [H, rowbytes] uint8 -> filtered scanlines (PNG spec 9.2) with the given filter type per row."""
    H, nb = img_bytes.shape
    out = bytearray()
    prev = np.zeros(nb, np.int32)
    for y in range(H):
        cur = img_bytes[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        t = types[y % len(types)]
        if t == 0:
            f = cur
        elif t == 1:
            f = cur - a
        elif t == 2:
            f = cur - prev
        elif t == 3:
            f = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pr = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            f = cur - pr
        out.append(t)
        out += (f & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def _raw_png(arr, types=(0, 1, 2, 3, 4), level=6, strategy=zlib.Z_DEFAULT_STRATEGY, idat_split=(1 << 30,), extra=True):
    """hand-rolled PNG writer: explicit filter types, zlib strategy and IDAT chunking"""
    H, W = arr.shape[:2]
    if arr.dtype == np.uint16:
        depth, ctype, bpp = 16, 0, 2
        rows = arr.astype(">u2").view(np.uint8).reshape(H, W * 2)
    elif arr.ndim == 3:
        depth, ctype, bpp = 8, 2, 3
        rows = arr.reshape(H, W * 3)
    else:
        depth, ctype, bpp = 8, 0, 1
        rows = arr.reshape(H, W)
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = co.compress(_filter_rows(rows, bpp, types)) + co.flush()
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 0))
    if extra:
        out += _chunk(b"tEXt", b"Comment\x00ancillary chunk before the image data") + _chunk(b"pHYs", struct.pack(">IIB", 1, 1, 0))
    pos, k = 0, 0
    while pos < len(z):
        n = idat_split[k % len(idat_split)]
        out += _chunk(b"IDAT", z[pos:pos + n])
        if k == 1:
            out += _chunk(b"IDAT", b"")  # an empty IDAT chunk is legal
        pos += n
        k += 1
    return out + _chunk(b"IEND", b"")


def _kitti_like(rng, H, W, p=0.2):
    d = rng.uniform(1.0, 200.0, (H, W))
    d = (d * 256).astype(np.uint16)
    d[rng.random((H, W)) > p] = 0
    # smooth stripes make the adaptive filters pick Sub/Up/Paeth
    d[: H // 3] = (np.linspace(256, 40000, W).astype(np.uint16))[None, :]
    return d


def test_kitti_disparity_png_pil_encoder(eng):
    rng = np.random.default_rng(1)
    files, want = [], []
    for lvl in (0, 1, 6, 9):
        d = _kitti_like(rng, 375, 1242)
        b = _pil_png(d, compress_level=lvl)
        assert np.array_equal(_pil_decode(b), d)
        files.append(b)
        want.append(d)
    disp, valid = eng.png_decode(files, 375, 1242)
    disp, valid = disp.cpu().numpy(), valid.cpu().numpy()
    for i, d in enumerate(want):
        ref = d / 256.0                                       # frame_utils.py:67
        assert np.array_equal(disp[i], ref.astype(np.float32)) and np.array_equal(disp[i].astype(np.float64), ref)
        assert np.array_equal(valid[i], (ref > 0.0).astype(np.uint8))
    # readDispMidd: same read without the division
    disp1, _ = eng.png_decode(files[:1], 375, 1242, scale=1.0)
    assert np.array_equal(disp1[0].cpu().numpy(), want[0].astype(np.float32))


@pytest.mark.parametrize("types,strategy,split", [((0,), zlib.Z_DEFAULT_STRATEGY, (1 << 30,)), ((1,), zlib.Z_FIXED, (7, 1, 4096)),
                                                  ((2,), zlib.Z_DEFAULT_STRATEGY, (3,)), ((3,), zlib.Z_FIXED, (1000,)),
                                                  ((4,), zlib.Z_DEFAULT_STRATEGY, (65536,)), ((0, 1, 2, 3, 4), zlib.Z_HUFFMAN_ONLY, (999, 5)),
                                                  ((4, 3, 2, 1, 0), zlib.Z_RLE, (1 << 30,))])
def test_every_filter_type_huffman_mode_and_idat_split(eng, types, strategy, split):
    rng = np.random.default_rng(len(types) + split[0])
    d = _kitti_like(rng, 61, 157, p=0.5)
    for lvl in (0, 6):
        b = _raw_png(d, types=types, level=lvl, strategy=strategy, idat_split=split)
        assert np.array_equal(_pil_decode(b), d)              # the hand-rolled writer is a valid PNG
        disp, valid = eng.png_decode([b], 61, 157)
        assert np.array_equal(disp[0].cpu().numpy(), (d / 256.0).astype(np.float32))
        assert np.array_equal(valid[0].cpu().numpy(), (d > 0).astype(np.uint8))


def test_8bit_gray_and_rgb_images(eng):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (48, 100), dtype=np.uint8)
    g[:, 30:60] = 77
    rgb = rng.integers(0, 256, (48, 100, 3), dtype=np.uint8)
    rgb[10:30] = rgb[10]
    out = eng.png_decode([_pil_png(g), _raw_png(g, types=(4, 1))], 48, 100, channels=1, want="u8").cpu().numpy()
    assert np.array_equal(out[0, ..., 0], g) and np.array_equal(out[1, ..., 0], g)
    out = eng.png_decode([_pil_png(rgb), _raw_png(rgb, types=(3, 4, 2))], 48, 100, channels=3, want="u8").cpu().numpy()
    assert np.array_equal(out[0], rgb) and np.array_equal(out[1], rgb)
    d8, v8 = eng.png_decode([_pil_png(g)], 48, 100, scale=1.0)
    assert np.array_equal(d8[0].cpu().numpy(), g.astype(np.float32))


def test_rejected_files(eng):
    rng = np.random.default_rng(4)
    d = _kitti_like(rng, 20, 33)
    good = _raw_png(d)
    inter = bytearray(good); inter[28] = 1                    # IHDR interlace method 1 (Adam7): unsupported
    # (a file cut in the middle of its deflate stream is reported as truncated or as a bad stream, whichever the
    # decoder notices first)
    for bad, code in ((b"not a png at all" * 8, "1"), (bytes(inter), "3"), (good[: len(good) // 2], "[58]"),
                      (_raw_png(_kitti_like(rng, 21, 33)), "4")):
        with pytest.raises(ValueError, match=f"status {code}"):
            eng.png_decode([good, bad], 20, 33)
    corrupt = bytearray(good)
    k = good.index(b"IDAT") + 4 + 20
    corrupt[k] ^= 0xFF                                        # garbage inside the deflate stream
    with pytest.raises(ValueError):
        eng.png_decode([bytes(corrupt)], 20, 33)


def test_adler32_of_the_inflated_stream_is_verified(eng):
    """zlib ends with the Adler-32 of the inflated bytes (RFC 1950); cv2.imread / PIL reject a file whose checksum is wrong,
    and so does the device decoder (status 9).  The checksum is the last four bytes of the last IDAT payload."""
    rng = np.random.default_rng(6)
    d = _kitti_like(rng, 40, 90)
    good = _raw_png(d, extra=False)
    disp, _ = eng.png_decode([good], 40, 90)
    assert np.array_equal(disp[0].cpu().numpy(), (d / 256.0).astype(np.float32))
    bad = bytearray(good)
    k = good.rindex(b"IEND") - 4 - 4 - 1     # last byte of the last IDAT payload (before its CRC and the IEND length field)
    bad[k] ^= 0x01
    with pytest.raises(ValueError, match="status 9"):
        eng.png_decode([bytes(bad)], 40, 90)


def test_batch_of_kitti_sized_files_unaligned_blob(eng):
    """32 files of different sizes back to back: no alignment or padding is required of the blob."""
    rng = np.random.default_rng(7)
    files, want = [], []
    for i in range(32):
        d = _kitti_like(rng, 375, 1242, p=0.05 + 0.01 * i)
        files.append(_pil_png(d, compress_level=6 if i % 2 else 3))
        want.append(d)
    assert len({len(f) % 16 for f in files}) > 1
    disp, valid = eng.png_decode(files, 375, 1242)
    disp = disp.cpu().numpy()
    for i, d in enumerate(want):
        assert np.array_equal(disp[i], (d / 256.0).astype(np.float32)), i
