// png_kernels.hip -- PNG files decoded on the device (the LiDAR / ground-truth stream of
// dataloaders/frame_utils.py:66-74, `cv2.imread(f, IMREAD_ANYDEPTH)`, never round-trips through a host image library).
//
// A PNG is a chunk list; the IDAT chunks concatenate to one zlib stream (RFC 1950) holding DEFLATE blocks (RFC 1951);
// the inflated bytes are H scanlines of 1 filter byte + W * bpp bytes, filtered with None/Sub/Up/Average/Paeth
// (PNG spec section 9).  Both steps are sequential per file (bit-serial Huffman codes, byte recurrences along x and y),
// so the unit of parallelism is the FILE: one workgroup per file of a batch; lane 0 decodes (tables in LDS), all
// lanes convert the finished rows.  Supported: non-interlaced gray (colour type 0) 8/16 bit -- what KITTI / Middlebury
// disparity PNGs are -- and RGB (type 2) 8 bit through the u8 entry; anything else returns a status code.
// Output of the disparity entry: disp = sample * scale (scale = 1/256 for readDispKITTI :67, 1 for readDispMidd :72),
// float32, and valid = disp > 0.
#include "vppx_internal.h"

struct PngJobs {
    const u8 *blob;        // the files of the batch, back to back
    const long long *offs; // [n + 1] byte offsets into blob (device)
    u8 *raw;               // [n][raw_stride] inflated scanlines (scratch)
    size_t raw_stride;
    float *disp;           // [n][H][W] or null
    u8 *valid;             // [n][H][W] or null
    u8 *out_u8;            // [n][H][W][C] or null (8-bit images)
    int *status;           // [n] 0 = ok
    int H, W, C;           // expected geometry (C = channels of the u8 output; 1 for disparity maps)
    float scale;
};

enum { PNG_OK = 0, PNG_E_SIG = 1, PNG_E_IHDR = 2, PNG_E_UNSUPPORTED = 3, PNG_E_SIZE = 4, PNG_E_STREAM = 5, PNG_E_HUFF = 6,
       PNG_E_FILTER = 7, PNG_E_TRUNC = 8 };

// ---- byte source: walks the IDAT chunks of one file ------------------------------------------------------
struct IdatReader {
    const u8 *f;     // file
    long long n;     // file size
    long long pos;   // next byte of the current IDAT payload
    long long end;   // end of the current IDAT payload
    int err;
    u32 bitbuf;
    int bitcnt;
    u32 cache;                 // the aligned 4-byte word that holds the last byte read (one global load per 4 bytes)
    unsigned long long cache_a;
    __device__ u32 be32(long long p) const { return ((u32)f[p] << 24) | ((u32)f[p + 1] << 16) | ((u32)f[p + 2] << 8) | (u32)f[p + 3]; }
    __device__ bool next_chunk()
    { // position on the next IDAT chunk (pos = end = end of a payload, followed by its 4-byte CRC)
        long long p = end + 4;
        while (p + 12 <= n) {
            const u32 len = be32(p);
            const u32 type = be32(p + 4);
            if (type == 0x49444154u) { // "IDAT"
                pos = p + 8;
                end = pos + len;
                if (end + 4 > n) { err = PNG_E_TRUNC; return false; }
                if (len == 0) { p = end + 4; continue; }
                return true;
            }
            if (type == 0x49454E44u) break; // "IEND"
            p += 12 + (long long)len;
        }
        err = PNG_E_TRUNC;
        return false;
    }
    __device__ u32 byte()
    {
        if (pos >= end && !next_chunk()) return 0;
        const unsigned long long a = (unsigned long long)(f + pos);
        pos++;
        if ((a & ~3ull) != cache_a) { // (the blob is padded: an aligned word never leaves the allocation)
            cache_a = a & ~3ull;
            cache = *(const u32 *)cache_a;
        }
        return (cache >> ((a & 3ull) * 8)) & 255u;
    }
    __device__ u32 bits(int k)
    { // k <= 16, LSB first (RFC 1951 3.1.1)
        while (bitcnt < k) {
            bitbuf |= byte() << bitcnt;
            bitcnt += 8;
        }
        const u32 v = bitbuf & ((1u << k) - 1u);
        bitbuf >>= k;
        bitcnt -= k;
        return v;
    }
};

// canonical Huffman code given by the number of codes of each length and the symbols in code order (RFC 1951 3.2.2)
struct Huff {
    u16 *count; // [16]
    u16 *sym;   // [n symbols]
};

__device__ int huff_build(const Huff &h, const u8 *len, int n)
{
    for (int i = 0; i < 16; i++) h.count[i] = 0;
    for (int i = 0; i < n; i++) h.count[len[i]]++;
    if (h.count[0] == n) return 0; // no codes (allowed for the distance alphabet)
    int left = 1;
    for (int l = 1; l < 16; l++) {
        left <<= 1;
        left -= h.count[l];
        if (left < 0) return -1; // over-subscribed
    }
    u16 offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h.count[l];
    for (int i = 0; i < n; i++)
        if (len[i]) h.sym[offs[len[i]]++] = (u16)i;
    return left; // > 0: incomplete code
}

__device__ int huff_decode(IdatReader &r, const Huff &h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= (int)r.bits(1);
        const int c = h.count[l];
        if (code - c < first) return h.sym[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__constant__ u16 k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ u8 k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ u16 k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ u8 k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ u8 k_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// inflate the zlib stream of the file's IDAT chunks into out[0..cap); returns bytes written or a negative status
__device__ long long inflate_idat(IdatReader &r, u8 *out, long long cap, u16 *lds /* >= 16+288+16+32 u16 + 320 bytes */)
{
    Huff lc, dc;
    lc.count = lds; lc.sym = lds + 16;
    dc.count = lds + 16 + 288; dc.sym = lds + 16 + 288 + 16;
    u8 *lens = (u8 *)(lds + 16 + 288 + 16 + 32);
    const u32 cmf = r.byte(), flg = r.byte(); // RFC 1950: CM = 8, no preset dictionary, header check
    if ((cmf & 15u) != 8u || (flg & 32u) || ((cmf << 8) | flg) % 31u) return -PNG_E_STREAM;
    long long n = 0;
    int last;
    do {
        last = (int)r.bits(1);
        const int type = (int)r.bits(2);
        if (r.err) return -r.err;
        if (type == 0) { // stored
            r.bitbuf = 0; r.bitcnt = 0;
            u32 len = r.byte(); len |= r.byte() << 8;
            u32 nlen = r.byte(); nlen |= r.byte() << 8;
            if ((len ^ 0xFFFFu) != nlen) return -PNG_E_STREAM;
            if (n + len > cap) return -PNG_E_SIZE;
            for (u32 i = 0; i < len; i++) out[n++] = (u8)r.byte();
        } else if (type == 1 || type == 2) {
            if (type == 1) { // fixed codes (RFC 1951 3.2.6)
                for (int i = 0; i < 144; i++) lens[i] = 8;
                for (int i = 144; i < 256; i++) lens[i] = 9;
                for (int i = 256; i < 280; i++) lens[i] = 7;
                for (int i = 280; i < 288; i++) lens[i] = 8;
                huff_build(lc, lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                huff_build(dc, lens, 30);
            } else { // dynamic codes (3.2.7)
                const int nlen = (int)r.bits(5) + 257, ndist = (int)r.bits(5) + 1, ncode = (int)r.bits(4) + 4;
                if (nlen > 286 || ndist > 30) return -PNG_E_HUFF;
                for (int i = 0; i < 19; i++) lens[i] = 0;
                for (int i = 0; i < ncode; i++) lens[k_clen_order[i]] = (u8)r.bits(3);
                if (huff_build(lc, lens, 19) != 0) return -PNG_E_HUFF; // the code-length code must be complete
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = huff_decode(r, lc);
                    if (sym < 0 || r.err) return -PNG_E_HUFF;
                    if (sym < 16) {
                        lens[idx++] = (u8)sym;
                    } else {
                        int prev = 0, rep;
                        if (sym == 16) {
                            if (idx == 0) return -PNG_E_HUFF;
                            prev = lens[idx - 1];
                            rep = 3 + (int)r.bits(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)r.bits(3);
                        } else {
                            rep = 11 + (int)r.bits(7);
                        }
                        if (idx + rep > nlen + ndist) return -PNG_E_HUFF;
                        while (rep--) lens[idx++] = (u8)prev;
                    }
                }
                if (lens[256] == 0) return -PNG_E_HUFF; // no end-of-block code
                // the distance lengths follow the literal/length ones in the same array
                int e = huff_build(dc, lens + nlen, ndist);
                if (e < 0 || (e > 0 && ndist - dc.count[0] != 1)) return -PNG_E_HUFF;
                e = huff_build(lc, lens, nlen);
                if (e < 0 || (e > 0 && nlen - lc.count[0] != 1)) return -PNG_E_HUFF;
            }
            while (true) {
                int sym = huff_decode(r, lc);
                if (sym < 0 || r.err) return -PNG_E_HUFF;
                if (sym < 256) {
                    if (n >= cap) return -PNG_E_SIZE;
                    out[n++] = (u8)sym;
                } else if (sym == 256) {
                    break;
                } else {
                    sym -= 257;
                    if (sym >= 29) return -PNG_E_HUFF;
                    const int len = k_len_base[sym] + (int)r.bits(k_len_extra[sym]);
                    const int ds = huff_decode(r, dc);
                    if (ds < 0 || ds >= 30) return -PNG_E_HUFF;
                    const long long dist = (long long)k_dist_base[ds] + (long long)r.bits(k_dist_extra[ds]);
                    if (dist > n) return -PNG_E_STREAM;
                    if (n + len > cap) return -PNG_E_SIZE;
                    for (int i = 0; i < len; i++, n++) out[n] = out[n - dist]; // may overlap: byte by byte
                }
            }
        } else {
            return -PNG_E_STREAM;
        }
    } while (!last);
    return r.err ? -(long long)r.err : n;
}

__global__ void __launch_bounds__(64) png_decode_kernel(PngJobs j)
{
    __shared__ u16 tables[16 + 288 + 16 + 32 + 160];
    __shared__ int s_status, s_bpp, s_depth;
    const int fidx = blockIdx.x, lane = threadIdx.x;
    const u8 *f = j.blob + j.offs[fidx];
    const long long n = j.offs[fidx + 1] - j.offs[fidx];
    u8 *raw = j.raw + (size_t)fidx * j.raw_stride;
    if (lane == 0) {
        int st = PNG_OK, bpp = 0, depth = 0;
        do {
            if (n < 8 + 25 + 12 || f[0] != 0x89 || f[1] != 'P' || f[2] != 'N' || f[3] != 'G' || f[4] != 13 || f[5] != 10 ||
                f[6] != 26 || f[7] != 10) { st = PNG_E_SIG; break; }
            IdatReader r;
            r.f = f; r.n = n; r.err = 0; r.bitbuf = 0; r.bitcnt = 0; r.cache = 0; r.cache_a = ~0ull;
            if (r.be32(8) != 13 || r.be32(12) != 0x49484452u) { st = PNG_E_IHDR; break; } // "IHDR"
            const u32 w = r.be32(16), h = r.be32(20);
            depth = f[24];
            const int ctype = f[25], comp = f[26], filt = f[27], inter = f[28];
            if (comp != 0 || filt != 0 || inter != 0 || !((ctype == 0 && (depth == 8 || depth == 16)) || (ctype == 2 && depth == 8))) {
                st = PNG_E_UNSUPPORTED; break;
            }
            const int ch = ctype == 2 ? 3 : 1;
            if ((int)w != j.W || (int)h != j.H || ch != j.C) { st = PNG_E_SIZE; break; }
            bpp = ch * depth / 8;
            const long long rowb = 1 + (long long)w * bpp, need = rowb * h;
            if ((size_t)need > j.raw_stride) { st = PNG_E_SIZE; break; }
            r.pos = r.end = 8 + 8 + 13; // "end of a payload": the IHDR data; next_chunk() skips its CRC
            const long long got = inflate_idat(r, raw, need, tables);
            if (got < 0) { st = (int)-got; break; }
            if (got != need) { st = PNG_E_SIZE; break; }
            // reverse the scanline filters in place (PNG 9.2): byte recurrences along x and y
            for (u32 y = 0; y < h && st == PNG_OK; y++) {
                u8 *cur = raw + y * rowb + 1;
                const u8 *up = y ? raw + (y - 1) * rowb + 1 : nullptr;
                const int ft = cur[-1];
                const long long nb = rowb - 1;
                switch (ft) {
                case 0: break;
                case 1:
                    for (long long i = bpp; i < nb; i++) cur[i] = (u8)(cur[i] + cur[i - bpp]);
                    break;
                case 2:
                    if (up) for (long long i = 0; i < nb; i++) cur[i] = (u8)(cur[i] + up[i]);
                    break;
                case 3:
                    for (long long i = 0; i < nb; i++) {
                        const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                        cur[i] = (u8)(cur[i] + ((a + b) >> 1));
                    }
                    break;
                case 4:
                    for (long long i = 0; i < nb; i++) {
                        const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                        cur[i] = (u8)(cur[i] + paeth(a, b, c));
                    }
                    break;
                default: st = PNG_E_FILTER;
                }
            }
        } while (false);
        s_status = st; s_bpp = bpp; s_depth = depth;
        j.status[fidx] = st;
    }
    __syncthreads();
    if (s_status != PNG_OK) return;
    // all lanes: samples -> outputs
    const int W = j.W, H = j.H, bpp = s_bpp;
    const long long rowb = 1 + (long long)W * bpp;
    const size_t npix = (size_t)H * W;
    if (j.out_u8) {
        u8 *o = j.out_u8 + (size_t)fidx * npix * j.C;
        for (size_t i = lane; i < npix * j.C; i += 64) {
            const size_t y = i / ((size_t)W * j.C), k = i % ((size_t)W * j.C);
            o[i] = raw[y * rowb + 1 + k];
        }
    }
    if (j.disp) {
        float *d = j.disp + (size_t)fidx * npix;
        u8 *v = j.valid ? j.valid + (size_t)fidx * npix : nullptr;
        for (size_t i = lane; i < npix; i += 64) {
            const size_t y = i / W, x = i % W;
            const u8 *p = raw + y * rowb + 1 + x * bpp;
            const u32 s = s_depth == 16 ? (((u32)p[0] << 8) | p[1]) : p[0]; // PNG samples are big-endian
            const float val = __fmul_rn((float)s, j.scale); // exact for scale = 2^-8: what `img / 256.0` gives
            d[i] = val;
            if (v) v[i] = val > 0.0f ? 1 : 0;
        }
    }
}

int handoff_png_decode(vppx_ctx *ctx, int n_files, const u8 *blob_dev, const long long *offs_dev, int H, int W, int C, int max_bpp,
                       u8 *raw_scratch, size_t raw_stride, float scale, float *disp, u8 *valid, u8 *out_u8, int *status_dev)
{
    (void)max_bpp;
    PngJobs j;
    j.blob = blob_dev; j.offs = offs_dev; j.raw = raw_scratch; j.raw_stride = raw_stride; j.disp = disp; j.valid = valid;
    j.out_u8 = out_u8; j.status = status_dev; j.H = H; j.W = W; j.C = C; j.scale = scale;
    png_decode_kernel<<<dim3(n_files), 64, 0, ctx->stream>>>(j);
    VPPX_CHECK_LAUNCH();
    return 0;
}
