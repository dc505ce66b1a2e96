// rsgm_kernels.hip -- rSGM back-end on gfx950 (CDNA4), hand-written HIP.
//
// Replaces, for the hot path only, the pyrSGM natives called from the reference's
// models/rsgm/rsgm.py (:25,26 census5x5_SSE; :44 costMeasureCensus5x5_xyd_SSE; :61
// aggregate_SSE; :141 matchWTA_SSE; :142 subPixelRefine; :145,173 median3x3_SSE; :170
// matchWTARight_SSE) and the numba/cv2 glue of compute_rsgm (:250-294).  The arithmetic
// spec is DESIGN.md section 4 (= oracle/rsgm_oracle.c, the checker).
//
// Data layout in HBM (per frame, padded to Hp x Wp multiples of 16):
//   gray    u8  [Hp][Wp]          census  u32 [Hp][Wp]  (24 bits used)
//   path_k  u8|u16 [Hp][Wp][D]    one volume per path, D innermost ("xyd", rsgm.py:42)
//   S       u16 [Hp][Wp][D]       disp    f32 [Hp][Wp]
//
// Wave64 mapping of the disparity axis: a 16-lane DPP row owns one pixel, lane l of the
// row owns the DPL = D/16 consecutive disparities d = DPL*l .. DPL*l+DPL-1, packed two
// per VGPR as u16 pairs (v_pk_add_u16 clamp / v_pk_min_u16 / v_pk_sub_u16).  The d+-1
// exchange is one DPP row_shr/row_shl per pixel, min over d is an in-lane packed min tree
// plus four DPP row_ror steps.  One wave therefore advances 4 scan lines at once.
#include "vppx_internal.h"

#include <stdlib.h>

#define INVALID_DISP_COST 16u
#define INVALID_DISP (-10.0f)

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u16 u16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// packed-u16 helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u32 pk_min(u32 a, u32 b)
{
    u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_adds(u32 a, u32 b)
{
    u16x2 r = __builtin_elementwise_add_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b)
{
    u16x2 r = __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_splat(u32 v) { return (v & 0xFFFFu) | (v << 16); }

// DPP controls (gfx9): row_shl:n 0x100+n, row_shr:n 0x110+n, row_ror:n 0x120+n
template <int CTRL>
__device__ __forceinline__ u32 dpp_keep(u32 old, u32 src)
{
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ u32 dpp_ror(u32 src) // row_ror has no invalid lanes: bound_ctrl lets it fold into the VALU op
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ u32 row_min_u32(u32 v)
{
    v = min(v, dpp_ror<0x128>(v)); // row_ror:8
    v = min(v, dpp_ror<0x124>(v)); // row_ror:4
    v = min(v, dpp_ror<0x122>(v)); // row_ror:2
    v = min(v, dpp_ror<0x121>(v)); // row_ror:1
    return v;
}
__device__ __forceinline__ u32 row_or_u32(u32 v)
{
    v |= dpp_keep<0x128>(v, v);
    v |= dpp_keep<0x124>(v, v);
    v |= dpp_keep<0x122>(v, v);
    v |= dpp_keep<0x121>(v, v);
    return v;
}

// unaligned (4-byte aligned) vector loads/stores of NW dwords
template <int NW>
__device__ __forceinline__ void load_words(const u32 *p, u32 (&w)[NW])
{
    struct __attribute__((packed, aligned(4))) V4 { u32x4 v; };
    struct __attribute__((packed, aligned(4))) V2 { u32x2 v; };
    int i = 0;
#pragma unroll
    for (; i + 4 <= NW; i += 4) {
        u32x4 t = ((const V4 *)(p + i))->v;
        w[i] = t.x; w[i + 1] = t.y; w[i + 2] = t.z; w[i + 3] = t.w;
    }
#pragma unroll
    for (; i + 2 <= NW; i += 2) {
        u32x2 t = ((const V2 *)(p + i))->v;
        w[i] = t.x; w[i + 1] = t.y;
    }
#pragma unroll
    for (; i < NW; i++) w[i] = p[i];
}
template <int NW>
__device__ __forceinline__ void store_words(u32 *p, const u32 (&w)[NW])
{
    struct __attribute__((packed, aligned(4))) V4 { u32x4 v; };
    struct __attribute__((packed, aligned(4))) V2 { u32x2 v; };
    int i = 0;
#pragma unroll
    for (; i + 4 <= NW; i += 4) {
        u32x4 t = {w[i], w[i + 1], w[i + 2], w[i + 3]};
        ((V4 *)(p + i))->v = t;
    }
#pragma unroll
    for (; i + 2 <= NW; i += 2) {
        u32x2 t = {w[i], w[i + 1]};
        ((V2 *)(p + i))->v = t;
    }
#pragma unroll
    for (; i < NW; i++) p[i] = w[i];
}

// streaming (read-once) loads
typedef u32 u32x3 __attribute__((ext_vector_type(3)));
template <int NW>
__device__ __forceinline__ void load_words_nt(const u32 *p, u32 (&w)[NW])
{
    int i = 0;
    if constexpr (NW == 3) { // one global_load_dwordx3
        typedef u32 u32x3a __attribute__((ext_vector_type(3), aligned(4)));
        const u32x3a t = __builtin_nontemporal_load((const u32x3a *)p);
        w[0] = t.x; w[1] = t.y; w[2] = t.z;
        return;
    }
    // (4-byte aligned vector types: a lane's chunk starts at a multiple of its own size, e.g. 24 bytes)
    typedef u32 u32x4a __attribute__((ext_vector_type(4), aligned(4)));
    typedef u32 u32x2a __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (; i + 4 <= NW; i += 4) {
        const u32x4a t = __builtin_nontemporal_load((const u32x4a *)(p + i));
        w[i] = t.x; w[i + 1] = t.y; w[i + 2] = t.z; w[i + 3] = t.w;
    }
#pragma unroll
    for (; i + 2 <= NW; i += 2) {
        const u32x2a t = __builtin_nontemporal_load((const u32x2a *)(p + i));
        w[i] = t.x; w[i + 1] = t.y;
    }
#pragma unroll
    for (; i < NW; i++) w[i] = __builtin_nontemporal_load(p + i);
}

// streaming stores for write-once volumes: keep them from evicting the census / gray rows that
// every step re-reads from L2
template <int NW>
__device__ __forceinline__ void store_words_nt(u32 *p, const u32 (&w)[NW])
{
    int i = 0;
    if constexpr (NW == 3) { // one global_store_dwordx3: the lanes' 12-byte pieces tile whole lines (2 + 1 would leave holes)
        typedef u32 u32x3a __attribute__((ext_vector_type(3), aligned(4)));
        const u32x3a t = {w[0], w[1], w[2]};
        __builtin_nontemporal_store(t, (u32x3a *)p);
        return;
    }
#pragma unroll
    for (; i + 4 <= NW; i += 4) {
        u32x4 t = {w[i], w[i + 1], w[i + 2], w[i + 3]};
        __builtin_nontemporal_store(t, (u32x4 *)(p + i));
    }
#pragma unroll
    for (; i + 2 <= NW; i += 2) {
        u32x2 t = {w[i], w[i + 1]};
        __builtin_nontemporal_store(t, (u32x2 *)(p + i));
    }
#pragma unroll
    for (; i < NW; i++) __builtin_nontemporal_store(w[i], p + i);
}

// ---------------------------------------------------------------------------------------
// pad (cv2.copyMakeBorder BORDER_REFLECT, rsgm.py:258-260) fused with RGB2GRAY
// (rsgm.py:11-12): gray(pad(img)) == pad(gray(img)).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? (-i - 1) : (2 * n - i - 1);
    return i;
}

struct ImgSet { // up to three same-shaped inputs / outputs handled by one launch (blockIdx.z = set * B + frame)
    const void *src[3];
    void *dst[3];
};

__global__ void __launch_bounds__(256) pad_gray_kernel(ImgSet io, int B, int H, int W, int C, int Hp, int Wp, int pad_t,
                                                       int pad_l)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int set = blockIdx.z / B, f = blockIdx.z % B;
    if (x >= Wp) return;
    const u8 *img = (const u8 *)(set == 0 ? io.src[0] : (set == 1 ? io.src[1] : io.src[2]));
    u8 *gray = (u8 *)(set == 0 ? io.dst[0] : (set == 1 ? io.dst[1] : io.dst[2]));
    const int sy = reflect_idx(y - pad_t, H), sx = reflect_idx(x - pad_l, W);
    const u8 *s = img + (((size_t)f * H + sy) * W + sx) * C;
    u32 v;
    if (C == 3) {
        v = (s[0] * 9798u + s[1] * 19235u + s[2] * 3735u + 16384u) >> 15;
    } else {
        v = s[0];
    }
    gray[((size_t)f * Hp + y) * Wp + x] = (u8)v;
}

int rsgm_launch_pad_gray_n(vppx_ctx *ctx, const RsgmGeom &g, int n, const u8 *const *img, u8 *const *gray)
{
    ImgSet io;
    for (int i = 0; i < 3; i++) { io.src[i] = i < n ? img[i] : nullptr; io.dst[i] = i < n ? gray[i] : nullptr; }
    dim3 grid((g.Wp + 255) / 256, g.Hp, g.B * n);
    pad_gray_kernel<<<grid, 256, 0, ctx->stream>>>(io, g.B, g.H, g.W, g.C, g.Hp, g.Wp, g.pad_t, g.pad_l);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int rsgm_launch_pad_gray(vppx_ctx *ctx, const RsgmGeom &g, const u8 *img, u8 *gray)
{
    return rsgm_launch_pad_gray_n(ctx, g, 1, &img, &gray);
}

// ---------------------------------------------------------------------------------------
// Hand-off to the deep front-ends (test.py:179-200): uint8 HWC -> float NCHW in [0,1]
// (`img / 255.` in float64, then .float()), replicate-padded to multiples of `mult` with the
// reference's lo = pad//2, hi = pad - pad//2 split.  Output float32 or bfloat16 (RNE).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) to_nchw_kernel(const u8 *__restrict__ src, void *__restrict__ dst, int H, int W, int C,
                                                      int Hq, int Wq, int pad_t, int pad_l, int bf16)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int fc = blockIdx.z; // frame * C + channel
    if (x >= Wq) return;
    const int f = fc / C, c = fc % C;
    const int sy = min(max(y - pad_t, 0), H - 1), sx = min(max(x - pad_l, 0), W - 1); // F.pad(mode='replicate')
    const u8 v = src[(((size_t)f * H + sy) * W + sx) * C + c];
    const float fv = (float)__ddiv_rn((double)v, 255.0);
    const size_t o = ((size_t)fc * Hq + y) * Wq + x;
    if (bf16) {
        const u32 b = __float_as_uint(fv);
        ((u16 *)dst)[o] = (u16)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16); // round to nearest even (no NaN/inf here)
    } else {
        ((float *)dst)[o] = fv;
    }
}

int rsgm_launch_to_nchw(vppx_ctx *ctx, int B, int H, int W, int C, int mult, const u8 *src, void *dst, int bf16)
{
    const int pad_h = (((H / mult) + 1) * mult - H) % mult, pad_w = (((W / mult) + 1) * mult - W) % mult;
    const int Hq = H + pad_h, Wq = W + pad_w;
    dim3 grid((Wq + 255) / 256, Hq, B * C);
    to_nchw_kernel<<<grid, 256, 0, ctx->stream>>>(src, dst, H, W, C, Hq, Wq, pad_h / 2, pad_w / 2, bf16);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// census 5x5 (call site rsgm.py:25-26).  24 bits, row-major, first neighbour = bit 23,
// bit = (neighbour < centre); 2-px border = 0.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) census5x5_kernel(ImgSet io, int B, int Hp, int Wp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int set = blockIdx.z / B, f = blockIdx.z % B;
    if (x >= Wp) return;
    const u8 *g = (const u8 *)(set == 0 ? io.src[0] : io.src[1]) + (size_t)f * Hp * Wp;
    u32 *out = (u32 *)(set == 0 ? io.dst[0] : io.dst[1]);
    u32 v = 0;
    if (y >= 2 && y < Hp - 2 && x >= 2 && x < Wp - 2) {
        const u32 c = g[(size_t)y * Wp + x];
#pragma unroll
        for (int dy = -2; dy <= 2; dy++)
#pragma unroll
            for (int dx = -2; dx <= 2; dx++) {
                if (dy == 0 && dx == 0) continue;
                v = (v << 1) | (u32)(g[(size_t)(y + dy) * Wp + (x + dx)] < c);
            }
    }
    out[((size_t)f * Hp + y) * Wp + x] = v;
}

int rsgm_launch_census_n(vppx_ctx *ctx, int B, int Hp, int Wp, int n, const u8 *const *gray, u32 *const *census)
{
    ImgSet io;
    for (int i = 0; i < 3; i++) { io.src[i] = i < n ? gray[i] : nullptr; io.dst[i] = i < n ? census[i] : nullptr; }
    dim3 grid((Wp + 255) / 256, Hp, B * n);
    census5x5_kernel<<<grid, 256, 0, ctx->stream>>>(io, B, Hp, Wp);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int rsgm_launch_census(vppx_ctx *ctx, int B, int Hp, int Wp, const u8 *gray, u32 *census)
{
    return rsgm_launch_census_n(ctx, B, Hp, Wp, 1, &gray, &census);
}

// pad + gray + census of the fused pipeline in ONE launch (round 3): set 0 = the un-patterned left image, only its gray
// image is needed (the P2 image, rsgm.py:270); sets 1, 2 = the patterned pair, only their census codes are: a block turns a
// 64 x 16 tile (+ 2 pixels of halo) of the RGB image into gray values in LDS and takes the 24 comparisons from there, so the
// patterned pair's gray images never go to HBM (2 x 17 MB per 32 frames written and read 25 times through the caches before).
#define PGC_TX 64
#define PGC_TY 16
struct PgcArgs {
    const u8 *img[3];
    u8 *gray0;
    u32 *census[2];
    int B, H, W, C, Hp, Wp, pad_t, pad_l;
};
// nbr - centre of two bytes picked out of two dwords (SDWA): the sign bit says nbr < centre
template <int BA, int BB>
__device__ __forceinline__ u32 sub_bytes(u32 a, u32 b)
{
    u32 d;
#define VPPX_SUBB(A, B)                                                                                                   \
    if constexpr (BA == A && BB == B)                                                                                     \
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #A " src1_sel:BYTE_" #B      \
            : "=v"(d)                                                                                                     \
            : "v"(a), "v"(b));
    VPPX_SUBB(0, 0) VPPX_SUBB(0, 1) VPPX_SUBB(0, 2) VPPX_SUBB(0, 3) VPPX_SUBB(1, 0) VPPX_SUBB(1, 1) VPPX_SUBB(1, 2) VPPX_SUBB(1, 3)
    VPPX_SUBB(2, 0) VPPX_SUBB(2, 1) VPPX_SUBB(2, 2) VPPX_SUBB(2, 3) VPPX_SUBB(3, 0) VPPX_SUBB(3, 1) VPPX_SUBB(3, 2) VPPX_SUBB(3, 3)
#undef VPPX_SUBB
    return d;
}
// census code of pixel P (0..3) of a thread's group from the five window rows (two dwords each: columns -2 .. +5 of the group)
template <int P>
__device__ __forceinline__ u32 census_of(const u32 (&w)[5][2])
{
    constexpr int CI = P + 2; // the centre's byte among the 8
    u32 v = 0;
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
        // (v << 1) | (nbr < centre), rsgm.py:25-26 -> census5x5_SSE bit order: rows top to bottom, columns left to right
#define VPPX_CBIT(DX)                                                                                                     \
    if (!(dy == 2 && DX == 2)) {                                                                                          \
        constexpr int NI = P + DX;                                                                                        \
        v = __builtin_amdgcn_alignbit(v, sub_bytes<(NI & 3), (CI & 3)>(w[dy][NI >> 2], w[2][CI >> 2]), 31);              \
    }
        VPPX_CBIT(0) VPPX_CBIT(1) VPPX_CBIT(2) VPPX_CBIT(3) VPPX_CBIT(4)
#undef VPPX_CBIT
    }
    return v;
}

__global__ void __launch_bounds__(256) pad_gray_census_kernel(PgcArgs a)
{
    // tile with halo: LDS column lx holds padded column x0 - 2 + lx, row ly padded row y0 - 2 + ly; the pitch is a multiple of
    // 4 so that the 8 window columns of a group of 4 pixels are two aligned dwords
    constexpr int LW = PGC_TX + 4, LH = PGC_TY + 4, LP = LW + 4;
    static_assert(LW % 4 == 0 && LP % 4 == 0, "groups of four columns");
    __shared__ __attribute__((aligned(16))) u8 tile[LH * LP];
    const int set = blockIdx.z / a.B, f = blockIdx.z % a.B;
    const int x0 = blockIdx.x * PGC_TX, y0 = blockIdx.y * PGC_TY;
    const u8 *img = set == 0 ? a.img[0] : (set == 1 ? a.img[1] : a.img[2]);
    // source rectangle of the tile without any reflection or clamping?  (all but the tiles on the frame's border)
    const int sy0 = y0 - 2 - a.pad_t, sx0 = x0 - 2 - a.pad_l;
    const bool interior = sy0 >= 0 && sy0 + LH <= a.H && sx0 >= 0 && sx0 + LW <= a.W && y0 + PGC_TY + 2 <= a.Hp && x0 + PGC_TX + 2 <= a.Wp;
    if (interior && (a.C == 3 || a.C == 1)) {
        // four pixels per thread and step: 12 (or 4) contiguous source bytes, gray by two byte dot products per pixel
        // ((R * 9798 + G * 19235 + B * 3735 + 16384) >> 15 with the weights split into high and low bytes), one LDS dword
        for (int gi = threadIdx.x; gi < (LW / 4) * LH; gi += 256) {
            const int ly = gi / (LW / 4), g4 = gi % (LW / 4);
            const u8 *sp = img + (((size_t)f * a.H + sy0 + ly) * a.W + sx0 + 4 * g4) * a.C;
            u32 out;
            struct __attribute__((packed, aligned(1))) U1 { u32 v; };
            if (a.C == 3) {
                const u32 w0 = ((const U1 *)sp)->v, w1 = ((const U1 *)(sp + 4))->v, w2 = ((const U1 *)(sp + 8))->v;
                const u32 p0 = w0, p1 = __builtin_amdgcn_alignbyte(w1, w0, 3), p2 = __builtin_amdgcn_alignbyte(w2, w1, 2), p3 = w2 >> 8;
                constexpr u32 WL = 0x00972346u, WH = 0x000E4B26u; // low / high bytes of {9798, 19235, 3735}, fourth byte ignored
                u32 gq[4];
                const u32 px[4] = {p0, p1, p2, p3};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    gq[q] = (__builtin_amdgcn_udot4(px[q], WL, 16384u, false) + (__builtin_amdgcn_udot4(px[q], WH, 0u, false) << 8)) >> 15;
                out = gq[0] | (gq[1] << 8) | (gq[2] << 16) | (gq[3] << 24);
            } else {
                out = ((const U1 *)sp)->v;
            }
            *(u32 *)(tile + ly * LP + 4 * g4) = out;
        }
    } else {
        for (int i = threadIdx.x; i < LW * LH; i += 256) {
            const int ly = i / LW, lx = i % LW;
            int y = y0 + ly - 2, x = x0 + lx - 2;
            y = min(max(y, 0), a.Hp - 1); // (outside the padded image only the census border reads, and that is zero anyway)
            x = min(max(x, 0), a.Wp - 1);
            const int sy = reflect_idx(y - a.pad_t, a.H), sx = reflect_idx(x - a.pad_l, a.W);
            const u8 *s = img + (((size_t)f * a.H + sy) * a.W + sx) * a.C;
            u32 v;
            if (a.C == 3) v = (s[0] * 9798u + s[1] * 19235u + s[2] * 3735u + 16384u) >> 15;
            else v = s[0];
            tile[ly * LP + lx] = (u8)v;
        }
    }
    __syncthreads();
    // one group of four pixels of one tile row per thread
    const int ly = threadIdx.x >> 4, g4 = threadIdx.x & 15;
    const int x = x0 + 4 * g4, y = y0 + ly;
    if (x >= a.Wp || y >= a.Hp) return; // (Wp is a multiple of 16: a group is inside or outside as a whole)
    const size_t o = ((size_t)f * a.Hp + y) * a.Wp + x;
    u32 w[5][2];
#pragma unroll
    for (int dy = 0; dy < 5; dy++) {
        const u32 *rp = (const u32 *)(tile + (ly + dy) * LP + 4 * g4);
        w[dy][0] = rp[0];
        w[dy][1] = rp[1];
    }
    if (set == 0) {
        *(u32 *)(a.gray0 + o) = __builtin_amdgcn_alignbyte(w[2][1], w[2][0], 2); // the four centre pixels
        return;
    }
    u32 c[4] = {census_of<0>(w), census_of<1>(w), census_of<2>(w), census_of<3>(w)};
    const bool yin = y >= 2 && y < a.Hp - 2;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (!(yin && x + q >= 2 && x + q < a.Wp - 2)) c[q] = 0;
    *(u32x4 *)((set == 1 ? a.census[0] : a.census[1]) + o) = u32x4{c[0], c[1], c[2], c[3]};
}

int rsgm_launch_pad_gray_census(vppx_ctx *ctx, const RsgmGeom &g, const u8 *left, const u8 *left_vpp, const u8 *right_vpp, u8 *gray_left,
                                u32 *census_l, u32 *census_r)
{
    PgcArgs a;
    a.img[0] = left; a.img[1] = left_vpp; a.img[2] = right_vpp;
    a.gray0 = gray_left; a.census[0] = census_l; a.census[1] = census_r;
    a.B = g.B; a.H = g.H; a.W = g.W; a.C = g.C; a.Hp = g.Hp; a.Wp = g.Wp; a.pad_t = g.pad_t; a.pad_l = g.pad_l;
    dim3 grid((g.Wp + PGC_TX - 1) / PGC_TX, (g.Hp + PGC_TY - 1) / PGC_TY, 3 * g.B);
    pad_gray_census_kernel<<<grid, 256, 0, ctx->stream>>>(a);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// materialised Hamming cost volume (call site rsgm.py:44) -- stage API only; the fused
// path recomputes costs from the census pair inside the aggregation kernel.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cost_kernel(const u32 *__restrict__ cl, const u32 *__restrict__ cr,
                                                   u16 *__restrict__ dsi, int Hp, int Wp, int D)
{
    // one thread per (pixel, d-pair)
    const size_t npairs = (size_t)Hp * Wp * (D / 2);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.z;
    if (i >= npairs) return;
    const int dp = (int)(i % (D / 2));
    const size_t pix = i / (D / 2);
    const int x = (int)(pix % Wp);
    const size_t base = (size_t)f * Hp * Wp;
    const u32 a = cl[base + pix];
    const int d0 = 2 * dp, d1 = d0 + 1;
    u32 c0 = d0 <= x ? __popc(a ^ cr[base + pix - d0]) : INVALID_DISP_COST;
    u32 c1 = d1 <= x ? __popc(a ^ cr[base + pix - d1]) : INVALID_DISP_COST;
    ((u32 *)dsi)[base * (D / 2) + i] = c0 | (c1 << 16);
}

// _guided_dsi (rsgm.py:116-127, numba): at every hint pixel the cost row is multiplied by
// k*(1 - exp(-(hint-d)^2 / (2 c^2))), k = 10, c = 1, in float64 and truncated back to uint16.
// hints / validhints are the un-padded [B,H,W] maps; rsgm.py:266-267 pads them with zeros.
__global__ void __launch_bounds__(256) guided_dsi_kernel(u16 *__restrict__ dsi, const float *__restrict__ hints,
                                                         const float *__restrict__ valid, int H, int W, int Hp, int Wp, int D,
                                                         int pad_t, int pad_l)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const size_t o = ((size_t)f * H + y) * W + x;
    if (!(valid[o] > 0)) return;
    const double hv = (double)hints[o];
    u16 *c = dsi + (((size_t)f * Hp + y + pad_t) * Wp + x + pad_l) * D;
    for (int d = 0; d < D; d++) {
        const double diff = __dsub_rn(hv, (double)d);
        const double t = __dmul_rn(10.0, __dsub_rn(1.0, exp(__ddiv_rn(-__dmul_rn(diff, diff), 2.0))));
        c[d] = (u16)(int)__dmul_rn((double)c[d], t);
    }
}

int rsgm_launch_guided_dsi(vppx_ctx *ctx, const RsgmGeom &g, u16 *dsi, const float *hints, const float *valid)
{
    dim3 grid((g.W + 255) / 256, g.H, g.B);
    guided_dsi_kernel<<<grid, 256, 0, ctx->stream>>>(dsi, hints, valid, g.H, g.W, g.Hp, g.Wp, g.D, g.pad_t, g.pad_l);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int rsgm_launch_cost(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u32 *cl, const u32 *cr, u16 *dsi)
{
    const size_t npairs = (size_t)Hp * Wp * (D / 2);
    dim3 grid((unsigned)((npairs + 255) / 256), 1, B);
    cost_kernel<<<grid, 256, 0, ctx->stream>>>(cl, cr, dsi, Hp, Wp, D);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// 8-path semi-global aggregation (call site rsgm.py:61, spec DESIGN 4.4).
//   L_r(p,d) = C(p,d) + min(L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1, min_k L_r(p-r,k)+P2)
//              - min_k L_r(p-r,k)              (u16 saturating adds; missing predecessor: L=C)
// Line-parallel: every scan line of every direction is an independent chain.  Vertical and
// diagonal lines are "wrapped" (x = (s +- t) mod Wp, chain restarts at the image border) so
// all of them have exactly Hp steps.  grid = (lines/16, 8 directions, B frames).
// ---------------------------------------------------------------------------------------
struct PathArgs {
    const u8 *gray;
    const u32 *cl;
    const u32 *cr;
    const u16 *dsi;
    const u16 *p2lut;
    void *out;
    int Hp, Wp, D, p1;
    int B, nlb;       // frames, line blocks per direction
    int dir_mask;     // bit k set: run direction k (W,NW,N,NE,E,SE,S,SW)
    int ndirs;        // popcount(dir_mask)
    int vol_of_dir[8]; // output volume index of each direction
    size_t vol_elems; // elements per path volume (B*Hp*Wp*D)
};

// min over the GW lanes that own one pixel (GW = 4, 8 or 16); every lane ends up with it
template <int GW>
__device__ __forceinline__ u32 grp_min_u32(u32 v)
{
    v = min(v, dpp_ror<0xB1>(v)); // quad_perm [1,0,3,2]
    v = min(v, dpp_ror<0x4E>(v)); // quad_perm [2,3,0,1]
    if (GW >= 8) v = min(v, dpp_ror<0x141>(v));  // row_half_mirror
    if (GW >= 16) v = min(v, dpp_ror<0x140>(v)); // row_mirror
    return v;
}

// Three-way packed minimum of u16 pairs whose values are all < 0x7C00: non-negative finite f16 bit patterns order
// like the integers they spell (denormals included: the kernels run with f16 denormals on, the HIP default), so
// gfx950's v_pk_minimum3_f16 is a v_pk_min3_u16 for them -- one half-rate VOP3P op instead of two.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_min3_small(u32 a, u32 b, u32 c)
{
    const f16x2 x = __builtin_bit_cast(f16x2, a), y = __builtin_bit_cast(f16x2, b), z = __builtin_bit_cast(f16x2, c);
    return __builtin_bit_cast(u32, __builtin_elementwise_minimum(__builtin_elementwise_minimum(x, y), z));
}

// One min-plus step for NP packed pairs per lane.  `first`/`last` tell whether this lane holds
// the lowest / highest disparities of its pixel (d-1 of the first and d+1 of the last do not
// exist: 0xFFFF sentinel).  The work is written stage by stage across all pairs: gfx950 needs a
// wait state between a packed (VOP3P) op and a consumer of its result, so NP independent chains
// are kept side by side instead of one dependent chain after the other.
// FUSE: the matching cost is added with v_bcnt_u32_b32's free accumulator (plain add); only
// used by the byte-volume variant, where nothing can saturate.  X0/X1 = cl ^ cr words.
// FUSE + PRE: same small-value arithmetic, but the packed costs C are given.
template <int NP, bool EXACT, int GW, bool FUSE, bool PRE = false>
__device__ __forceinline__ void sgm_update(u32 (&L)[NP], const u32 (&C)[NP], const u32 (&X0)[NP], const u32 (&X1)[NP],
                                           u32 P1pk, u32 P2pk, u32 &minpk, const u32 (&inact)[NP], bool first, bool last)
{
    // FUSE: the three-way minimum below reads the words as f16 bit patterns, so "no neighbour" must stay a finite
    // f16 after P1 has been added: 0x3FFF instead of 0xFFFF (every real value of the byte variant is < 0x300)
    constexpr u32 NONE = FUSE ? 0x3FFF3FFFu : 0xFFFFFFFFu;
    u32 left_in = dpp_keep<0x111>(NONE, L[NP - 1]); // row_shr:1 : lane-1's last pair
    u32 right_in = dpp_keep<0x101>(NONE, L[0]);     // row_shl:1 : lane+1's first pair
    if (GW < 16) {
        left_in = first ? NONE : left_in;
        right_in = last ? NONE : right_in;
    }
    // FUSE = byte volumes: every value stays far below 2^16, so the packed adds / subtracts cannot carry or borrow
    // across the halves and plain 32-bit v_add_u32 / v_sub_u32 do the same job at twice the issue rate of the
    // VOP3P forms on gfx950 (tools/valu_bench.hip: 2 vs 4 cycles per wave-instruction)
    const u32 t2 = FUSE ? minpk + P2pk : pk_adds(minpk, P2pk);
    u32 al[NP + 1]; // al[i] = {L[2i-1], L[2i]}
    al[0] = __builtin_amdgcn_alignbit(L[0], left_in, 16);
#pragma unroll
    for (int i = 1; i < NP; i++) al[i] = __builtin_amdgcn_alignbit(L[i], L[i - 1], 16);
    al[NP] = __builtin_amdgcn_alignbit(right_in, L[NP - 1], 16);
    u32 m[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] = pk_min(al[i], al[i + 1]);
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] = FUSE ? m[i] + P1pk : pk_adds(m[i], P1pk);
    if (FUSE) {
#pragma unroll
        for (int i = 0; i < NP; i++) m[i] = pk_min3_small(m[i], L[i], t2);
    } else {
#pragma unroll
        for (int i = 0; i < NP; i++) m[i] = pk_min(m[i], L[i]);
#pragma unroll
        for (int i = 0; i < NP; i++) m[i] = pk_min(m[i], t2);
    }
    if (FUSE && PRE) { // small-value arithmetic, the packed costs are given (shared by several paths of a pixel)
#pragma unroll
        for (int i = 0; i < NP; i++) m[i] += C[i];
    } else if (FUSE) {
        // m += popc(x0) rides on v_bcnt_u32_b32's accumulator operand, the high half on a shift + add (hipcc emits
        // v_lshlrev_b32 + v_add_u32 rather than one v_lshl_add_u32; forcing the latter through inline asm saved 0.4 % at
        // B=32 and cost 13 % at B=1, where the scheduler could no longer interleave the chains: left to the compiler).
        // The empty asm keeps the two adds from being re-associated into (lshl_or + add).
#pragma unroll
        for (int i = 0; i < NP; i++) {
            u32 t = (u32)__popc(X0[i]) + m[i];
            asm volatile("" : "+v"(t));
            m[i] = ((u32)__popc(X1[i]) << 16) + t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NP; i++) m[i] = pk_adds(m[i], C[i]);
    }
#pragma unroll
    for (int i = 0; i < NP; i++) {
        m[i] = FUSE ? m[i] - minpk : pk_sub(m[i], minpk);
        if (!EXACT) m[i] |= inact[i]; // slots d >= D stay at the 0xFFFF sentinel (never the minimum)
        L[i] = m[i];
    }
    // min over this lane's pairs as a balanced tree (three-way where the values allow it), then across the pixel's lanes
    if (FUSE) {
#pragma unroll
        for (int n = NP; n > 1; n = (n + 2) / 3)
#pragma unroll
            for (int i = 0; 3 * i < n; i++) {
                if (3 * i + 2 < n) m[i] = pk_min3_small(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
                else if (3 * i + 1 < n) m[i] = pk_min(m[3 * i], m[3 * i + 1]);
                else m[i] = m[3 * i];
            }
    } else {
#pragma unroll
        for (int w = 1; w < NP; w <<= 1)
#pragma unroll
            for (int i = 0; i + w < NP; i += 2 * w) m[i] = pk_min(m[i], m[i + w]);
    }
    u32 mm = min(m[0] & 0xFFFFu, m[0] >> 16);
    mm = grp_min_u32<GW>(mm);
    minpk = pk_splat(mm);
}

// operands of one step, fetched one step ahead of their use (software prefetch: a scan line is
// a serial chain, so without it every step would expose a full L2/HBM round trip)
template <int DPL, bool FROM_DSI>
struct StepIn {
    u32 w[FROM_DSI ? DPL / 2 : DPL]; // census words of the right image, or packed u16 costs
    u32 clv;
    int I;
};

template <int DPL, bool EXACT, bool FROM_DSI>
__device__ __forceinline__ void load_step(StepIn<DPL, FROM_DSI> &s, const u8 *gray_f, const u32 *cl_f, const u32 *cr_f,
                                          const u16 *dsi_f, int pixl, int D, int dbase, const u32 (&inact)[DPL / 2])
{
    constexpr int NP = DPL / 2;
    if constexpr (FROM_DSI) {
        const u32 *cp = (const u32 *)(dsi_f + (size_t)((u32)pixl * (u32)D + (u32)dbase));
        if (EXACT) {
            load_words<NP>(cp, s.w);
        } else {
#pragma unroll
            for (int i = 0; i < NP; i++) s.w[i] = inact[i] ? 0u : cp[i];
        }
        s.clv = 0;
    } else {
        // uniform base + 32-bit lane offset (scalar-base addressing, no 64-bit lane arithmetic); the base sits
        // 512 words before the frame so that the offset of the first pixels (x - d < 0) stays non-negative
        const u32 pb = (u32)pixl << 2;
        const char *crb = (const char *)(cr_f - 512);
        load_words<DPL>((const u32 *)(crb + (size_t)(pb + (u32)((512 - dbase - (DPL - 1)) * 4))), s.w); // w[j] = cr[x - (dbase + DPL-1-j)]
        s.clv = *(const u32 *)((const char *)cl_f + (size_t)pb);
    }
    s.I = gray_f[(size_t)(u32)pixl];
}

// packed matching costs of one step.  d > x has no right-image pixel: InvalidDispCost.
template <int DPL, bool FROM_DSI>
__device__ __forceinline__ void step_costs(const StepIn<DPL, FROM_DSI> &s, int lim /* k valid iff k <= lim */,
                                           u32 (&C)[DPL / 2])
{
    constexpr int NP = DPL / 2;
    if constexpr (FROM_DSI) {
#pragma unroll
        for (int i = 0; i < NP; i++) C[i] = s.w[i];
    } else {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            u32 c0 = __popc(s.clv ^ s.w[DPL - 1 - 2 * i]);
            u32 c1 = __popc(s.clv ^ s.w[DPL - 2 - 2 * i]);
            c0 = (2 * i <= lim) ? c0 : INVALID_DISP_COST;
            c1 = (2 * i + 1 <= lim) ? c1 : INVALID_DISP_COST;
            C[i] = (c1 << 16) | c0;
        }
    }
}

// LDS words per wave of the store transposition below: 8 pixels x 192 bytes + their 8 pixel indices
#define TR_WORDS (384 + 8)

template <int DPL, bool EXACT, typename OT>
__device__ __forceinline__ void store_step(OT *out_f, int pixl, int D, int dbase, const u32 (&L)[DPL / 2],
                                           const u32 (&inact)[DPL / 2], u32 *tr = nullptr)
{
    constexpr int NP = DPL / 2;
    const u32 off = (u32)pixl * (u32)D + (u32)dbase;
    if constexpr (sizeof(OT) == 2) {
        u32 *op = (u32 *)((u16 *)out_f + off);
        if (EXACT) {
            store_words<NP>(op, L);
        } else {
#pragma unroll
            for (int i = 0; i < NP; i++)
                if (!inact[i]) op[i] = L[i];
        }
    } else {
        // u16 pairs -> bytes (values are < 256 by construction of the u8 variant)
        u32 bw[NP / 2];
#pragma unroll
        for (int i = 0; i + 1 < NP; i += 2) bw[i / 2] = __builtin_amdgcn_perm(L[i + 1], L[i], 0x06040200u);
        if constexpr (DPL == 24) {
            if (tr) {
                // 8 lanes x 24 disparities per pixel: a lane's 24 bytes are not a power of two, so its two stores
                // (16 + 8 bytes at a 24-byte stride) each leave byte holes in every 128-byte line they touch, and the
                // streaming (nt) write path sends such partial lines on as they are (store-only timing of this launch:
                // 10.6 ms with the holes, 4.7 ms without).  Transpose through LDS so that each store instruction
                // covers whole contiguous lines: 64 x 16 bytes = bytes [0, 1024) of the wave's 8 x 192, then
                // 64 x 8 bytes = bytes [1024, 1536).  The pixels of a wave need not be neighbours in memory (rows of a
                // horizontal line block, wrap of a diagonal), so their indices travel through LDS as well.
                // One wave, in-order LDS queue: no barrier.
                const int lane = threadIdx.x & 63;
                u32x2 *wp = (u32x2 *)(tr + lane * 6);
                wp[0] = u32x2{bw[0], bw[1]};
                wp[1] = u32x2{bw[2], bw[3]};
                wp[2] = u32x2{bw[4], bw[5]};
                tr[384 + (lane >> 3)] = (u32)pixl;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const u32x4 a = *(const u32x4 *)(tr + lane * 4);
                const u32x2 b = *(const u32x2 *)(tr + 256 + lane * 2);
                const u32 pa = tr[384 + lane / 12], pb = tr[384 + (128 + lane) / 24];
                __builtin_nontemporal_store(a, (u32x4 *)((u8 *)out_f + (pa * (u32)D + (u32)(lane % 12) * 16u)));
                __builtin_nontemporal_store(b, (u32x2 *)((u8 *)out_f + (pb * (u32)D + (u32)((128 + lane) % 24) * 8u)));
                return;
            }
        }
        store_words_nt<NP / 2>((u32 *)((u8 *)out_f + off), bw);
    }
}

// state of one scan line while it is being walked
template <int DPL>
struct LineState {
    u32 L[DPL / 2];
    u32 minpk;
    int prevI;
};

template <int DPL, bool EXACT, bool FROM_DSI, typename OT, bool DIAG, int GW>
__device__ __forceinline__ void line_step(LineState<DPL> &st, const StepIn<DPL, FROM_DSI> &in, OT *out_f, const u32 *s_lut,
                                          int x, int pixl, int D, int dbase, int wrap_edge, u32 P1pk,
                                          const u32 (&inact)[DPL / 2], bool first, bool last, bool line_active)
{
    constexpr int NP = DPL / 2;
    // |I(p) - I(p-r)| of two bytes in one v_sad_u8; the table holds P2 already splatted into both halves
    const u32 di = __builtin_amdgcn_sad_u8((u32)in.I, (u32)st.prevI, 0u);
    u32 P2pk = s_lut[di];
    if (DIAG) {
        // chain restart at the image border (missing predecessor => L = C)
        const bool restart = (x == wrap_edge);
        if (FROM_DSI) {
            // arbitrary u16 costs: with L = 0 and min = 0 the update yields exactly C
            const u32 keep = restart ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < NP; i++) st.L[i] &= keep;
            st.minpk &= keep;
        } else {
            // census costs (<= 24, no saturation): with P1 = P2 = 0 the minimum term is min_k L_prev(k)
            // itself, so L = C + min - min = C: two selects instead of zeroing the whole state
            P1pk = restart ? 0u : P1pk;
            P2pk = restart ? 0u : P2pk;
        }
    }
    // byte volumes (no saturation possible) computed from the census pair fuse the cost add into
    // the popcounts; the masked form is only needed where d can exceed x (first D-1 columns).
    // Wave-uniform branch: only those waves pay for the selects.
    constexpr bool CAN_FUSE = !FROM_DSI && EXACT && sizeof(OT) == 1;
    const int lim = x - dbase;
    if (CAN_FUSE && __builtin_amdgcn_ballot_w64(lim < DPL - 1) == 0) {
        u32 X0[NP], X1[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            X0[i] = in.clv ^ in.w[(FROM_DSI ? NP : DPL) - 1 - (FROM_DSI ? 0 : 2 * i)];
            X1[i] = in.clv ^ in.w[(FROM_DSI ? NP : DPL) - (FROM_DSI ? 1 : 2) - (FROM_DSI ? 0 : 2 * i)];
        }
        sgm_update<NP, EXACT, GW, true>(st.L, X0, X0, X1, P1pk, P2pk, st.minpk, inact, first, last);
    } else {
        u32 C[NP];
        step_costs<DPL, FROM_DSI>(in, FROM_DSI ? DPL : lim, C);
        sgm_update<NP, EXACT, GW, false>(st.L, C, C, C, P1pk, P2pk, st.minpk, inact, first, last);
    }
    // No branch around the store in the fast variant: a store inside a conditional block makes hipcc's s_waitcnt
    // bookkeeping assume the smaller number of younger operations, so every wait for a prefetched load also drained
    // the stores issued after it (vmcnt(0) on the loop back edge: the wave idled for a store round trip per step).
    // The lanes of a clamped, inactive line hold exactly the values of the last line and may store them again.
    u32 *tr = (CAN_FUSE && DPL == 24 && GW == 8) ? (u32 *)s_lut + 256 + (threadIdx.x >> 6) * TR_WORDS : nullptr;
    if (CAN_FUSE || line_active) store_step<DPL, EXACT, OT>(out_f, pixl, D, dbase, st.L, inact, tr);
    st.prevI = in.I;
}

// one scan line (GW lanes) from start to end; DIAG lines restart where they wrap around the
// image.  The loop is unrolled by two with ping-pong operand buffers so that the loads of
// step t+1 are in flight while step t computes, without register copies.
template <int DPL, bool EXACT, bool FROM_DSI, typename OT, bool DIAG, int GW>
__device__ __forceinline__ void run_line(const u8 *gray_f, const u32 *cl_f, const u32 *cr_f, const u16 *dsi_f, OT *out_f,
                                         const u32 *s_lut, int Wp, int D, int x, int y, int dxs, int dys, int nsteps,
                                         int dbase, u32 P1pk, bool first, bool last, bool line_active)
{
    constexpr int NP = DPL / 2;
    u32 inact[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) inact[i] = (!EXACT && (dbase + 2 * i >= D)) ? 0xFFFFFFFFu : 0u;
    const int wrap_edge = dxs > 0 ? 0 : Wp - 1;
    LineState<DPL> st;
#pragma unroll
    for (int i = 0; i < NP; i++) st.L[i] = 0; // first step: L = 0, min = 0 => L = C
    st.minpk = 0;
    int pix0 = y * Wp + x, x0 = x;
    StepIn<DPL, FROM_DSI> A, Bq;
    load_step<DPL, EXACT, FROM_DSI>(A, gray_f, cl_f, cr_f, dsi_f, pix0, D, dbase, inact);
    st.prevI = A.I;
    const int dpix = dys * Wp + dxs; // pixel index advance per step (before the diagonal wrap)
    (void)y;
    // Enter the loop with the first operands "used" (no load in flight).  hipcc's s_waitcnt pass merges the counter
    // state of the loop entry with that of the back edge; with loads pending on entry, the merged state keeps waits in
    // the loop header that, on the back edge, drain the volume store issued a few instructions earlier: a full store
    // round trip exposed every iteration.  (Worth 0 on the 8-direction launch, whose waves hide it; the W/E launch of
    // the fused layout is a latency chain per line.)
#pragma unroll
    for (int i = 0; i < (FROM_DSI ? DPL / 2 : DPL); i++) asm volatile("" ::"v"(A.w[i]));
    asm volatile("" ::"v"(A.clv), "v"(A.I));
    for (int t = 0; t < nsteps; t += 2) {
        // position of step t+1 (clamped at the end of the line: a harmless reload).  Only diagonal lines can
        // leave the image sideways: they wrap to the other border of the same row.
        int x1 = x0 + dxs, p1 = pix0 + dpix;
        if (DIAG) {
            const int wr = x1 >= Wp ? -Wp : (x1 < 0 ? Wp : 0);
            x1 += wr;
            p1 += wr;
        }
        // the fast variant only sees padded frames (Hp, Wp multiples of 16): an even number of steps, no tail branch
        constexpr bool EVEN = !FROM_DSI && EXACT && sizeof(OT) == 1;
        const bool have1 = EVEN || (t + 1 < nsteps);
        const int pix1 = have1 ? p1 : pix0;
        load_step<DPL, EXACT, FROM_DSI>(Bq, gray_f, cl_f, cr_f, dsi_f, pix1, D, dbase, inact);
        line_step<DPL, EXACT, FROM_DSI, OT, DIAG, GW>(st, A, out_f, s_lut, x0, pix0, D, dbase, wrap_edge, P1pk, inact,
                                                      first, last, line_active);
        // position of step t+2
        int x2 = x1 + dxs, p2 = pix1 + dpix;
        if (DIAG) {
            const int wr = x2 >= Wp ? -Wp : (x2 < 0 ? Wp : 0);
            x2 += wr;
            p2 += wr;
        }
        const int pix2 = (t + 2 < nsteps) ? p2 : pix1;
        load_step<DPL, EXACT, FROM_DSI>(A, gray_f, cl_f, cr_f, dsi_f, pix2, D, dbase, inact);
        if (have1)
            line_step<DPL, EXACT, FROM_DSI, OT, DIAG, GW>(st, Bq, out_f, s_lut, x1, pix1, D, dbase, wrap_edge, P1pk, inact,
                                                          first, last, line_active);
        x0 = x2;
        pix0 = pix2;
    }
}

template <int GW, int DPL, bool EXACT, bool FROM_DSI, typename OT>
__global__ void __launch_bounds__(256) sgm_paths_kernel(PathArgs a)
{
    constexpr int LPB = 256 / GW; // scan lines per block
    // P2 table + (byte variant of the 8 x 24 layout) the waves' store transposition buffers
    constexpr bool TR = !FROM_DSI && EXACT && sizeof(OT) == 1 && DPL == 24 && GW == 8;
    __shared__ __attribute__((aligned(16))) u32 s_lut[256 + (TR ? 4 * TR_WORDS : 0)];
    s_lut[threadIdx.x] = pk_splat(a.p2lut[threadIdx.x]);
    __syncthreads();

    // blockIdx.x enumerates (line block, direction, frame).  When the batch is a multiple of 8
    // the enumeration is XCD-aware: consecutive block ids go round-robin to the 8 XCDs, so frame
    // f is pinned to XCD f % 8 and its census / gray rows stay in one 4 MiB L2.
    const int nlb = a.nlb; // line blocks per direction
    const int per_frame = nlb * a.ndirs;
    int f, within;
    if (a.B % 8 == 0) {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        f = (j / per_frame) * 8 + xcd;
        within = j % per_frame;
    } else {
        f = blockIdx.x / per_frame;
        within = blockIdx.x % per_frame;
    }
    // long (horizontal) lines first so that they do not form the tail of the launch
    const int dslot = within / nlb, lb = within % nlb;
    int dir = 0;
    {
        // slot -> direction, in the order 0,4,1,2,3,5,6,7 restricted to the enabled directions
        const int order[8] = {0, 4, 1, 2, 3, 5, 6, 7};
        int seen = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool on = (a.dir_mask >> order[i]) & 1;
            if (on && seen == dslot) dir = order[i];
            seen += on ? 1 : 0;
        }
    }
    const bool horiz = (dir == 0 || dir == 4);
    const int nlines = horiz ? a.Hp : a.Wp;
    if (lb * LPB >= nlines) return;
    int line = lb * LPB + (threadIdx.x / GW);
    const bool line_active = line < nlines; // partial last block: redo the last line, store nothing
    line = line_active ? line : nlines - 1;
    const int lg = threadIdx.x % GW;
    const int nsteps = horiz ? a.Wp : a.Hp;
    const int Wp = a.Wp, Hp = a.Hp, D = a.D;

    int x, y, dxs, dys;
    if (horiz) {
        y = line; dys = 0;
        x = (dir == 0) ? 0 : Wp - 1;
        dxs = (dir == 0) ? 1 : -1;
    } else {
        x = line;
        const bool down = dir < 4; // 1,2,3 go down; 5,6,7 go up
        y = down ? 0 : Hp - 1;
        dys = down ? 1 : -1;
        dxs = (dir == 2 || dir == 6) ? 0 : ((dir == 1 || dir == 7) ? 1 : -1);
    }
    const bool diag = (dxs != 0) && (dys != 0);
    const int dbase = DPL * lg;
    const u32 P1pk = pk_splat(a.p1 > 65535 ? 65535u : (u32)(a.p1 < 0 ? 0 : a.p1));
    const size_t fpix = (size_t)f * Hp * Wp;
    const u8 *gray_f = a.gray + fpix;
    const u32 *cl_f = FROM_DSI ? nullptr : a.cl + fpix;
    const u32 *cr_f = FROM_DSI ? nullptr : a.cr + fpix;
    const u16 *dsi_f = FROM_DSI ? a.dsi + fpix * D : nullptr;
    OT *out_f = (OT *)a.out + (size_t)a.vol_of_dir[dir] * a.vol_elems + fpix * D;
    const bool first = lg == 0, last = lg == GW - 1;
    if (diag)
        run_line<DPL, EXACT, FROM_DSI, OT, true, GW>(gray_f, cl_f, cr_f, dsi_f, out_f, s_lut, Wp, D, x, y, dxs, dys,
                                                     nsteps, dbase, P1pk, first, last, line_active);
    else
        run_line<DPL, EXACT, FROM_DSI, OT, false, GW>(gray_f, cl_f, cr_f, dsi_f, out_f, s_lut, Wp, D, x, y, dxs, dys,
                                                      nsteps, dbase, P1pk, first, last, line_active);
}

// ---------------------------------------------------------------------------------------
// W and E of the fused layout at D = 192 (round 4): sgm_paths_kernel<16, 12> with the right-image census window kept in
// registers.  A lane's 12 disparities need cr[x - dbase - j], j = 0..11: from one step to the next the window moves by ONE
// word, yet the line kernel reloads all 12 (three unaligned 16-byte loads per lane and step, plus their addressing).  Here
// the loop is unrolled by 12 -- the window's length -- so that "which register holds cr[x - dbase - j]" is a compile-time
// rotation (r[(K -+ j) mod 12] at unrolled position K) and a step brings in one new word: an aligned 16-byte load per lane
// every FOUR steps, eight steps ahead of its use.  The left census word and the gray value of a step come four steps ahead
// through two 4-entry rings, the volume stores use the instruction's immediate offset inside a 12-step group.  Per step:
// 3.25 memory instructions instead of 6, no per-step address arithmetic, the cost of d > x (first 191 columns) selected in
// a wave-uniform branch.  Same arithmetic as sgm_update<.., FUSE, PRE>: bit-identical volumes.
// Four rows per wave (16 lanes x 12 disparities per pixel), four waves per block; both directions in one launch.
// ---------------------------------------------------------------------------------------
struct We12Args {
    const u8 *gray;
    const u32 *cl;
    const u32 *cr;
    const u16 *p2lut;
    u8 *out;          // [2][B][Hp][Wp][D]: W, E
    int Hp, Wp, p1, B;
    size_t vol_elems;
    // The last, part-filled layer of waves (round 6).  A launch is N equal waves of Wp dependent steps each; the chip takes them
    // in layers of one wave per SIMD, and the N mod 1024 waves of the last layer occupy a fraction of the SIMDs for a whole
    // line's time (16 frames of 540 x 960: 4.25 layers cost 5).  The lines of that last layer are therefore cut into `pieces`
    // consecutive pieces of `piece_groups` groups of DPL steps, one wave per piece (blocks n_whole, n_whole + 1, ...: the last
    // to be dispatched, one per SIMD): piece j waits -- asleep, issuing nothing -- until piece j - 1 of its line has left its
    // path state (the lane's packed L values; everything else is reloaded) in `hand` and raised the line's flag, continues the
    // line from there and passes it on.  A piece whose predecessor does not show up within `timeout_ticks` computes the line
    // from its start itself: same values, no dependency, never a hang.
    u32 *hand;        // [tail line][piece boundary][64 lanes][DPL / 2] dwords, then a flag per tail line: serial << 4 | pieces done
    int n_whole;      // lines run whole: blocks [0, n_whole)
    int lead;         // blocks in front of the grid that do nothing (a multiple of 8)
    int ntail, tail_base, tail_pitch; // tail lines; first block of their pieces (a multiple of 8); blocks per piece index (a multiple of 8)
    int pieces;       // pieces per tail line (1: nothing is cut)
    int piece_groups; // groups per piece (the last piece takes what is left)
    u32 serial;       // launch serial in the flags (a flag of another launch never matches)
    long long timeout_ticks;
#ifdef VPPX_EXPERIMENT
    unsigned long long *trace; // experiment builds: per block (start, end) of the 100 MHz wall clock and HW_ID / XCC_ID (tools/we_trace.py)
#endif
};

// DPL = disparities per lane = length of the window = steps per unrolled group (8, 12, 16: D = 128, 192, 256)
template <int DPL>
struct We12State {
    static constexpr int NQ = DPL / 4; // quads of steps per group = sets of prefetched words
    u32 L[DPL / 2];
    u32 minpk;
    u32 prevI;
    // sets of per-quad operands of the left view, indexed by the quad's position in the unrolled group (compile time): two when a
    // group holds an even number of quads, else one per quad
    static constexpr int NS = (NQ % 2 == 0) ? 2 : NQ;
    u32 r[DPL];   // the window, rotating
    u32 T[NQ][4]; // new right-census words of this quad of steps and of the quads ahead
    u32 CLq[NS][4]; // LQ >= 1: left census words of this quad of steps and of the next one (one aligned 16-byte load per quad)
    u32 Gq[NS];     // LQ >= 1 (LQ = 2: only these): their gray values, four to a dword
    u32 CL[4];      // LQ = 0 / 2: left census word of this step and of the three ahead (a load per step)
    u32 GI[4];      // LQ = 0: gray value likewise
};

// Addresses are "buffer resource of the frame + scalar position along the line + 32-bit lane offset" (what differs between
// lanes is the row and the disparity chunk; the position is wave-uniform), so a step spends no vector arithmetic on them.
struct We12Lane {
    __amdgpu_buffer_rsrc_t cr, cl, gray, out; // the frame's right census (starting 512 words early), left census, gray image, volume
    int cr_off;          // lane offsets in bytes: (row * Wp - dbase + 512) * 4, row * Wp * 4, row * Wp, row * Wp * D + dbase
    int cl_off, gray_off, out_off;
    int dbase;
};

template <bool EAST, int DPL>
__device__ __forceinline__ void we12_load_quad(u32 (&T)[4], const We12Lane &ln, int Q, int Wp)
{
    if (!EAST) { // steps x = 4Q .. 4Q+3 bring in cr[x - dbase]
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ln.cr, ln.cr_off, 16 * Q, 0);
        T[0] = v.x; T[1] = v.y; T[2] = v.z; T[3] = v.w;
    } else {     // steps x = Wp-1-4Q-i bring in cr[x - dbase - (DPL-1)]: descending, one word off a 16-byte boundary
        // (the constant part sits in the signed lane offset -- it stays positive: 512 guard words -- so that the scalar offset,
        // which the hardware adds as an UNSIGNED 32-bit value, is never negative: 4Q <= Wp - 4)
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ln.cr, ln.cr_off - 4 * (DPL + 3), 4 * (Wp - 4 * Q), 0);
        T[0] = v.w; T[1] = v.z; T[2] = v.y; T[3] = v.x;
    }
}

// The left census words and gray bytes of quad Q of a line's steps (W: pixels 4Q .. 4Q+3, E: Wp-4-4Q .. Wp-1-4Q, an aligned
// quad either way): two loads per FOUR steps, a whole quad ahead of their first use.  (Until round 6 a word and a byte were
// loaded per step; the byte's zero-extension sat right behind its load, so every step waited -- vmcnt(0): on gfx9 stores count
// too -- for its own loads and for the previous step's volume store.)
template <bool EAST, int DPL>
__device__ __forceinline__ void we12_load_left_quad(u32 (&CLq)[4], u32 &Gq, const We12Lane &ln, int Q, int Wp)
{
    const int Qx = EAST ? Wp / 4 - 1 - Q : Q;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ln.cl, ln.cl_off, 16 * Qx, 0);
    CLq[0] = v.x; CLq[1] = v.y; CLq[2] = v.z; CLq[3] = v.w;
    Gq = __builtin_amdgcn_raw_buffer_load_b32(ln.gray, ln.gray_off, 4 * Qx, 0);
}

// LQ: how the left view's operands come in.  0: a census word and a gray byte per step, four steps ahead (rounds 4-5); 1: both per
// quad of steps; 2: the gray bytes per quad, the census words per step.
template <bool EAST, int DPL, int K, int LQ>
__device__ __forceinline__ void we12_step(We12State<DPL> &st, const u32 *s_lut, const We12Lane &ln, int t0, int Wp, u32 P1pk, bool masked,
                                          int nquads)
{
    constexpr int NP = DPL / 2, NQ = DPL / 4, D = 16 * DPL;
    constexpr int AHEAD = NQ >= 3 ? 2 : 1; // quads between the load of a word and its step
    const int t = t0 + K;
    if (t >= Wp) return; // (wave-uniform: the last group of a line whose length is no multiple of DPL)
    const int x = EAST ? Wp - 1 - t : t;
    // ---- the step's new right-census word enters the window
    st.r[EAST ? (K + DPL - 1) % DPL : K] = st.T[K / 4][K % 4];
    constexpr int NS = We12State<DPL>::NS, SET = (K / 4) % NS, WI = EAST ? 3 - K % 4 : K % 4;
    const u32 clv = LQ == 1 ? st.CLq[SET][WI] : st.CL[K % 4];
    const u32 I = LQ == 0 ? st.GI[K % 4] : __builtin_amdgcn_ubfe(st.Gq[SET], 8 * WI, 8);
    if (LQ != 1) { // ---- operands of the steps ahead (scalar, clamped positions: t is wave-uniform)
        int ta = t + 4;
        ta = ta < Wp ? ta : Wp - 1;
        const int xa = EAST ? Wp - 1 - ta : ta;
        st.CL[K % 4] = __builtin_amdgcn_raw_buffer_load_b32(ln.cl, ln.cl_off, 4 * xa, 0);
        if (LQ == 0) st.GI[K % 4] = (u32)__builtin_amdgcn_raw_buffer_load_b8(ln.gray, ln.gray_off, xa, 0);
    }
    if (K % 4 == 0) {
        // ---- operands of the quads ahead; the sets refilled here held the words of the quad before this one
        if (LQ != 0) {
            int Qn = t / 4 + 1;
            Qn = Qn < nquads ? Qn : nquads - 1;
            if (LQ == 1) we12_load_left_quad<EAST, DPL>(st.CLq[(K / 4 + 1) % NS], st.Gq[(K / 4 + 1) % NS], ln, Qn, Wp);
            else st.Gq[(K / 4 + 1) % NS] = __builtin_amdgcn_raw_buffer_load_b32(ln.gray, ln.gray_off, 4 * (EAST ? Wp / 4 - 1 - Qn : Qn), 0);
        }
        int Q = t / 4 + AHEAD;
        Q = Q < nquads ? Q : nquads - 1;
        we12_load_quad<EAST, DPL>(st.T[(K / 4 + AHEAD) % NQ], ln, Q, Wp);
    }
    // ---- matching costs of the lane's DPL disparities: pair i = (dbase + 2i, dbase + 2i + 1)
    u32 C[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const u32 w0 = st.r[EAST ? (K + 2 * i) % DPL : (K - 2 * i + 2 * DPL) % DPL];
        const u32 w1 = st.r[EAST ? (K + 2 * i + 1) % DPL : (K - 2 * i - 1 + 2 * DPL) % DPL];
        const u32 c0 = __popc(clv ^ w0), c1 = __popc(clv ^ w1);
        C[i] = (c1 << 16) | c0;
    }
    if (masked) { // d > x has no right-image pixel: InvalidDispCost.  (A real, wave-uniform branch: the empty asm keeps the
                  // compiler from turning it into a dozen selects that every step would execute.)
        const int lim = x - ln.dbase;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const u32 keep = (2 * i <= lim ? 0xFFFFu : 0u) | (2 * i + 1 <= lim ? 0xFFFF0000u : 0u);
            u32 c = (C[i] & keep) | (~keep & (INVALID_DISP_COST | (INVALID_DISP_COST << 16)));
            asm volatile("" : "+v"(c));
            C[i] = c;
        }
    }
    const u32 di = __builtin_amdgcn_sad_u8(I, st.prevI, 0u);
    const u32 P2pk = s_lut[di];
    u32 inact[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) inact[i] = 0;
    sgm_update<NP, true, 16, true, true>(st.L, C, C, C, P1pk, P2pk, st.minpk, inact, false, false);
    st.prevI = I;
    // ---- the pixel's DPL bytes of this lane (values < 256 by construction of the byte variant), streaming store
    u32 bw[NP / 2];
#pragma unroll
    for (int i = 0; i < NP / 2; i++) bw[i] = __builtin_amdgcn_perm(st.L[2 * i + 1], st.L[2 * i], 0x06040200u);
    if constexpr (DPL == 8) __builtin_amdgcn_raw_buffer_store_b64(u32x2{bw[0], bw[1]}, ln.out, ln.out_off, D * x, 2 /* nt */);
    else if constexpr (DPL == 12) __builtin_amdgcn_raw_buffer_store_b96(u32x3{bw[0], bw[1], bw[2]}, ln.out, ln.out_off, D * x, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(u32x4{bw[0], bw[1], bw[2], bw[3]}, ln.out, ln.out_off, D * x, 2);
}

template <bool EAST, int DPL, int K0, int KN, int LQ>
__device__ __forceinline__ void we12_steps(We12State<DPL> &st, const u32 *s_lut, const We12Lane &ln, int t0, int Wp, u32 P1pk, bool masked,
                                           int nquads)
{
    if constexpr (K0 < KN) {
        we12_step<EAST, DPL, K0, LQ>(st, s_lut, ln, t0, Wp, P1pk, masked, nquads);
        we12_steps<EAST, DPL, K0 + 1, KN, LQ>(st, s_lut, ln, t0, Wp, P1pk, masked, nquads);
    }
}

// Steps [g0 * DPL, g1 * DPL) of a line.  g0 > 0: the line's state after step g0 * DPL - 1 comes from `hand_in` once `flag` says
// so; g1 * DPL < Wp: the state goes to `hand_out` and the flag is raised to `piece + 1`.
template <bool EAST, int DPL, int LQ>
__device__ __forceinline__ void we12_line(const We12Args &a, const u32 *s_lut, const We12Lane &ln, u32 P1pk, int g0, int g1, int piece,
                                          const u32 *hand_in, u32 *hand_out, u32 *flag)
{
    constexpr int NP = DPL / 2, NQ = DPL / 4, D = 16 * DPL;
    constexpr int AHEAD = NQ >= 3 ? 2 : 1;
    const int Wp = a.Wp, nquads = Wp / 4;
    We12State<DPL> st;
    if (g0 > 0) {
        // wait for the piece before this one (wave-uniform; s_sleep keeps the wave off the issue ports its SIMD's other waves need)
        const long long t_start = (long long)__builtin_amdgcn_s_memrealtime();
        bool there = false;
        for (;;) {
            // (readfirstlane: what the flag says is the same for the whole wave, and the compiler must know it -- every position
            // of the line below is a scalar; without it each load and store was wrapped in a waterfall loop: W/E + 10 %)
            const u32 v = (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if ((v >> 4) == (a.serial & 0x0FFFFFFFu) && (int)(v & 15u) >= piece) { there = true; break; }
            if ((long long)__builtin_amdgcn_s_memrealtime() - t_start > a.timeout_ticks) break;
            __builtin_amdgcn_s_sleep(32);
        }
        if (!there) g0 = 0; // nobody came: the whole line up to g1, alone (same values)
        g0 = __builtin_amdgcn_readfirstlane(g0);
    }
    if (g0 > 0) {
        // The record was written on another CU of the SAME XCD (the pieces of a line are blocks 8 apart, and consecutive block ids
        // go round-robin over the XCDs: rsgm_vert3_probe has checked that): the hand-off goes through that XCD's L2 like the
        // lock-step kernel's edge records -- sc1 loads (the CU's L1 bypassed) of what sc1 stores put there, no cache-wide
        // invalidate or write-back.  (Across XCDs the same loads may meet a stale line of the reader's own L2: one wrong frame in
        // a full-size test; an acquire / release fence pair per piece is right but costs every wave of the XCD its cached lines.)
        asm volatile("" ::: "memory");
        const u32 *hp = hand_in + (threadIdx.x & 63) * NP;
#pragma unroll
        for (int i = 0; i < NP; i++) st.L[i] = __hip_atomic_load(hp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // min over the pixel's disparities, as the step before would have left it
        u32 m = st.L[0];
#pragma unroll
        for (int i = 1; i < NP; i++) m = pk_min(m, st.L[i]);
        u32 mm = min(m & 0xFFFFu, m >> 16);
        mm = grp_min_u32<16>(mm);
        st.minpk = pk_splat(mm);
    } else {
#pragma unroll
        for (int i = 0; i < NP; i++) st.L[i] = 0; // first step: L = 0, min = 0 => L = C
        st.minpk = 0;
    }
    const int tb = g0 * DPL; // first step of this piece
    {
        // the window before step tb: W needs cr[x-dbase-(DPL-1) .. x-dbase-1] (x = tb) in slots 1..DPL-1, E needs cr[x-dbase-j] (x = Wp-1-tb) in slot j
        const int so = 4 * (EAST ? Wp - DPL - tb : tb - DPL);
        u32 blk[DPL];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ln.cr, ln.cr_off + so + 16 * q, 0, 0);
            blk[4 * q] = v.x; blk[4 * q + 1] = v.y; blk[4 * q + 2] = v.z; blk[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int m = 0; m < DPL; m++) st.r[m] = EAST ? blk[DPL - 1 - m] : blk[m];
    }
#pragma unroll
    for (int q = 0; q < AHEAD; q++) we12_load_quad<EAST, DPL>(st.T[q], ln, tb / 4 + q < nquads ? tb / 4 + q : nquads - 1, Wp);
    if (LQ == 1) we12_load_left_quad<EAST, DPL>(st.CLq[0], st.Gq[0], ln, tb / 4, Wp);
    if (LQ != 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ta = tb + i < Wp ? tb + i : Wp - 1;
            const int xa = EAST ? Wp - 1 - ta : ta;
            st.CL[i] = __builtin_amdgcn_raw_buffer_load_b32(ln.cl, ln.cl_off, 4 * xa, 0);
            if (LQ == 0) st.GI[i] = (u32)__builtin_amdgcn_raw_buffer_load_b8(ln.gray, ln.gray_off, xa, 0);
        }
    }
    // the gray value of the step before: the line's first pixel has none, |dI| = 0 there (and L = C whatever P2)
    if (tb == 0) st.prevI = LQ == 0 ? st.GI[0] : __builtin_amdgcn_ubfe(st.Gq[0], EAST ? 24 : 0, 8);
    else st.prevI = (u32)__builtin_amdgcn_raw_buffer_load_b8(ln.gray, ln.gray_off, EAST ? Wp - tb : tb - 1, 0);
    const int te = g1 * DPL < Wp ? g1 * DPL : Wp;
    for (int t0 = tb; t0 < te; t0 += DPL) {
        // any step of this group in the first D-1 columns?  (W: at the start of the line, E: at its end)
        const bool masked = EAST ? (Wp - 1 - (t0 + DPL - 1) < D - 1) : (t0 < D - 1);
        we12_steps<EAST, DPL, 0, DPL, LQ>(st, s_lut, ln, t0, Wp, P1pk, masked, nquads);
    }
    if (te < Wp && !(a.serial & 0x10000000u)) { // the next piece of this line continues from here (tests mute this: We12Args)
        u32 *hp = hand_out + (threadIdx.x & 63) * NP;
#pragma unroll
        for (int i = 0; i < NP; i++) __hip_atomic_store(hp + i, st.L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the record is in the XCD's L2 before the flag goes out
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, (a.serial << 4) | (u32)(piece + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifndef WE12_WPB
#define WE12_WPB 1 // waves per block (1: 0.875 ms per 16 frames, 4: 0.90 -- single-wave blocks spread more evenly over the SIMDs)
#endif
template <int DPL, int LQ>
__global__ void __launch_bounds__(64 * WE12_WPB) sgm_we12_kernel(We12Args a)
{
    constexpr int D = 16 * DPL;
    __shared__ u32 s_lut[256];
#ifdef VPPX_EXPERIMENT
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = threadIdx.x; i < 256; i += 64 * WE12_WPB) s_lut[i] = pk_splat(a.p2lut[i]);
    __syncthreads();
    // blocks [0, n_whole) run a line quad each; the blocks behind them a piece of one of the remaining ("tail") line quads
    // `lead` blocks come first and leave at once: one per CU.  A CU hands its waves to its SIMDs in a fixed cycle (0, 2, 1, 3), but
    // the FIRST wave it receives after another kernel lands off that cycle (tools/we_trace.py --order: one SIMD of a fifth of the
    // CUs ends up with 5 whole lines and another with 3), which only matters once the launch is cut to fill every SIMD alike.
    int unit = (int)blockIdx.x - a.lead, piece = 0;
    if (unit < 0) return;
    const int ngroups = (a.Wp + DPL - 1) / DPL; // (the last group of a line may be a part of one)
    int g0 = 0, g1 = ngroups;
    const u32 *hand_in = nullptr;
    u32 *hand_out = nullptr, *flag = nullptr;
    if (a.pieces > 1 && unit >= a.n_whole) {
        // Behind the whole lines (and a few blocks that only bring the count to a multiple of 8): piece j of tail line i is block
        // tail_base + j * tail_pitch + i, tail_pitch a multiple of 8 -- all pieces of a line on one XCD, piece j dispatched before
        // piece j + 1.
        if (unit < a.tail_base) return;
        const int k = unit - a.tail_base, tail = k % a.tail_pitch;
        piece = k / a.tail_pitch;
        if (tail >= a.ntail) return;
        // The pieces are a chain: each must finish before the next can start, so they go first on their SIMDs.  (The issue
        // arbiter takes the oldest wave among equals and these, the last blocks of the grid, are the youngest: without the
        // priority the first piece of a 16-frame launch finished after 615 us of 875, tools/we_trace.py.)
        __builtin_amdgcn_s_setprio(3);
        unit = a.n_whole + tail;
        g0 = piece * a.piece_groups;
        g1 = piece == a.pieces - 1 ? ngroups : g0 + a.piece_groups;
        // records and flags per line quad, i.e. per wave of the block: [tail block][wave][piece boundary][64 lanes][DPL / 2]
        const int wv = WE12_WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), tl = tail * WE12_WPB + wv;
        u32 *rec = a.hand + ((size_t)tl * (a.pieces - 1)) * (64 * (DPL / 2));
        hand_in = piece > 0 ? rec + (size_t)(piece - 1) * (64 * (DPL / 2)) : nullptr;
        hand_out = rec + (size_t)piece * (64 * (DPL / 2)); // (the last piece never writes)
        flag = a.hand + (size_t)a.ntail * WE12_WPB * (a.pieces - 1) * (64 * (DPL / 2)) + tl;
    }
    // `unit` enumerates (row block, direction, frame); XCD-aware like sgm_paths_kernel: frame f -> XCD f % 8
    const int nrb = a.Hp / (4 * WE12_WPB), per_frame = nrb * 2;
    int f, within;
    if (a.B % 8 == 0) {
        const int id = unit, xcd = id & 7, j = id >> 3;
        f = (j / per_frame) * 8 + xcd;
        within = j % per_frame;
    } else {
        f = unit / per_frame;
        within = unit % per_frame;
    }
    const int east = within / nrb, rb = within % nrb;
    const int y = rb * (4 * WE12_WPB) + (int)(threadIdx.x >> 4);
    const int lg = threadIdx.x & 15;
    const int Wp = a.Wp;
    const size_t fpix = (size_t)f * a.Hp * Wp;
    const int npf = a.Hp * Wp; // pixels of a frame (npf * D < 2^32: checked by the launcher)
    We12Lane ln;
    ln.dbase = DPL * lg;
    // (the census buffer has a 512-word guard in front: x - d < 0 reads; the resources end with the frame)
    ln.cr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.cr + fpix - 512), 0, (npf + 512) * 4, 0x00020000);
    ln.cl = __builtin_amdgcn_make_buffer_rsrc((void *)(a.cl + fpix), 0, npf * 4, 0x00020000);
    ln.gray = __builtin_amdgcn_make_buffer_rsrc((void *)(a.gray + fpix), 0, npf, 0x00020000);
    ln.out = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)east * a.vol_elems + fpix * D), 0, (int)((u32)npf * (u32)D), 0x00020000);
    const int rowoff = y * Wp;
    ln.cr_off = (rowoff + 512 - ln.dbase) * 4;
    ln.cl_off = rowoff * 4;
    ln.gray_off = rowoff;
    ln.out_off = rowoff * D + ln.dbase;
    const u32 P1pk = pk_splat((u32)a.p1);
    if (east) we12_line<true, DPL, LQ>(a, s_lut, ln, P1pk, g0, g1, piece, hand_in, hand_out, flag);
    else we12_line<false, DPL, LQ>(a, s_lut, ln, P1pk, g0, g1, piece, hand_in, hand_out, flag);
#ifdef VPPX_EXPERIMENT
    if (a.trace && threadIdx.x == 0) {
        u32 hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[3 * (size_t)blockIdx.x] = trace_t0;
        a.trace[3 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        a.trace[3 * (size_t)blockIdx.x + 2] = ((unsigned long long)(xcc & 0xFu) << 32) | hw;
    }
#endif
}

static int launch_we12(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u8 *gray, const u32 *cl, const u32 *cr, const u16 *p2lut, int p1,
                       void *paths)
{
    We12Args a;
    a.gray = gray; a.cl = cl; a.cr = cr; a.p2lut = p2lut; a.out = (u8 *)paths;
    a.Hp = Hp; a.Wp = Wp; a.B = B;
    a.p1 = p1 < 0 ? 0 : (p1 > 231 ? 231 : p1); // exact for P1 >= P2max (see rsgm_launch_paths)
    a.vol_elems = (size_t)B * Hp * Wp * D;
    // ---- the last layer of waves, cut into pieces (We12Args; VPPX_VARIANT=we_whole keeps every line whole)
    const int nunits = B * 2 * (Hp / (4 * WE12_WPB)), dpl = D / 16, ngroups = (Wp + dpl - 1) / dpl;
    a.hand = nullptr; a.n_whole = nunits; a.pieces = 1; a.piece_groups = ngroups; a.serial = 0; a.timeout_ticks = 0;
    a.ntail = 0; a.tail_base = nunits; a.tail_pitch = 8; a.lead = 0;
    {
        static int nsimd_of[VPPX_MAX_DEVICES] = {};
        const int dv = ctx->device & (VPPX_MAX_DEVICES - 1);
        if (!nsimd_of[dv]) {
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || ncu <= 0) ncu = 256;
            nsimd_of[dv] = 4 * ncu;
        }
        // (in blocks of WE12_WPB waves: a layer is one wave per SIMD; "we_layer=N": tests cut small launches)
        const int nsimd = ctx->knobs.we_layer > 0 ? ctx->knobs.we_layer : nsimd_of[dv] / WE12_WPB, tail = nunits % nsimd;
        // With c pieces per tail line the busiest SIMD carries ceil(tail * c / nsimd) pieces of 1 / c line each, where an uncut tail
        // costs a whole line: the c in 2..8 (4 bits of the flag count the pieces) with the smallest such load, if that saves at least
        // a tenth of a line.  Only behind a whole layer: the pieces' chain needs the other waves' work to hide in.
        // (the pieces of a line hand over through ONE XCD's L2: only where block ids were seen to go round-robin over 8 XCDs)
        // and only where the launch has the chip to itself: next to a lock-step launch the pieces' priority costs that kernel more than
        // the cut saves (8 frames of 540x960x192: W/E 1.19 -> 0.90 ms, the lock-step launch 1.21 -> 1.32, the call 2.15 -> 2.24)
        if (ctx->knobs.we_split && !ctx->capturing && !ctx->we_beside_vert && ctx->vert3_probed && ctx->v3.ok && nunits > nsimd && tail > 0) {
            int pieces = 1;
            double best = 0.9;
            for (int c = 2; c <= 8 && c <= ngroups; c++) {
                const double load = (double)(((long long)tail * c + nsimd - 1) / nsimd) / c;
                if (load < best - 1e-9) { best = load; pieces = c; }
            }
            if (pieces > 1) {
                const size_t rec_dwords = ((size_t)tail * (pieces - 1) * (64 * (dpl / 2)) + tail) * WE12_WPB;
                u32 *hand;
                int rc = ws_get(ctx, WS_WE_HAND, rec_dwords, &hand);
                if (rc) return rc;
                if (ctx->we_hand_seen != (void *)hand || ctx->we_hand_cap != ctx->ws[WS_WE_HAND].cap) {
                    // fresh memory may hold anything -- for instance the flags of another context's launch with the same serial
                    // (a wrong frame in a full-size test, once in a few runs): cleared once, then only this context's serials live here
                    VPPX_HIP(hipMemsetAsync(hand, 0, ctx->ws[WS_WE_HAND].cap, ctx->stream));
                    ctx->we_hand_seen = (void *)hand;
                    ctx->we_hand_cap = ctx->ws[WS_WE_HAND].cap;
                }
                a.hand = hand; a.n_whole = nunits - tail; a.pieces = pieces; a.piece_groups = ngroups / pieces;
                a.ntail = tail; a.tail_base = (a.n_whole + 7) & ~7; a.tail_pitch = (tail + 7) & ~7;
                a.lead = ctx->knobs.we_layer > 0 ? 8 : nsimd_of[dv] / 4; // one per CU
                a.serial = (++ctx->we_serial) & 0x0FFFFFFFu;
                if (a.serial == 0) a.serial = (++ctx->we_serial) & 0x0FFFFFFFu;
                a.timeout_ticks = 20LL * 100000LL; // 20 ms of the 100 MHz wall clock: a whole W/E launch takes 1-3
                if (ctx->knobs.we_mute) { // tests: no piece passes its state on, every later piece gives up after 0.2 ms and computes its line from the start
                    a.timeout_ticks = 20000LL;
                    a.serial |= 0x10000000u; // (readers look for the serial without this bit)
                }
            }
        }
    }
    const dim3 grid((unsigned)(a.pieces > 1 ? a.lead + a.tail_base + a.pieces * a.tail_pitch : nunits));
#ifdef VPPX_EXPERIMENT
    a.trace = nullptr;
    const char *trace_path = ctx->exp_we_trace.empty() ? nullptr : ctx->exp_we_trace.c_str(); // VPPX_EXP_WE_TRACE (vppx_create)
    if (trace_path && !ctx->capturing) {
        u64 *tr;
        int rc = ws_get(ctx, WS_WE_TRACE, (size_t)3 * grid.x, &tr);
        if (rc) return rc;
        a.trace = (unsigned long long *)tr;
        VPPX_HIP(hipMemsetAsync(tr, 0, (size_t)3 * grid.x * 8, ctx->stream)); // (blocks that only pad the grid leave zeros)
    }
#endif
    // how the left view's operands are loaded (bit-identical): next to a lock-step launch fewer memory instructions count;
    // alone on the chip the per-step loads are what was measured faster (NOTEBOOK, round 6).  VPPX_VARIANT=we_lq0 / we_lq1 force one.
    const int lq = ctx->knobs.we_lq >= 0 ? ctx->knobs.we_lq : (ctx->we_beside_vert ? 1 : 0);
#define WE12_LAUNCH(DPL)                                                                                      \
    do {                                                                                                      \
        if (lq == 1) sgm_we12_kernel<DPL, 1><<<grid, 64 * WE12_WPB, 0, ctx->stream>>>(a);                     \
        else sgm_we12_kernel<DPL, 0><<<grid, 64 * WE12_WPB, 0, ctx->stream>>>(a);                             \
    } while (0)
    if (D == 128) WE12_LAUNCH(8);
    else if (D == 192) WE12_LAUNCH(12);
    else WE12_LAUNCH(16);
#undef WE12_LAUNCH
    VPPX_CHECK_LAUNCH();
#ifdef VPPX_EXPERIMENT
    if (a.trace) { // the last launch's trace, every time: (grid, n_whole, pieces, tail_base, tail_pitch, ntail), then 3 words per block
        std::vector<unsigned long long> h((size_t)3 * grid.x + 6);
        VPPX_HIP(hipMemcpyAsync(h.data() + 6, a.trace, (size_t)3 * grid.x * 8, hipMemcpyDeviceToHost, ctx->stream));
        VPPX_HIP(hipStreamSynchronize(ctx->stream));
        h[0] = grid.x; h[1] = (unsigned long long)a.n_whole; h[2] = (unsigned long long)a.pieces;
        h[3] = (unsigned long long)(a.lead + a.tail_base); h[4] = (unsigned long long)a.tail_pitch; h[5] = (unsigned long long)a.ntail;
        if (FILE *f = fopen(trace_path, "wb")) {
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
#endif
    return 0;
}

template <int GW, int DPL>
static int launch_paths_t(vppx_ctx *ctx, PathArgs a, int B, bool from_dsi, int elem_bytes)
{
    constexpr int LPB = 256 / GW;
    a.nlb = ((a.Hp > a.Wp ? a.Hp : a.Wp) + LPB - 1) / LPB;
    dim3 grid((unsigned)(a.nlb * a.ndirs * B), 1, 1);
    const bool exact = (a.D == GW * DPL);
    if (from_dsi) {
        if (exact) sgm_paths_kernel<GW, DPL, true, true, u16><<<grid, 256, 0, ctx->stream>>>(a);
        else sgm_paths_kernel<GW, DPL, false, true, u16><<<grid, 256, 0, ctx->stream>>>(a);
    } else if (!exact) {
        sgm_paths_kernel<GW, DPL, false, false, u16><<<grid, 256, 0, ctx->stream>>>(a);
    } else if (elem_bytes == 2) {
        sgm_paths_kernel<GW, DPL, true, false, u16><<<grid, 256, 0, ctx->stream>>>(a);
    } else {
        sgm_paths_kernel<GW, DPL, true, false, u8><<<grid, 256, 0, ctx->stream>>>(a);
    }
    VPPX_CHECK_LAUNCH();
    return 0;
}

// disparities per lane in the 16-lane-per-pixel kernels (sum / WTA)
static inline int dpl_for(int D) { return D <= 64 ? 4 : (D <= 128 ? 8 : (D <= 192 ? 12 : 16)); }

int rsgm_paths_elem_bytes(int D, int maxp2)
{
    // per-path values are bounded by Cmax + P2max (L_r - min L_r <= P2): bytes suffice when that
    // is < 256 and D fills the lane layout exactly
    return (24 + maxp2 <= 255 && D % 64 == 0) ? 1 : 2;
}

int rsgm_launch_paths(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u8 *gray, const u32 *cl, const u32 *cr,
                      const u16 *dsi, const u16 *p2lut, int p1, void *paths, int elem_bytes, int dir_mask)
{
    PathArgs a;
    a.dir_mask = dir_mask & 0xFF;
    a.ndirs = __builtin_popcount(a.dir_mask);
    if (a.ndirs == 0) return 0;
    {
        // all 8 directions: volume k = direction k (oracle path order); a subset is packed densely
        int nv = 0;
        for (int k = 0; k < 8; k++) a.vol_of_dir[k] = (a.dir_mask == 0xFF) ? k : (((a.dir_mask >> k) & 1) ? nv++ : 0);
    }
    a.gray = gray; a.cl = cl; a.cr = cr; a.dsi = dsi; a.p2lut = p2lut; a.out = paths;
    a.Hp = Hp; a.Wp = Wp; a.D = D; a.p1 = p1;
    a.B = B;
    a.nlb = 0;
    a.vol_elems = (size_t)B * Hp * Wp * D;
    if ((size_t)Hp * Wp * D >= ((size_t)1 << 32)) { vppx_set_error("frame volume too large"); return VPPX_E_UNSUPPORTED; }
    const bool from_dsi = dsi != nullptr;
    if (from_dsi || D % 64 != 0) elem_bytes = 2; // those variants always write u16 volumes
    // Byte volumes run the small-value update (plain 32-bit adds on packed pairs, f16-pattern three-way minimum): its
    // operands must stay far below 0x7C00, which an unclamped P1 would break (0x3FFF + P1).  Clamping P1 to any value
    // >= P2max is exact: L(d+-1) + P1 >= min_k L(k) + P1 >= min_k L(k) + P2, so the P1 terms can never win the minimum
    // once P1 >= P2, and byte volumes imply P2max <= 231.
    if (elem_bytes == 1 && p1 > 231) a.p1 = 231;
    // lanes per pixel x disparities per lane: few lanes per pixel amortise the per-step overhead
    // (min reduction, P2 lookup, addressing) over more disparities
    const int gw_override = ctx->knobs.gw; // VPPX_VARIANT gw4 / gw8 / gw16: lanes per pixel (other layouts per D range)
    {
        // W + E of the fused layout at D = 128 / 192 / 256: the register-window kernel (VPPX_VARIANT we_line: the line kernel)
        const int we12 = !ctx->knobs.we_line;
        if (we12 && (D == 128 || D == 192 || D == 256) && a.dir_mask == 0x11 && !from_dsi && elem_bytes == 1 && Hp % 16 == 0 && Wp % 16 == 0 &&
            Wp >= 16 && gw_override <= 0)
            return launch_we12(ctx, B, Hp, Wp, D, gray, cl, cr, p2lut, p1, paths);
    }
    if (D <= 64) {
        if (gw_override == 8) return launch_paths_t<8, 8>(ctx, a, B, from_dsi, elem_bytes);
        return launch_paths_t<4, 16>(ctx, a, B, from_dsi, elem_bytes);
    }
    // 32 disparities per lane need ~150 VGPRs (3 waves/SIMD) and run 1.7-1.8x slower than 16 per lane
    if (D <= 128) {
        if (gw_override == 4) return launch_paths_t<4, 32>(ctx, a, B, from_dsi, elem_bytes);
        return launch_paths_t<8, 16>(ctx, a, B, from_dsi, elem_bytes);
    }
    if (D <= 192) {
        if (gw_override == 4) return launch_paths_t<4, 48>(ctx, a, B, from_dsi, elem_bytes);
        // fewer than 8 frames do not fill the chip with 8 lanes per pixel (1920 waves per frame): twice the
        // waves at 7 waves/SIMD win below that (B=1: 0.45 vs 0.60 ms), 8 x 24 wins from B=16 on (-5 %)
        // the W/E-only launch of the fused layout has 4352 8-lane waves for 4096 slots: its second round is a tail as long
        // as the first; twice the waves at 7 per SIMD pack better (B=32: 2.34 -> 2.06 ms)
        if (gw_override == 16 || (gw_override != 8 && (B < 8 || a.dir_mask == 0x11)))
            return launch_paths_t<16, 12>(ctx, a, B, from_dsi, elem_bytes);
        return launch_paths_t<8, 24>(ctx, a, B, from_dsi, elem_bytes);
    }
    if (gw_override == 8) return launch_paths_t<8, 32>(ctx, a, B, from_dsi, elem_bytes);
    return launch_paths_t<16, 16>(ctx, a, B, from_dsi, elem_bytes);
}

// ---------------------------------------------------------------------------------------
// Vertical / diagonal paths by band marching (fast path of the fused pipeline).
// Pass 0 walks the image top-down and carries N, NW, NE; pass 1 walks bottom-up and carries
// S, SW, SE.  A workgroup owns VT = 64 columns and advances VR = 16 rows per launch; the three
// paths' state of the previous row lives in LDS as bytes (double buffered), so the three chains
// that meet in a pixel are summed in registers and ONE byte per cell is stored per pass instead
// of three path volumes.  Diagonal chains cross strip borders, so a workgroup also recomputes a
// halo that shrinks by one column per row (columns [x0-15+j, x0+64+15-j) at band row j): no
// inter-workgroup communication inside a launch; the state of the last band row is handed to
// the next launch through a small global buffer (double buffered by band parity).
// Requires byte-sized values: 3*(24 + P2max) <= 255, and D = 16*DPL.
// ---------------------------------------------------------------------------------------
#define VT 64
#define VR 16
#define VC (VT + 2 * VR) // LDS column slots
struct VertArgs {
    const u8 *gray;
    const u32 *cl;
    const u32 *cr;
    const u16 *p2lut;
    u8 *sv;      // [2 passes][B][Hp][Wp][D] summed paths of a pass
    u8 *gst;     // [2 parity][2 pass][B][3][Wp][D] state hand-off
    u16 *gmin;   // [2 parity][2 pass][B][3][Wp]
    int B, Hp, Wp, D, p1, band;
    size_t vol_elems; // B*Hp*Wp*D
};

template <int DPL>
__global__ void __launch_bounds__(768) sgm_vert_kernel(VertArgs a)
{
    constexpr int NP = DPL / 2;
    constexpr int D = 16 * DPL;
    constexpr int NW3 = DPL / 4; // dwords of bytes per lane
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    // layout: state[2][3][VC][D] bytes | mins[2][3][VC] u16 | lut[256] u16
    u8 *st = lds;
    u16 *mn = (u16 *)(lds + (size_t)2 * 3 * VC * D);
    u16 *s_lut = mn + 2 * 3 * VC;
    const int strip = blockIdx.x, pass = blockIdx.y, f = blockIdx.z;
    const int Wp = a.Wp, Hp = a.Hp;
    const int x0 = strip * VT;
    const int g = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const int dbase = DPL * l16;
    for (int i = threadIdx.x; i < 256; i += 768) s_lut[i] = a.p2lut[i];

    const size_t fpix = (size_t)f * Hp * Wp;
    const u8 *gray_f = a.gray + fpix;
    const u32 *cl_f = a.cl + fpix;
    const u32 *cr_f = a.cr + fpix;
    u8 *sv_f = a.sv + (size_t)pass * a.vol_elems + fpix * D;
    const size_t gst_stride = (size_t)2 * a.B * 3 * Wp; // columns per parity
    const int par_in = (a.band + 1) & 1, par_out = a.band & 1;
    const size_t gbase_in = ((size_t)par_in * 2 * a.B + (size_t)pass * a.B + f) * 3 * Wp;
    const size_t gbase_out = ((size_t)par_out * 2 * a.B + (size_t)pass * a.B + f) * 3 * Wp;
    (void)gst_stride;

    // ---- load the previous band's last row (columns x0-VR .. x0+VT+VR) into buffer 0 -------------
    if (a.band > 0) {
        const int ndw = D / 4;
        for (int i = threadIdx.x; i < 3 * VC * ndw; i += 768) {
            const int q = i / (VC * ndw), rem = i % (VC * ndw);
            const int c = rem / ndw, wdx = rem % ndw;
            const int x = x0 - VR + c;
            if (x >= 0 && x < Wp)
                ((u32 *)(st + ((size_t)(0 * 3 + q) * VC + c) * D))[wdx] = ((const u32 *)(a.gst + (gbase_in + (size_t)q * Wp + x) * D))[wdx];
        }
        for (int i = threadIdx.x; i < 3 * VC; i += 768) {
            const int q = i / VC, c = i % VC;
            const int x = x0 - VR + c;
            if (x >= 0 && x < Wp) mn[(0 * 3 + q) * VC + c] = a.gmin[gbase_in + (size_t)q * Wp + x];
        }
    }
    __syncthreads();

    const u32 P1pk = pk_splat(a.p1 > 65535 ? 65535u : (u32)(a.p1 < 0 ? 0 : a.p1));
    const int dy = pass == 0 ? 1 : -1;
    u32 inact[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) inact[i] = 0;

    // operands of a (row, column) step, fetched one step ahead (global loads are L2 hits but would
    // otherwise be fully exposed: only 3 waves per SIMD share a CU with this LDS footprint)
    struct VIn {
        u32 w[DPL];
        u32 clv;
        int I, Ip[3];
        bool valid;
    };
    const int rows_here = min(VR, Hp - a.band * VR);
    auto vfetch = [&](int sidx, VIn &o) {
        const int j = sidx >> 1, it = sidx & 1;
        const int c = j + 1 + g + 48 * it;
        const int x = x0 - VR + c;
        const int row = a.band * VR + j;
        o.valid = (j < rows_here) && !(c > VC - 2 - j || x < 0 || x >= Wp);
        if (o.valid) {
            const int y = pass == 0 ? row : Hp - 1 - row;
            const int pixl = y * Wp + x;
            load_words<DPL>(cr_f + (pixl - dbase - (DPL - 1)), o.w);
            o.clv = cl_f[pixl];
            o.I = gray_f[pixl];
            if (row > 0) {
                const u8 *gp = gray_f + (y - dy) * Wp + x;
                o.Ip[0] = gp[0];
                o.Ip[1] = x > 0 ? gp[-1] : 0;
                o.Ip[2] = x < Wp - 1 ? gp[1] : 0;
            }
        }
    };
    VIn cur, nxt;
    vfetch(0, cur);
    for (int sidx = 0; sidx < 2 * rows_here; sidx++) {
        const int j = sidx >> 1, it = sidx & 1;
        vfetch(sidx + 1, nxt);
        const int row = a.band * VR + j;
        const int y = pass == 0 ? row : Hp - 1 - row;
        const bool first_row = (row == 0);
        const int pbuf = j & 1, cbuf = (j + 1) & 1;
        const int c = j + 1 + g + 48 * it;
        const int x = x0 - VR + c;
        if (cur.valid) { // uniform per 16-lane group
            const int pixl = y * Wp + x;
            const int I = cur.I;
            u32 C[NP];
            {
                const int lim = x - dbase;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    u32 c0 = __popc(cur.clv ^ cur.w[DPL - 1 - 2 * i]);
                    u32 c1 = __popc(cur.clv ^ cur.w[DPL - 2 - 2 * i]);
                    C[i] = (c1 << 16) | c0;
                }
                if (__builtin_amdgcn_ballot_w64(lim < DPL - 1) != 0) {
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        const u32 lo = (2 * i <= lim) ? (C[i] & 0xFFFFu) : INVALID_DISP_COST;
                        const u32 hi = (2 * i + 1 <= lim) ? (C[i] >> 16) : INVALID_DISP_COST;
                        C[i] = lo | (hi << 16);
                    }
                }
            }
            u32 acc[NP];
#pragma unroll
            for (int i = 0; i < NP; i++) acc[i] = 0;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                // q = 0: straight (N / S), 1: predecessor at x-1 (NW / SW), 2: predecessor at x+1 (NE / SE)
                const int off = q == 0 ? 0 : (q == 1 ? -1 : 1);
                const bool reset = first_row || (q == 1 && x == 0) || (q == 2 && x == Wp - 1);
                u32 L[NP];
                u32 minpk = 0, P2pk = 0;
                if (!reset) {
                    const u32 *sp = (const u32 *)(st + ((size_t)(pbuf * 3 + q) * VC + (c + off)) * D + dbase);
#pragma unroll
                    for (int i = 0; i < NW3; i++) {
                        const u32 bw = sp[i];
                        L[2 * i] = __builtin_amdgcn_perm(bw, bw, 0x0c010c00u);
                        L[2 * i + 1] = __builtin_amdgcn_perm(bw, bw, 0x0c030c02u);
                    }
                    minpk = pk_splat(mn[(pbuf * 3 + q) * VC + c + off]);
                    int di = I - cur.Ip[q];
                    di = di < 0 ? -di : di;
                    P2pk = pk_splat(s_lut[di]);
                } else {
#pragma unroll
                    for (int i = 0; i < NP; i++) L[i] = 0;
                }
                sgm_update<NP, true, 16, false>(L, C, C, C, P1pk, P2pk, minpk, inact, l16 == 0, l16 == 15);
                u32 *dp = (u32 *)(st + ((size_t)(cbuf * 3 + q) * VC + c) * D + dbase);
#pragma unroll
                for (int i = 0; i < NW3; i++) dp[i] = __builtin_amdgcn_perm(L[2 * i + 1], L[2 * i], 0x06040200u);
                if (l16 == 0) mn[(cbuf * 3 + q) * VC + c] = (u16)(minpk & 0xFFFFu);
#pragma unroll
                for (int i = 0; i < NP; i++) acc[i] = pk_adds(acc[i], L[i]);
            }
            if (c >= VR && c < VR + VT) { // owned column: one byte per cell for the three paths
                u32 bw[NW3];
#pragma unroll
                for (int i = 0; i < NW3; i++) bw[i] = __builtin_amdgcn_perm(acc[2 * i + 1], acc[2 * i], 0x06040200u);
                store_words_nt<NW3>((u32 *)(sv_f + (size_t)((u32)pixl * (u32)D + (u32)dbase)), bw);
            }
        }
        if (it == 1) __syncthreads();
        cur = nxt;
    }
    // ---- hand the last row's state of the owned columns to the next band ---------------------------
    {
        const int fbuf = rows_here & 1; // buffer written by the last processed row
        const int ndw = D / 4;
        for (int i = threadIdx.x; i < 3 * VT * ndw; i += 768) {
            const int q = i / (VT * ndw), rem = i % (VT * ndw);
            const int c = VR + rem / ndw, wdx = rem % ndw;
            const int x = x0 - VR + c;
            if (x < Wp)
                ((u32 *)(a.gst + (gbase_out + (size_t)q * Wp + x) * D))[wdx] = ((const u32 *)(st + ((size_t)(fbuf * 3 + q) * VC + c) * D))[wdx];
        }
        for (int i = threadIdx.x; i < 3 * VT; i += 768) {
            const int q = i / VT, c = VR + i % VT;
            const int x = x0 - VR + c;
            if (x < Wp) a.gmin[gbase_out + (size_t)q * Wp + x] = mn[(fbuf * 3 + q) * VC + c];
        }
    }
}

bool rsgm_vert_supported(int D, int maxp2) { return (D == 64 || D == 128 || D == 192 || D == 256) && 3 * (24 + maxp2) <= 255; }

size_t rsgm_vert_state_bytes(int B, int Wp, int D) { return (size_t)2 * 2 * B * 3 * Wp * D; }
size_t rsgm_vert_min_elems(int B, int Wp) { return (size_t)2 * 2 * B * 3 * Wp; }

int rsgm_launch_vert(vppx_ctx *ctx, hipStream_t stream, int B, int Hp, int Wp, int D, const u8 *gray, const u32 *cl,
                     const u32 *cr, const u16 *p2lut, int p1, u8 *sv, u8 *gst, u16 *gmin)
{
    VertArgs a;
    a.gray = gray; a.cl = cl; a.cr = cr; a.p2lut = p2lut; a.sv = sv; a.gst = gst; a.gmin = gmin;
    a.B = B; a.Hp = Hp; a.Wp = Wp; a.D = D; a.p1 = p1;
    a.vol_elems = (size_t)B * Hp * Wp * D;
    const int nstrips = (Wp + VT - 1) / VT;
    const int nbands = (Hp + VR - 1) / VR;
    const size_t ldsb = (size_t)2 * 3 * VC * D + (size_t)2 * 3 * VC * 2 + 512;
    dim3 grid(nstrips, 2, B);
#define LAUNCH_V(DPLV)                                                                                        \
    do {                                                                                                      \
        static bool attr_set[VPPX_MAX_DEVICES] = {};                                                          \
        if (!attr_set[ctx->device & (VPPX_MAX_DEVICES - 1)]) {                                                \
            VPPX_HIP(hipFuncSetAttribute((const void *)sgm_vert_kernel<DPLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
            attr_set[ctx->device & (VPPX_MAX_DEVICES - 1)] = true;                                            \
        }                                                                                                     \
        for (int band = 0; band < nbands; band++) {                                                           \
            a.band = band;                                                                                    \
            sgm_vert_kernel<DPLV><<<grid, 768, ldsb, stream>>>(a);                                            \
        }                                                                                                     \
    } while (0)
    if (D == 64) LAUNCH_V(4); else if (D == 128) LAUNCH_V(8); else if (D == 192) LAUNCH_V(12); else LAUNCH_V(16);
#undef LAUNCH_V
    VPPX_CHECK_LAUNCH();
    (void)ctx;
    return 0;
}

// ---------------------------------------------------------------------------------------
// Vertical / diagonal paths three at a time with the state in registers (round 2; byte volumes, D = 8 * DPL).
// One launch runs two passes per frame: pass 0 walks the rows top-down and carries N, NW, NE; pass 1 walks
// bottom-up and carries S, SW, SE.  A wave owns 8 neighbouring columns (8 lanes x 24 disparities per pixel, the
// layout of sgm_paths_kernel<8, 24>) for the whole pass and keeps the three paths' L_r of the previous row in
// VGPRs.  Per row: the census costs are formed ONCE and shared by the three min-plus updates, the straight path
// updates in place, the two diagonal states move one pixel sideways (ds_bpermute inside the wave; the wave's edge
// pixel comes from the neighbouring wave through a small global record served by the XCD's L2), and ONE byte per
// cell (the sum of the three paths, <= 3 * (24 + P2max) <= 255) is stored: 2 volumes instead of 6, and the
// sum/WTA kernel reads 4 volumes instead of 8.
// All waves of a (frame, pass) group advance in lock step: every row needs both neighbours' previous row.  The
// hand-off has no flag and no drain: every 16-byte piece of an edge record carries the row number in the bits the
// values leave free (values < 1024 per half), the reader polls the record itself (sc1 loads: L1 bypassed, `tools/
// pingpong_bench.hip`: 0.4 us per hand-off inside an XCD) until all its dwords show the row it waits for.  The
// records are zeroed before every launch.  Blocks of a group are consecutive in one XCD's dispatch order
// (blockIdx & 7 = XCD) and blocks start in index order, so every block a resident block waits for is resident or
// next in line; polls are bounded all the same.
// ---------------------------------------------------------------------------------------
// Min-plus step of the fused kernel: small-value arithmetic with given packed costs like sgm_update<.., FUSE, PRE>,
// but pair i of a lane holds the disparities (dbase + i, dbase + NP + i) instead of two neighbours.  The d-1 / d+1
// operands of BOTH halves of pair i are then simply pairs i-1 and i+1: no v_alignbit per pair, only one at each end
// of the lane (d-1 of the first half comes from the previous lane, d+1 of the last from the next).
template <int NP, int GW>
__device__ __forceinline__ void sgm_update_split(u32 (&L)[NP], const u32 (&C)[NP], u32 P1pk, u32 P2pk, u32 &minpk, bool first,
                                                 bool last)
{
    constexpr u32 NONE = 0x3FFF3FFFu;
    u32 prev = dpp_keep<0x111>(NONE, L[NP - 1]); // row_shr:1 : lane-1's last pair (its high half is d = dbase - 1)
    u32 next = dpp_keep<0x101>(NONE, L[0]);      // row_shl:1 : lane+1's first pair (its low half is d = dbase + 2 NP)
    if (GW < 16) {
        prev = first ? NONE : prev;
        next = last ? NONE : next;
    }
    const u32 below0 = __builtin_amdgcn_alignbit(L[NP - 1], prev, 16); // {d = dbase - 1, d = dbase + NP - 1}
    const u32 above_last = __builtin_amdgcn_alignbit(next, L[0], 16);  // {d = dbase + NP, d = dbase + 2 NP}
    const u32 t2 = minpk + P2pk;
    u32 m[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] = pk_min(i == 0 ? below0 : L[i - 1], i == NP - 1 ? above_last : L[i + 1]);
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] += P1pk;
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] = pk_min3_small(m[i], L[i], t2);
    // + C - min as ONE three-operand add (v_add3_u32) with the negated packed minimum: every half of m + C is >= the
    // minimum, so no borrow crosses the halves and the 32-bit sum is the packed difference
    u32 negmin = 0u - minpk;
    asm volatile("" : "+v"(negmin)); // (keeps LLVM from turning the sum back into an add and a subtract)
#pragma unroll
    for (int i = 0; i < NP; i++) {
        m[i] = m[i] + C[i] + negmin;
        L[i] = m[i];
    }
#pragma unroll
    for (int n = NP; n > 1; n = (n + 2) / 3)
#pragma unroll
        for (int i = 0; 3 * i < n; i++) {
            if (3 * i + 2 < n) m[i] = pk_min3_small(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
            else if (3 * i + 1 < n) m[i] = pk_min(m[3 * i], m[3 * i + 1]);
            else m[i] = m[3 * i];
        }
    u32 mm = min(m[0] & 0xFFFFu, m[0] >> 16);
    mm = grp_min_u32<GW>(mm);
    minpk = pk_splat(mm);
}

struct Vert3Args {
    const u8 *gray;
    const u32 *cl;
    const u32 *cr;
    const u16 *p2lut;
    u8 *sv;     // [2 passes][B][Hp][Wp][D]
    u32 *xbuf;  // [2B groups][nwv][V3_RING rows][2 directions][8 lanes][XW] edge records
    unsigned *err; // host-visible word, set (to the launch serial) when a wave gave up waiting for a neighbour (results void)
    unsigned *err_dev; // the same in device memory: what the launches queued behind this one look at (void_if_lost_kernel)
    int B, Hp, Wp, p1;
    int nwv, nbg; // waves / blocks per group
    size_t vol_elems;
    // bounds of one wait for a neighbour's record: wall-clock ticks (s_memrealtime, constant rate: the bound that
    // counts) and, for tests, a number of polls (0 = unbounded: VPPX_V3_SPIN_LIMIT)
    long long timeout_ticks;
    unsigned spin_limit;
    unsigned serial; // launch serial written to *err (never 0)
};
// Rows of edge records kept per wave.  A wave may write the record of row t+1 while a neighbour is still reading the
// one of row t-1 (the NW record of a row is published before the NE record of the row before has been consumed), so
// the slot reused must be at least three rows old: a ring of four.
#define V3_RING 4
#define V3_TAGMASK 0xFC00FC00u
#ifndef V3_TAG_BOTH
#define V3_TAG_BOTH 0
#endif
#if V3_TAG_BOTH
#define V3_TAG_LAST(v, t) ((v) | (t))
#else
#define V3_TAG_LAST(v, t) (v)
#endif

// The edge records of row t-1 (tag word T): group-0 lanes read the left wave's NW record, group-7 lanes the right
// wave's NE record (a lane reads at most one; lanes without a record read a record of zeros).  Buffer loads with the
// sc1 bit (device scope: L1 bypassed, served by the XCD's L2) that the compiler can see, so it keeps their results
// where they land and waits for them where they are used: v3_edges_issue starts them, the straight path's update
// runs, v3_edges_complete checks the tags, polls again while a record is not there yet, and strips the tags.
// A tag on the first dword of every 16-byte piece and on the minimum.  A lane's aligned 16-byte store and 16-byte load are single
// requests to one cache line: a piece is seen whole or not at all -- until round 6 both ends of a piece were tagged and checked all
// the same; `tools/tear_bench.hip` then read 6.5e9 pieces (6e8 of them freshly written) through this very store / load pair without
// one torn piece, while its control -- four dword stores -- tears one piece in a hundred.  -DV3_TAG_BOTH=1 builds the old check.
template <int NP>
struct V3Edge {
    u32x4 p[NP / 4]; // the NP pairs in 16-byte pieces
    u32 m;           // the packed minimum
};
#define V3_SC1 16 // cache-policy operand of the buffer load builtins: bit 4 = sc1 on gfx940+
template <int NP>
__device__ __forceinline__ void v3_edges_issue(__amdgpu_buffer_rsrc_t xr, int off, V3Edge<NP> &r)
{
#pragma unroll
    for (int k = 0; k < NP / 4; k++) r.p[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, off + 16 * k, 0, V3_SC1);
    r.m = __builtin_amdgcn_raw_buffer_load_b32(xr, off + 4 * NP, 0, V3_SC1);
}
template <int NP>
__device__ __forceinline__ void v3_edges_complete(__amdgpu_buffer_rsrc_t xr, int off, bool want, u32 T, V3Edge<NP> &r, bool &dead,
                                                  long long timeout_ticks, unsigned spin_limit)
{
    // The wait is bounded in TIME (s_memrealtime runs at a constant rate whatever the shader clock and the L2 latency
    // are; looked at every 32nd poll) and, when a test asks for it, in polls.  A wave that gave up once does not wait
    // again: its results are void anyway, and its neighbours must not inherit a full timeout per row.
    unsigned spins = 0;
    long long t0 = 0;
    for (;;) {
        u32 bad = r.m ^ T;
#pragma unroll
        for (int k = 0; k < NP / 4; k++) bad |= V3_TAG_BOTH ? (r.p[k].x ^ T) | (r.p[k].w ^ T) : (r.p[k].x ^ T);
        const bool ok = !want || (bad & V3_TAGMASK) == 0;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        ++spins;
        if (dead || (spin_limit && spins >= spin_limit)) {
            dead = true;
            break;
        }
        if ((spins & 31u) == 1u) {
            const long long now = (long long)__builtin_amdgcn_s_memrealtime();
            if (spins == 1u) t0 = now;
            else if (now - t0 > timeout_ticks) {
                dead = true;
                break;
            }
        }
        __builtin_amdgcn_s_sleep(1);
        v3_edges_issue<NP>(xr, off, r);
    }
#pragma unroll
    for (int k = 0; k < NP / 4; k++) {
        r.p[k].x &= ~V3_TAGMASK;
        if (V3_TAG_BOTH) r.p[k].w &= ~V3_TAGMASK;
    }
    r.m &= ~V3_TAGMASK;
}

// dwords per lane of an edge record (NP pairs + the packed minimum, rounded up to whole 64-byte slots)
template <int NP>
struct V3Rec { static constexpr int XW = (NP + 1 <= 16) ? 16 : (NP + 1 <= 32 ? 32 : 64); };

template <int DPL>
__global__ void __launch_bounds__(256) sgm_vert3_kernel(Vert3Args a)
{
    constexpr int NP = DPL / 2, D = 8 * DPL, V3_XW = V3Rec<NP>::XW;
    static_assert(NP % 4 == 0, "edge records are made of 16-byte pieces");
    constexpr int TRW = (DPL == 24) ? TR_WORDS : (DPL == 32 ? 512 : 1); // store transposition words per wave
    __shared__ __attribute__((aligned(16))) u32 s_lut[256 + 4 * TRW];
    __shared__ u32 s_mask[NP * 256]; // [pair][thread]: validity masks of the lanes whose column is below D - 1
    s_lut[threadIdx.x] = pk_splat(a.p2lut[threadIdx.x]);
    __syncthreads();
    const int id = blockIdx.x, xcd = id & 7, jb = id >> 3;
    const int group = (jb / a.nbg) * 8 + xcd, bi = jb % a.nbg;
    if (group >= 2 * a.B) return; // the grid is rounded up to whole rounds of 8 groups (one per XCD)
    const int f = group >> 1, pass = group & 1;
    const int wv = bi * 4 + (int)(threadIdx.x >> 6);
    if (wv >= a.nwv) return;
    const int lane = threadIdx.x & 63, g = lane >> 3, lg = lane & 7;
    const int x = wv * 8 + g, dbase = DPL * lg;
    const int Wp = a.Wp, Hp = a.Hp;
    const size_t fpix = (size_t)f * Hp * Wp;
    const u8 *gray_f = a.gray + fpix;
    const u32 *cl_f = a.cl + fpix;
    const u32 *cr_f = a.cr + fpix;
    u8 *sv_f = a.sv + (size_t)pass * a.vol_elems + fpix * D;
    u32 *tr = s_lut + 256 + (threadIdx.x >> 6) * TRW;
    const u32 P1pk = pk_splat(a.p1 > 65535 ? 65535u : (u32)(a.p1 < 0 ? 0 : a.p1));
    const bool first = lg == 0, last = lg == 7;
    const bool has_left = wv > 0, has_right = wv + 1 < a.nwv;
    constexpr int REC = V3_RING * 2 * 8 * V3_XW; // dwords per wave: [row & 3][direction][lane]
    bool dead = false;
    u32 *xb_own = a.xbuf + ((size_t)group * a.nwv + wv) * REC;
    // group-0 lanes take the left wave's NW edge (direction 0), group-7 lanes the right wave's NE edge (direction 1)
    const bool edge_lane = (g == 0 && has_left) || (g == 7 && has_right);
    // (byte offsets into the record buffer, read through a buffer resource)
    const int xb_in_off = (int)(((size_t)group * a.nwv + (g == 0 ? (has_left ? wv - 1 : wv) : (has_right ? wv + 1 : wv))) * REC +
                                ((g == 0 ? 0 : 1) * 8 + lg) * V3_XW) * 4;
    const int zero_off = (int)((size_t)2 * a.B * a.nwv * REC) * 4; // one record nobody writes
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, zero_off + V3_XW * 4, 0x00020000);
    const int bp_left = ((lane - 8) & 63) << 2, bp_right = ((lane + 8) & 63) << 2;
    // Some d > x in this wave (the leftmost 24 waves of a pass): those cells cost InvalidDispCost.  A lane's column never
    // changes, so which halves of its pairs are valid is a per-lane constant: 12 mask words per lane, kept in LDS (the
    // lock step makes the whole group run at the pace of these waves: their extra work per row counts 120-fold).
    const bool masked = wv * 8 < D - 1;
    if (masked) {
        const int lim = x - dbase;
#pragma unroll
        for (int i = 0; i < NP; i++) s_mask[i * 256 + threadIdx.x] = (i <= lim ? 0xFFFFu : 0u) | (NP + i <= lim ? 0xFFFF0000u : 0u);
    }
    u32 inact[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) inact[i] = 0;

    u32 L0[NP], L1[NP], L2[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) L0[i] = L1[i] = L2[i] = 0;
    u32 mn0 = 0, mn1 = 0, mn2 = 0;
    int prevI = 0;
    const int dy = pass == 0 ? 1 : -1;
    int y = pass == 0 ? 0 : Hp - 1;
    const int xl = x > 0 ? x - 1 : 0, xr = x < Wp - 1 ? x + 1 : Wp - 1;
    StepIn<DPL, false> A;
    load_step<DPL, true, false>(A, gray_f, cl_f, cr_f, nullptr, y * Wp + x, D, dbase, inact);
    int Ipl = 0, Ipr = 0;
    // the row's bytes leave one row late (right after the next row's poll), so that no wait of this wave ever has a
    // young volume store in front of it
    // (DPL 24 / 32: two pieces per lane out of the LDS transposition; DPL 8 / 16: the lane's own 8 / 16 bytes, which
    // already tile whole lines)
    u32x4 pend_a = {0, 0, 0, 0}, pend_b = {0, 0, 0, 0};
    u32 pend_oa = ((u32)(y * Wp + wv * 8) * (u32)D) + (u32)lane * (DPL < 24 ? DPL : 16), pend_ob = pend_oa; // harmless first store (rewritten)
    auto store_pend = [&]() {
        if constexpr (DPL == 8) {
            __builtin_nontemporal_store(u32x2{pend_a.x, pend_a.y}, (u32x2 *)(sv_f + pend_oa));
        } else if constexpr (DPL == 16) {
            __builtin_nontemporal_store(pend_a, (u32x4 *)(sv_f + pend_oa));
        } else if constexpr (DPL == 24) {
            __builtin_nontemporal_store(pend_a, (u32x4 *)(sv_f + pend_oa));
            __builtin_nontemporal_store(u32x2{pend_b.x, pend_b.y}, (u32x2 *)(sv_f + pend_ob));
        } else {
            __builtin_nontemporal_store(pend_a, (u32x4 *)(sv_f + pend_oa));
            __builtin_nontemporal_store(pend_b, (u32x4 *)(sv_f + pend_ob));
        }
    };
    for (int t = 0; t < Hp; t++, y += dy) {
        const int pixl = y * Wp + x;
        // ---- costs of this row, once for the three paths
        // (pair i of a lane = disparities dbase + i and dbase + 12 + i: sgm_update_split)
        u32 C[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const u32 c0 = __popc(A.clv ^ A.w[DPL - 1 - i]);
            const u32 c1 = __popc(A.clv ^ A.w[DPL - 1 - NP - i]);
            C[i] = (c1 << 16) | c0;
        }
        if (masked) { // one bit-field insert per pair: keep the halves with d <= x, InvalidDispCost in the others
            constexpr u32 INVpk = INVALID_DISP_COST | (INVALID_DISP_COST << 16);
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const u32 m = s_mask[i * 256 + threadIdx.x];
                C[i] = (m & C[i]) | (~m & INVpk);
            }
        }
        const int I = A.I;
        const u32 P2a = s_lut[__builtin_amdgcn_sad_u8((u32)I, (u32)prevI, 0u)];
        const u32 P2b = s_lut[__builtin_amdgcn_sad_u8((u32)I, (u32)Ipl, 0u)];
        const u32 P2c = s_lut[__builtin_amdgcn_sad_u8((u32)I, (u32)Ipr, 0u)];
        // tag words of the records read now (row t-1) and written now (row t): row + 1 in bits 10-15 / 26-31
        const u32 Tin = (((u32)t & 63u) << 10) | (((u32)t >> 6) << 26);
        const u32 Tout = (((u32)(t + 1) & 63u) << 10) | (((u32)(t + 1) >> 6) << 26);
        const int par_in = (t - 1) & (V3_RING - 1), par_out = t & (V3_RING - 1);
        // ---- the neighbours' edges of the previous row: one round trip to L2, covered by the straight path's update
        // lanes that take no record (and everybody in the first row) keep zeros: the path starts / restarts there
        V3Edge<NP> er;
        const bool want = edge_lane && t > 0;
        const int eoff = want ? xb_in_off + par_in * (2 * 8 * V3_XW * 4) : zero_off;
        v3_edges_issue<NP>(xrsrc, eoff, er);
        // ---- straight path: needs nobody else's state
        sgm_update_split<NP, 8>(L0, C, P1pk, P2a, mn0, first, last);
        v3_edges_complete<NP>(xrsrc, eoff, want, Tin, er, dead, a.timeout_ticks, a.spin_limit);
        __builtin_amdgcn_s_setprio(2); // from here to the second publish the neighbours wait for this wave
        {
            u32 e[NP + 1];
#pragma unroll
            for (int q = 0; q < NP / 4; q++) {
                e[4 * q] = er.p[q].x; e[4 * q + 1] = er.p[q].y; e[4 * q + 2] = er.p[q].z; e[4 * q + 3] = er.p[q].w;
            }
            e[NP] = er.m;
            // the two diagonal states move one pixel sideways: inside the wave by ds_bpermute, at its ends from the records.
            // The edge lanes take the record with MOVES UNDER AN EXEC MASK (two divergent regions of 13 full-rate v_mov each)
            // rather than 26 v_cndmask, which issue at half rate on gfx950; the empty asm keeps hipcc from turning the
            // regions back into selects.
#pragma unroll
            for (int i = 0; i < NP; i++) {
                L1[i] = (u32)__builtin_amdgcn_ds_bpermute(bp_left, (int)L1[i]);
                L2[i] = (u32)__builtin_amdgcn_ds_bpermute(bp_right, (int)L2[i]);
            }
            mn1 = (u32)__builtin_amdgcn_ds_bpermute(bp_left, (int)mn1);
            mn2 = (u32)__builtin_amdgcn_ds_bpermute(bp_right, (int)mn2);
            if (g == 0) {
                asm volatile("; NW edge from the left wave's record");
#pragma unroll
                for (int i = 0; i < NP; i++) L1[i] = e[i];
                mn1 = e[NP];
            }
            if (g == 7) {
                asm volatile("; NE edge from the right wave's record");
#pragma unroll
                for (int i = 0; i < NP; i++) L2[i] = e[i];
                mn2 = e[NP];
            }
        }
        // ---- last row's bytes (everything older has just been drained: this store has a whole row to complete)
        store_pend();
        // ---- NW / SW, then NE / SE; each publishes its edge pixel as soon as it is known
        sgm_update_split<NP, 8>(L1, C, P1pk, P2b, mn1, first, last);
        if (t + 1 < Hp && g == 7 && has_right) { // the last pixel's state goes to the right wave
            u32 *q = xb_own + par_out * (2 * 8 * V3_XW) + (0 * 8 + lg) * V3_XW;
#pragma unroll
            for (int i = 0; i < NP; i += 4) *(u32x4 *)(q + i) = u32x4{L1[i] | Tout, L1[i + 1], L1[i + 2], V3_TAG_LAST(L1[i + 3], Tout)};
            q[NP] = mn1 | Tout;
        }
        sgm_update_split<NP, 8>(L2, C, P1pk, P2c, mn2, first, last);
        if (t + 1 < Hp && g == 0 && has_left) { // the first pixel's state goes to the left wave
            u32 *q = xb_own + par_out * (2 * 8 * V3_XW) + (1 * 8 + lg) * V3_XW;
#pragma unroll
            for (int i = 0; i < NP; i += 4) *(u32x4 *)(q + i) = u32x4{L2[i] | Tout, L2[i + 1], L2[i + 2], V3_TAG_LAST(L2[i + 3], Tout)};
            q[NP] = mn2 | Tout;
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- operands of the next row.  Issued this late on purpose: in flight during the updates they cost 26 more
        // live VGPRs (156 instead of 127, 3 waves per SIMD instead of 4); the shorter distance to their use is covered
        // by the fourth wave
        {
            const int yn = (t + 1 < Hp) ? y + dy : y;
            load_step<DPL, true, false>(A, gray_f, cl_f, cr_f, nullptr, yn * Wp + x, D, dbase, inact);
            Ipl = gray_f[y * Wp + xl];
            Ipr = gray_f[y * Wp + xr];
        }
        // ---- one byte per cell: the three paths summed, transposed through LDS into whole-line pieces (store_step)
        {
            // bytes in disparity order: the low halves of the 12 pairs, then the high halves
            u32 tq[NP / 2], bw[NP / 2];
#pragma unroll
            for (int k = 0; k < NP / 2; k++) // {lo(2k), lo(2k+1), hi(2k), hi(2k+1)} as bytes
                tq[k] = __builtin_amdgcn_perm(L0[2 * k + 1] + L1[2 * k + 1] + L2[2 * k + 1], L0[2 * k] + L1[2 * k] + L2[2 * k], 0x06020400u);
#pragma unroll
            for (int j = 0; j < NP / 4; j++) {
                bw[j] = __builtin_amdgcn_perm(tq[2 * j + 1], tq[2 * j], 0x05040100u);
                bw[NP / 4 + j] = __builtin_amdgcn_perm(tq[2 * j + 1], tq[2 * j], 0x07060302u);
            }
            const u32 base = (u32)(y * Wp + wv * 8) * (u32)D; // the wave's 8 pixels are neighbours in memory
            if constexpr (DPL == 8) {
                pend_a = u32x4{bw[0], bw[1], 0, 0};
                pend_oa = base + (u32)lane * 8u;
            } else if constexpr (DPL == 16) {
                pend_a = u32x4{bw[0], bw[1], bw[2], bw[3]};
                pend_oa = base + (u32)lane * 16u;
            } else if constexpr (DPL == 24) { // bytes [0, 1024) and [1024, 1536) of the wave's 8 x 192 (store_step)
                u32x2 *wp = (u32x2 *)(tr + lane * 6);
                wp[0] = u32x2{bw[0], bw[1]};
                wp[1] = u32x2{bw[2], bw[3]};
                wp[2] = u32x2{bw[4], bw[5]};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                pend_a = *(const u32x4 *)(tr + lane * 4);
                const u32x2 pb = *(const u32x2 *)(tr + 256 + lane * 2);
                pend_b = u32x4{pb.x, pb.y, 0, 0};
                pend_oa = base + (u32)lane * 16u;
                pend_ob = base + 1024u + (u32)lane * 8u;
            } else { // 32 bytes per lane: bytes [0, 1024) and [1024, 2048) of the wave's 8 x 256
                u32x4 *wp = (u32x4 *)(tr + lane * 8);
                wp[0] = u32x4{bw[0], bw[1], bw[2], bw[3]};
                wp[1] = u32x4{bw[4], bw[5], bw[6], bw[7]};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                pend_a = *(const u32x4 *)(tr + lane * 4);
                pend_b = *(const u32x4 *)(tr + 256 + lane * 4);
                pend_oa = base + (u32)lane * 16u;
                pend_ob = base + 1024u + (u32)lane * 16u;
            }
        }
        prevI = I;
    }
    store_pend();
    if (dead && lane == 0 && a.err) {
        __hip_atomic_store(a.err, a.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.err_dev) __hip_atomic_store(a.err_dev, a.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------
// The same fused three-path pass with 16 pixels per wave: 4 lanes x DPL disparities per pixel (round 3; D = 4 * DPL = 192).
// Everything on gfx950 that this path runs is bound by VALU issue (tools/agg_probe.py: the 8 x 24 kernel takes the same
// time per frame at 2 waves per SIMD as at 4, and 91 % of its time with every neighbour wait removed), so what counts is
// instructions per pixel, and those are dominated by what does NOT scale with the disparities a lane holds: the min
// reduction across a pixel's lanes, the sideways move of the diagonal states, tags, P2 look-ups, addressing.  Twice the
// disparities per lane halve that share, and the lane layout lane = 16 * chunk + pixel makes the sideways move one
// DPP row_shr:1 / row_shl:1 per register whose `old` operand IS the neighbour wave's edge record (lane 0 / 15 of a row
// has no source and keeps it): no ds_bpermute, no select (the 8 x 24 kernel spends 26 + 26 of its 404 instructions per
// row there, and its LDS crossbar is 35 % busy with them).  What crosses DPP rows is small: the d +- 1 operands at a
// chunk's ends (2 ds_bpermute per path) and the minimum over the 4 chunks (2 per path).  ~200 VGPRs: 2 waves per SIMD.
// Edge records, tags, lock step, store transposition: as in sgm_vert3_kernel.
// ---------------------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ u32 sgm_update_split_pn(u32 (&L)[NP], const u32 (&C)[NP], u32 P1pk, u32 P2pk, u32 minpk, u32 prev, u32 next)
{
    // prev = the pair NP-1 of the lane that holds the next lower disparities (its high half is d = dbase - 1), next = pair 0
    // of the lane above (its low half is d = dbase + 2 NP); 0x3FFF3FFF where there is none.  Returns the lane's own minimum.
    const u32 below0 = __builtin_amdgcn_alignbit(L[NP - 1], prev, 16);
    const u32 above_last = __builtin_amdgcn_alignbit(next, L[0], 16);
    const u32 t2 = minpk + P2pk;
    u32 negmin = 0u - minpk;
    asm volatile("" : "+v"(negmin));
    u32 m[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] = pk_min(i == 0 ? below0 : L[i - 1], i == NP - 1 ? above_last : L[i + 1]);
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] += P1pk;
#pragma unroll
    for (int i = 0; i < NP; i++) m[i] = pk_min3_small(m[i], L[i], t2);
#pragma unroll
    for (int i = 0; i < NP; i++) {
        m[i] = m[i] + C[i] + negmin;
        L[i] = m[i];
    }
#pragma unroll
    for (int n = NP; n > 1; n = (n + 2) / 3)
#pragma unroll
        for (int i = 0; 3 * i < n; i++) {
            if (3 * i + 2 < n) m[i] = pk_min3_small(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
            else if (3 * i + 1 < n) m[i] = pk_min(m[3 * i], m[3 * i + 1]);
            else m[i] = m[3 * i];
        }
    return min(m[0] & 0xFFFFu, m[0] >> 16);
}

template <int DPL>
__global__ void __launch_bounds__(256, 2) sgm_vert4_kernel(Vert3Args a)
{
    constexpr int NP = DPL / 2, D = 4 * DPL, V3_XW = V3Rec<NP>::XW;
    static_assert(NP % 4 == 0, "edge records are made of 16-byte pieces");
    static_assert((DPL * 4) % 16 == 0 && DPL % 16 == 0, "a lane's bytes are whole 16-byte pieces");
    constexpr int TRW = 16 * D / 4; // store transposition words per wave: 16 pixels x D bytes
    constexpr int NQ = DPL / 16;    // 16-byte pieces of a lane's DPL bytes
    constexpr u32 NONE = 0x3FFF3FFFu;
    __shared__ __attribute__((aligned(16))) u32 s_lut[256 + 4 * TRW];
    __shared__ u32 s_mask[NP * 256]; // [pair][thread]: validity masks of the lanes whose column is below D - 1
    s_lut[threadIdx.x] = pk_splat(a.p2lut[threadIdx.x]);
    __syncthreads();
    const int id = blockIdx.x, xcd = id & 7, jb = id >> 3;
    const int group = (jb / a.nbg) * 8 + xcd, bi = jb % a.nbg;
    if (group >= 2 * a.B) return; // the grid is rounded up to whole rounds of 8 groups (one per XCD)
    const int f = group >> 1, pass = group & 1;
    const int wv = bi * 4 + (int)(threadIdx.x >> 6);
    if (wv >= a.nwv) return;
    const int lane = threadIdx.x & 63, g = lane & 15, lg = lane >> 4;
    const int x = wv * 16 + g, dbase = DPL * lg;
    const int Wp = a.Wp, Hp = a.Hp;
    const size_t fpix = (size_t)f * Hp * Wp;
    const u8 *gray_f = a.gray + fpix;
    const u32 *cl_f = a.cl + fpix;
    const u32 *cr_f = a.cr + fpix;
    u8 *sv_f = a.sv + (size_t)pass * a.vol_elems + fpix * D;
    u32 *tr = s_lut + 256 + (threadIdx.x >> 6) * TRW;
    const u32 P1pk = pk_splat(a.p1 > 65535 ? 65535u : (u32)(a.p1 < 0 ? 0 : a.p1));
    const bool first = lg == 0, last = lg == 3;
    const bool has_left = wv > 0, has_right = wv + 1 < a.nwv;
    constexpr int REC = V3_RING * 2 * 4 * V3_XW; // dwords per wave: [row & 3][direction][chunk lane]
    bool dead = false;
    u32 *xb_own = a.xbuf + ((size_t)group * a.nwv + wv) * REC;
    // pixel-0 lanes take the left wave's NW edge (direction 0), pixel-15 lanes the right wave's NE edge (direction 1)
    const bool edge_lane = (g == 0 && has_left) || (g == 15 && has_right);
    const int xb_in_off = (int)(((size_t)group * a.nwv + (g == 0 ? (has_left ? wv - 1 : wv) : (has_right ? wv + 1 : wv))) * REC +
                                ((g == 0 ? 0 : 1) * 4 + lg) * V3_XW) * 4;
    const int zero_off = (int)((size_t)2 * a.B * a.nwv * REC) * 4; // one record nobody writes
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, zero_off + V3_XW * 4, 0x00020000);
    const int bp_prev = ((lane - 16) & 63) << 2, bp_next = ((lane + 16) & 63) << 2;
    const int bp_x16 = (lane ^ 16) << 2, bp_x32 = (lane ^ 32) << 2;
    const bool masked = wv * 16 < D - 1;
    if (masked) {
        const int lim = x - dbase;
#pragma unroll
        for (int i = 0; i < NP; i++) s_mask[i * 256 + threadIdx.x] = (i <= lim ? 0xFFFFu : 0u) | (NP + i <= lim ? 0xFFFF0000u : 0u);
    }
    u32 inact[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) inact[i] = 0;

    u32 L0[NP], L1[NP], L2[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) L0[i] = L1[i] = L2[i] = 0;
    u32 mn0 = 0, mn1 = 0, mn2 = 0;
    u32 prev0 = NONE, next0 = NONE; // the straight path's operands from the neighbouring chunks (fetched a row ahead)
    int prevI = 0;
    const int dy = pass == 0 ? 1 : -1;
    int y = pass == 0 ? 0 : Hp - 1;
    const int xl = x > 0 ? x - 1 : 0, xr = x < Wp - 1 ? x + 1 : Wp - 1;
    StepIn<DPL, false> A;
    load_step<DPL, true, false>(A, gray_f, cl_f, cr_f, nullptr, y * Wp + x, D, dbase, inact);
    int Ipl = 0, Ipr = 0;
    u32x4 pend[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) pend[k] = u32x4{0, 0, 0, 0};
    u32 pend_o = ((u32)(y * Wp + wv * 16) * (u32)D) + (u32)lane * 16u; // harmless first store (rewritten)
    auto store_pend = [&]() {
#pragma unroll
        for (int k = 0; k < NQ; k++) __builtin_nontemporal_store(pend[k], (u32x4 *)(sv_f + pend_o + (u32)k * 1024u));
    };
    // minimum over the 4 chunk lanes of a pixel (16 and 32 lanes apart: two DPP rows, two halves of the wave)
    auto pixel_min = [&](u32 mm) -> u32 {
        mm = min(mm, (u32)__builtin_amdgcn_ds_bpermute(bp_x16, (int)mm));
        mm = min(mm, (u32)__builtin_amdgcn_ds_bpermute(bp_x32, (int)mm));
        return pk_splat(mm);
    };
    for (int t = 0; t < Hp; t++, y += dy) {
        // ---- costs of this row, once for the three paths (pair i of a lane = disparities dbase + i and dbase + NP + i)
        u32 C[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const u32 c0 = __popc(A.clv ^ A.w[DPL - 1 - i]);
            const u32 c1 = __popc(A.clv ^ A.w[DPL - 1 - NP - i]);
            C[i] = (c1 << 16) | c0;
        }
        if (masked) {
            constexpr u32 INVpk = INVALID_DISP_COST | (INVALID_DISP_COST << 16);
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const u32 m = s_mask[i * 256 + threadIdx.x];
                C[i] = (m & C[i]) | (~m & INVpk);
            }
        }
        const int I = A.I;
        const u32 P2a = s_lut[__builtin_amdgcn_sad_u8((u32)I, (u32)prevI, 0u)];
        const u32 P2b = s_lut[__builtin_amdgcn_sad_u8((u32)I, (u32)Ipl, 0u)];
        const u32 P2c = s_lut[__builtin_amdgcn_sad_u8((u32)I, (u32)Ipr, 0u)];
        const u32 Tin = (((u32)t & 63u) << 10) | (((u32)t >> 6) << 26);
        const u32 Tout = (((u32)(t + 1) & 63u) << 10) | (((u32)(t + 1) >> 6) << 26);
        const int par_in = (t - 1) & (V3_RING - 1), par_out = t & (V3_RING - 1);
        // ---- the neighbours' edges of the previous row: one round trip to L2, covered by the straight path's update
        V3Edge<NP> er;
        const bool want = edge_lane && t > 0;
        const int eoff = want ? xb_in_off + par_in * (2 * 4 * V3_XW * 4) : zero_off;
        v3_edges_issue<NP>(xrsrc, eoff, er);
        // ---- straight path
        {
            const u32 mm = sgm_update_split_pn<NP>(L0, C, P1pk, P2a, mn0, prev0, next0);
            // next row's end operands and this row's minimum: LDS crossbar round trips that nobody waits for before the next row
            const u32 pv = (u32)__builtin_amdgcn_ds_bpermute(bp_prev, (int)L0[NP - 1]);
            const u32 nx = (u32)__builtin_amdgcn_ds_bpermute(bp_next, (int)L0[0]);
            mn0 = pixel_min(mm);
            prev0 = first ? NONE : pv;
            next0 = last ? NONE : nx;
        }
        v3_edges_complete<NP>(xrsrc, eoff, want, Tin, er, dead, a.timeout_ticks, a.spin_limit);
        __builtin_amdgcn_s_setprio(2); // from here to the second publish the neighbours wait for this wave
        u32 prev1, next1, prev2, next2;
        {
            u32 e[NP + 1];
#pragma unroll
            for (int q = 0; q < NP / 4; q++) {
                e[4 * q] = er.p[q].x; e[4 * q + 1] = er.p[q].y; e[4 * q + 2] = er.p[q].z; e[4 * q + 3] = er.p[q].w;
            }
            e[NP] = er.m;
            // the two diagonal states move one pixel sideways: DPP inside the 16 pixels of a row of lanes; the lane without a
            // source (pixel 0 / pixel 15) keeps `old`, which is the record it has just read (zeros where the path restarts).
            // The registers the chunk-end operands come from move first, so that their crossbar trips overlap the other moves.
            L1[NP - 1] = dpp_keep<0x111>(e[NP - 1], L1[NP - 1]);
            L1[0] = dpp_keep<0x111>(e[0], L1[0]);
            prev1 = (u32)__builtin_amdgcn_ds_bpermute(bp_prev, (int)L1[NP - 1]);
            next1 = (u32)__builtin_amdgcn_ds_bpermute(bp_next, (int)L1[0]);
            L2[NP - 1] = dpp_keep<0x101>(e[NP - 1], L2[NP - 1]);
            L2[0] = dpp_keep<0x101>(e[0], L2[0]);
            prev2 = (u32)__builtin_amdgcn_ds_bpermute(bp_prev, (int)L2[NP - 1]);
            next2 = (u32)__builtin_amdgcn_ds_bpermute(bp_next, (int)L2[0]);
#pragma unroll
            for (int i = 1; i < NP - 1; i++) {
                L1[i] = dpp_keep<0x111>(e[i], L1[i]); // row_shr:1 : pixel g takes pixel g - 1
                L2[i] = dpp_keep<0x101>(e[i], L2[i]); // row_shl:1 : pixel g takes pixel g + 1
            }
            mn1 = dpp_keep<0x111>(e[NP], mn1);
            mn2 = dpp_keep<0x101>(e[NP], mn2);
            prev1 = first ? NONE : prev1; next1 = last ? NONE : next1;
            prev2 = first ? NONE : prev2; next2 = last ? NONE : next2;
        }
        // ---- last row's bytes
        store_pend();
        // ---- NW / SW, then NE / SE; each publishes its edge pixel as soon as it is known
        {
            const u32 mm = sgm_update_split_pn<NP>(L1, C, P1pk, P2b, mn1, prev1, next1);
            mn1 = pixel_min(mm);
        }
        if (t + 1 < Hp && g == 15 && has_right) { // the last pixel's state goes to the right wave
            u32 *q = xb_own + par_out * (2 * 4 * V3_XW) + (0 * 4 + lg) * V3_XW;
#pragma unroll
            for (int i = 0; i < NP; i += 4) *(u32x4 *)(q + i) = u32x4{L1[i] | Tout, L1[i + 1], L1[i + 2], V3_TAG_LAST(L1[i + 3], Tout)};
            q[NP] = mn1 | Tout;
        }
        {
            const u32 mm = sgm_update_split_pn<NP>(L2, C, P1pk, P2c, mn2, prev2, next2);
            mn2 = pixel_min(mm);
        }
        if (t + 1 < Hp && g == 0 && has_left) { // the first pixel's state goes to the left wave
            u32 *q = xb_own + par_out * (2 * 4 * V3_XW) + (1 * 4 + lg) * V3_XW;
#pragma unroll
            for (int i = 0; i < NP; i += 4) *(u32x4 *)(q + i) = u32x4{L2[i] | Tout, L2[i + 1], L2[i + 2], V3_TAG_LAST(L2[i + 3], Tout)};
            q[NP] = mn2 | Tout;
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- operands of the next row
        {
            const int yn = (t + 1 < Hp) ? y + dy : y;
            load_step<DPL, true, false>(A, gray_f, cl_f, cr_f, nullptr, yn * Wp + x, D, dbase, inact);
            Ipl = gray_f[y * Wp + xl];
            Ipr = gray_f[y * Wp + xr];
        }
        // ---- one byte per cell: the three paths summed; a lane's DPL bytes in disparity order = the low halves of its
        // pairs, then the high halves; transposed through LDS so that every store instruction writes whole lines
        {
            u32 tq[NP / 2], bw[NP / 2];
#pragma unroll
            for (int k = 0; k < NP / 2; k++) // {lo(2k), lo(2k+1), hi(2k), hi(2k+1)} as bytes
                tq[k] = __builtin_amdgcn_perm(L0[2 * k + 1] + L1[2 * k + 1] + L2[2 * k + 1], L0[2 * k] + L1[2 * k] + L2[2 * k], 0x06020400u);
#pragma unroll
            for (int j = 0; j < NP / 4; j++) {
                bw[j] = __builtin_amdgcn_perm(tq[2 * j + 1], tq[2 * j], 0x05040100u);
                bw[NP / 4 + j] = __builtin_amdgcn_perm(tq[2 * j + 1], tq[2 * j], 0x07060302u);
            }
            u32x4 *wp = (u32x4 *)(tr + g * (D / 4) + lg * (DPL / 4)); // pixel-major: pixel g, bytes [dbase, dbase + DPL)
#pragma unroll
            for (int k = 0; k < NQ; k++) wp[k] = u32x4{bw[4 * k], bw[4 * k + 1], bw[4 * k + 2], bw[4 * k + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < NQ; k++) pend[k] = *(const u32x4 *)(tr + k * 256 + lane * 4);
            pend_o = (u32)(y * Wp + wv * 16) * (u32)D + (u32)lane * 16u; // the wave's 16 pixels are neighbours in memory
        }
        prevI = I;
    }
    store_pend();
    if (dead && lane == 0 && a.err) {
        __hip_atomic_store(a.err, a.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.err_dev) __hip_atomic_store(a.err_dev, a.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The lock-step hand-off goes through ONE L2: all blocks of a (frame, pass) group must run on the same XCD, which the
// kernel gets from "consecutive block ids go round-robin over the XCDs" (group blocks are 8 ids apart).  Checked once
// per context on the device it runs on: 256 blocks report the XCD they run on (also true, trivially, when the device is
// a single-XCD partition).
__global__ void __launch_bounds__(64) xcc_probe_kernel(u32 *out)
{
    if (threadIdx.x == 0) {
        u32 xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = xcc & 0xFu;
    }
}
template <int DPL>
static int v3_blocks_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)sgm_vert3_kernel<DPL>, 256, 0) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    return n;
}

// Fills ctx->v3: how this device places consecutive block ids (one XCD id per blockIdx & 7, eight different ones), how
// many CUs one XCD has and how many blocks of each fused kernel fit a CU -- asked of the runtime for THIS build of the
// kernels on THIS device (VGPR counts move with the compiler; SKUs, partitions and CU masks move the CU count).
int rsgm_vert3_probe(vppx_ctx *ctx, u32 *scratch_dev /* >= 256 words */, bool *ok)
{
    u32 h[256];
    xcc_probe_kernel<<<256, 64, 0, ctx->stream>>>(scratch_dev);
    VPPX_CHECK_LAUNCH();
    VPPX_HIP(hipMemcpyAsync(h, scratch_dev, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    VPPX_HIP(hipStreamSynchronize(ctx->stream));
    bool pattern = true;
    for (int i = 8; i < 256; i++)
        if (h[i] != h[i & 7]) pattern = false;
    u32 seen = 0;
    for (int i = 0; i < 8; i++) seen |= 1u << (h[i] & 15u);
    const int nxcd = __builtin_popcount(seen);
    int ncu = 0;
    VPPX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) {
        (void)hipGetLastError();
        khz = 100000; // s_memrealtime of gfx9: 100 MHz
    }
    ctx->v3.nxcd = nxcd;
    ctx->v3.cus_per_xcd = nxcd > 0 ? ncu / nxcd : 0;
    ctx->v3.blocks_per_cu[0] = v3_blocks_per_cu<8>();
    ctx->v3.blocks_per_cu[1] = v3_blocks_per_cu<16>();
    ctx->v3.blocks_per_cu[2] = v3_blocks_per_cu<24>();
    ctx->v3.blocks_per_cu[3] = v3_blocks_per_cu<32>();
    {
        const void *wide_fn[4] = {(const void *)sgm_vert4_kernel<16>, (const void *)sgm_vert4_kernel<32>, (const void *)sgm_vert4_kernel<48>,
                                  (const void *)sgm_vert4_kernel<64>};
        for (int k = 0; k < 4; k++) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wide_fn[k], 256, 0) != hipSuccess) {
                (void)hipGetLastError();
                n = 0;
            }
            ctx->v3.blocks_per_cu16[k] = n;
        }
    }
    ctx->v3.wall_khz = khz;
    // The kernel decodes (group, block) from blockIdx assuming 8 XCDs that take block ids round-robin.  A single-XCD
    // partition (CPX) passes the pattern test trivially, but there the 8 interleaved groups would have to share the one
    // XCD: no fused layout on anything but the 8-XCD placement the kernel was written for.
    ctx->v3.ok = pattern && nxcd == 8 && ctx->v3.cus_per_xcd > 0;
    *ok = ctx->v3.ok;
    return 0;
}

bool rsgm_vert3_supported(int B, int Hp, int Wp, int D, int maxp2)
{
    // shape conditions only (rsgm_vert3_fits has the residency condition): edge values and their packed minimum must
    // stay below 1024 (tag bits), the three-path sum below 256; record offsets are 32-bit byte offsets into one buffer
    // (kept below 1 GiB)
    return (D == 64 || D == 128 || D == 192 || D == 256) && 3 * (24 + maxp2) <= 255 && B > 0 && Wp % 8 == 0 && Hp < 4095 &&
           (size_t)2 * B * (Wp / 8) * (V3_RING * 2 * 8 * 32) * sizeof(u32) < ((size_t)1 << 30);
}

// All blocks of a (frame, pass) group must be resident together, with room to spare for the blocks of the next group
// that arrive early: at most 3/4 of the block slots of one XCD, counted with what the runtime says about this device
// and this build (after rsgm_vert3_probe).  Wider frames take the 8-path layout.
// 16 pixels per wave (sgm_vert4_kernel) or 8 (sgm_vert3_kernel)?  The wide kernel spends 10-15 % fewer instructions per
// pixel, but its groups are half as many blocks at half as many blocks per CU: it wins once the groups of a launch fill an
// XCD's slots at least once.  Measured at 540 x 960 x 192 (15-block groups, 4 per XCD resident), ms per launch 8 / 16 pixels
// per wave: B = 8 1.00 / 1.18, B = 12 1.34 / 1.58, B = 16 1.79 / 1.61, B = 24 2.99 / 2.94, B = 32 3.85 / 3.27.
// VPPX_V3_PPW = 8 / 16 forces one of them.
static int v3_dk(int D) { return D == 64 ? 0 : (D == 128 ? 1 : (D == 192 ? 2 : 3)); }
static bool v3_wide(const vppx_ctx *ctx, int B, int Wp, int D)
{
    if (Wp % 16 != 0 || ctx->v3.ppw == 8) return false;
    if (ctx->v3.ppw == 16) return true;
    const int nbg = (Wp / 16 + 3) / 4;
    const int resident = ctx->v3.cus_per_xcd * ctx->v3.blocks_per_cu16[v3_dk(D)] / (nbg > 0 ? nbg : 1); // whole groups per XCD
    return resident > 0 && (2 * B + 7) / 8 >= resident;
}

bool rsgm_vert3_wide(const vppx_ctx *ctx, int B, int Wp, int D) { return v3_wide(ctx, B, Wp, D); }

// Frames per full round of the 16-pixels-per-wave kernel at this width: (whole groups resident per XCD) x 8 XCDs / 2 passes.
// A batch that is a multiple of it leaves no part-filled last round (0: the fused layout does not apply).
int rsgm_vert3_frames_per_round(const vppx_ctx *ctx, int Wp, int D)
{
    if (!ctx->v3.ok || Wp % 16 != 0 || !(D == 64 || D == 128 || D == 192 || D == 256)) return 0;
    const int nbg = (Wp / 16 + 3) / 4;
    const int resident = ctx->v3.cus_per_xcd * ctx->v3.blocks_per_cu16[v3_dk(D)] / nbg;
    return resident * ctx->v3.nxcd / 2;
}

bool rsgm_vert3_fits(const vppx_ctx *ctx, int B, int Wp, int D)
{
    if (!ctx->v3.ok) return false;
    const int k = v3_dk(D);
    const bool wide = v3_wide(ctx, B, Wp, D);
    const int nwv = wide ? Wp / 16 : Wp / 8;
    const int nbg = (nwv + 3) / 4;
    const int bpc = wide ? ctx->v3.blocks_per_cu16[k] : ctx->v3.blocks_per_cu[k];
    return nbg <= ctx->v3.cus_per_xcd * bpc * 3 / 4;
}
// dwords of edge records per wave: [ring row][direction][lanes per pixel][record slot]
static int v3_rec_dwords(bool wide, int D)
{
    if (!wide) return V3_RING * 2 * 8 * (D == 256 ? V3Rec<16>::XW : 16);
    return V3_RING * 2 * 4 * (D == 64 ? V3Rec<8>::XW : (D == 128 ? V3Rec<16>::XW : (D == 192 ? V3Rec<24>::XW : V3Rec<32>::XW)));
}
size_t rsgm_vert3_xbuf_bytes(int B, int Wp, int D)
{
    // (sized for whichever of the two kernels needs more, + the record of zeros; twice: a batch may run as two launches)
    const size_t narrow = (size_t)2 * B * (Wp / 8) * v3_rec_dwords(false, D);
    const size_t wide = (size_t)2 * B * (Wp / 16) * v3_rec_dwords(true, D);
    return ((narrow > wide ? narrow : wide) + 128) * sizeof(u32);
}
static size_t v3_xbuf_bytes_one(int B, int Wp, int D)
{
    const size_t narrow = (size_t)2 * B * (Wp / 8) * v3_rec_dwords(false, D);
    const size_t wide = (size_t)2 * B * (Wp / 16) * v3_rec_dwords(true, D);
    return ((narrow > wide ? narrow : wide) + 64) * sizeof(u32);
}

// How a batch meets the rounds of the lock-step kernel: *whole_frames = the frames of its whole rounds (0 when the batch is at
// most one round), and whether what is left -- the remainder, or a batch smaller than a round -- fills less than about two
// thirds of a round's block slots.  Such a launch leaves SIMDs idle for its whole duration (78 waves of a 1248-column group on
// an XCD's 128 SIMDs; two 8-pixel groups of a 960-column frame on 512 slots): the W/E launch of the same part, which needs
// nothing of it, then runs NEXT to it (run_aggregation).  Next to a full round it costs more than it hides (NOTEBOOK).
void rsgm_vert3_plan(const vppx_ctx *ctx, int B, int Wp, int D, int *whole_frames, bool *rest_underfilled)
{
    *whole_frames = 0;
    *rest_underfilled = false;
    const int fpr = rsgm_vert3_frames_per_round(ctx, Wp, D); // frames per round of the 16-pixel kernel
    if (fpr <= 0 || !ctx->v3.ok) return;
    int rest = B;
    if (v3_wide(ctx, B, Wp, D) && B > fpr && B % fpr != 0) {
        *whole_frames = B - B % fpr;
        rest = B % fpr;
    } else if (B >= fpr) {
        return; // whole rounds only
    }
    *rest_underfilled = 3 * rest <= 2 * fpr;
}

// One launch of the fused vertical kernel over the frames [f0, f0 + nB) of a batch of B_total (the volume's pass stride);
// first: the batch's first launch (new serial, records cleared).
int rsgm_launch_vert3_range(vppx_ctx *ctx, hipStream_t stream, int B_total, int f0, int nB, int Hp, int Wp, int D, const u8 *gray,
                            const u32 *cl, const u32 *cr, const u16 *p2lut, int p1, u8 *sv, u32 *xbuf, unsigned *err, unsigned *err_dev,
                            bool next_to_we)
{
    if (f0 == 0) {
        ++ctx->v3.serial; // one serial per aggregation, whatever the number of launches (void_if_lost_kernel)
        if (ctx->v3.serial == 0) ++ctx->v3.serial;
        // no record of an earlier launch may match: cleared here, unless a pipelined call's front_end has already queued the
        // clear on this stream behind the previous launch (vppx_api.hip; never inside a graph capture: a replayed graph must
        // carry its own clear)
        const size_t xbytes = rsgm_vert3_xbuf_bytes(B_total, Wp, D);
        if (!(ctx->xbuf_cleared && !ctx->capturing && ctx->xbuf_last == (void *)xbuf && ctx->xbuf_last_bytes >= xbytes && stream == ctx->stream))
            VPPX_HIP(hipMemsetAsync(xbuf, 0, xbytes, stream));
        ctx->xbuf_last = (void *)xbuf;
        ctx->xbuf_last_bytes = xbytes;
        ctx->xbuf_cleared = false;
    } else {
        xbuf += v3_xbuf_bytes_one(f0, Wp, D) / sizeof(u32); // a second launch's records lie behind the first's
    }
    const size_t fpix = (size_t)f0 * Hp * Wp;
    Vert3Args a;
    a.gray = gray + fpix; a.cl = cl + fpix; a.cr = cr + fpix; a.p2lut = p2lut; a.sv = sv + fpix * D; a.xbuf = xbuf; a.err = err; a.err_dev = err_dev;
    a.B = nB; a.Hp = Hp; a.Wp = Wp;
    a.p1 = p1 > 231 ? 231 : p1; // exact for P1 >= P2max (see rsgm_launch_paths); keeps the small-value update in range
    bool wide = v3_wide(ctx, nB, Wp, D);
    if (!wide && next_to_we && Wp % 16 == 0 && ctx->v3.ppw == 0) {
        // With W/E running next to it the launch's idle SIMDs are not idle: what counts is its own instruction count, and the
        // 16-pixel kernel spends 13 % fewer -- from half a round on (below that its groups are alone on their XCDs and run at a
        // lone wave's pace whichever kernel they are).  540x960x192, ms per call 8 / 16 pixels per wave: 8 frames 2.25 / 2.14,
        // 9 frames 2.72 / 2.67, 10 frames 3.06 / 2.78, 4 frames 1.35 / 1.68; 375x1242x192 (3 groups per XCD and round), 8 frames:
        // 2.32 / 2.35 -- hence only where a round holds at least 4 groups per XCD.
        const int nbg16 = (Wp / 16 + 3) / 4;
        const int resident = ctx->v3.cus_per_xcd * ctx->v3.blocks_per_cu16[v3_dk(D)] / (nbg16 > 0 ? nbg16 : 1);
        wide = resident >= 4 && 2 * nB >= ctx->v3.nxcd * ((resident + 1) / 2);
    }
    {
        // residency of the kernel actually chosen for THIS launch (rsgm_vert3_fits priced the whole batch's choice): a group
        // that cannot be co-resident with room to spare takes the other kernel when that one fits
        const int k = v3_dk(D);
        auto fits = [&](bool w) {
            const int nbg = ((w ? Wp / 16 : Wp / 8) + 3) / 4;
            const int bpc = w ? ctx->v3.blocks_per_cu16[k] : ctx->v3.blocks_per_cu[k];
            return (!w || Wp % 16 == 0) && nbg <= ctx->v3.cus_per_xcd * bpc * 3 / 4;
        };
        if (!fits(wide) && fits(!wide)) wide = !wide;
    }
    if (f0 == 0) ctx->v3.last_ppw = wide ? 16 : 8;
    a.nwv = wide ? Wp / 16 : Wp / 8;
    a.nbg = (a.nwv + 3) / 4;
    a.vol_elems = (size_t)B_total * Hp * Wp * D;
    a.timeout_ticks = (long long)ctx->v3.timeout_ms * (long long)(ctx->v3.wall_khz > 0 ? ctx->v3.wall_khz : 100000);
    a.spin_limit = ctx->v3.spin_limit;
    a.serial = ctx->v3.serial;
    const dim3 grid((unsigned)((2 * nB + 7) / 8 * 8 * a.nbg)); // whole rounds of 8 groups; groups >= 2 nB exit at once
    if (wide && D == 64) sgm_vert4_kernel<16><<<grid, 256, 0, stream>>>(a);
    else if (wide && D == 128) sgm_vert4_kernel<32><<<grid, 256, 0, stream>>>(a);
    else if (wide && D == 192) sgm_vert4_kernel<48><<<grid, 256, 0, stream>>>(a);
    else if (wide) sgm_vert4_kernel<64><<<grid, 256, 0, stream>>>(a);
    else if (D == 64) sgm_vert3_kernel<8><<<grid, 256, 0, stream>>>(a);
    else if (D == 128) sgm_vert3_kernel<16><<<grid, 256, 0, stream>>>(a);
    else if (D == 192) sgm_vert3_kernel<24><<<grid, 256, 0, stream>>>(a);
    else sgm_vert3_kernel<32><<<grid, 256, 0, stream>>>(a);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int rsgm_launch_vert3(vppx_ctx *ctx, hipStream_t stream, int B, int Hp, int Wp, int D, const u8 *gray, const u32 *cl,
                      const u32 *cr, const u16 *p2lut, int p1, u8 *sv, u32 *xbuf, unsigned *err, unsigned *err_dev)
{
    return rsgm_launch_vert3_range(ctx, stream, B, 0, B, Hp, Wp, D, gray, cl, cr, p2lut, p1, sv, xbuf, err, err_dev, false);
}

// ---------------------------------------------------------------------------------------
// S = sum of the 8 path volumes, fused with the left WTA + uniqueness (rsgm.py:141) and the
// equiangular sub-pixel refinement (rsgm.py:142).  One 16-lane row per pixel.
// ---------------------------------------------------------------------------------------
template <int DPL>
__device__ __forceinline__ float wta_rows(const u32 (&S)[DPL / 2], int dbase, int n /*valid d: d < n*/, int D,
                                          u32 factor_uniq, bool do_subpixel, bool x_interior)
{
    constexpr int NP = DPL / 2;
    // key = cost*256 + d : first minimum wins on ties
    u32 key = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int d0 = dbase + 2 * i, d1 = d0 + 1;
        const u32 k0 = d0 < n ? (((S[i] & 0xFFFFu) << 8) | (u32)d0) : 0xFFFFFFFFu;
        const u32 k1 = d1 < n ? (((S[i] >> 16) << 8) | (u32)d1) : 0xFFFFFFFFu;
        key = min(key, min(k0, k1));
    }
    key = row_min_u32(key);
    const int best = (int)(key & 0xFFu);
    const u32 minc = key >> 8;
    u32 sec = 65535u, cm1 = 0, cp1 = 0, nbhit = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int d0 = dbase + 2 * i, d1 = d0 + 1;
        const u32 s0 = S[i] & 0xFFFFu, s1 = S[i] >> 16;
        if (d0 < n && d0 != best) sec = min(sec, s0);
        if (d1 < n && d1 != best) sec = min(sec, s1);
        if (d0 == best - 1) cm1 = s0;
        if (d1 == best - 1) cm1 = s1;
        if (d0 == best + 1) cp1 = s0;
        if (d1 == best + 1) cp1 = s1;
    }
    sec = row_min_u32(sec);
    cm1 = row_or_u32(cm1);
    cp1 = row_or_u32(cp1);
    (void)nbhit;
    bool ok = (1024u * minc <= sec * factor_uniq);
    if (!ok) {
        if (best > 0 && cm1 == sec) ok = true;
        if (best + 1 < n && cp1 == sec) ok = true;
    }
    float disp = ok ? (float)best : INVALID_DISP;
    if (do_subpixel && x_interior && disp > 0.0f && best >= 1 && best <= D - 2) {
        const int c0 = (int)cm1, c1 = (int)minc, c2 = (int)cp1;
        const int den = (c2 < c0) ? c0 - c1 : c2 - c1;
        if (den != 0) disp = (float)best + __fdiv_rn((float)(c0 - c2), __fmul_rn(2.0f, (float)den));
    }
    return disp;
}

// The same decision for byte volumes (S <= 8 * 255 < 4096), written for gfx950's issue rates: 16-bit
// keys (S << 4 | index in the lane) keep a top-2 tournament in packed ops (3 per register instead of a compare/select
// chain per value), the 16 lanes of the pixel merge their (first, second) pairs in four DPP steps, and the two
// neighbour values S[best -+ 1] are fetched through `nb` (the caller has the pixel's column in LDS anyway).
// The second-smallest KEY carries the second-smallest VALUE (ties included), which is all the rule needs.
template <int DPL, typename NB>
__device__ __forceinline__ float wta_top2(const u32 (&S)[DPL / 2], int dbase, int n /*valid d: d < n*/, int D, u32 factor_uniq,
                                          bool do_subpixel, bool x_interior, NB nb)
{
    constexpr int NP = DPL / 2;
    static_assert(DPL <= 16, "4-bit local index");
    u32 K[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) K[i] = (S[i] << 4) | (u32)((2 * i) | ((2 * i + 1) << 16));
    if (__builtin_amdgcn_ballot_w64(n < D) != 0) { // some pixel of this wave has a clipped search range
        const int c = n - dbase;                   // local indices >= c are not candidates
#pragma unroll
        for (int i = 0; i < NP; i++) K[i] |= (2 * i >= c ? 0x0000FFFFu : 0u) | (2 * i + 1 >= c ? 0xFFFF0000u : 0u);
    }
    u32 m1 = K[0], m2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 1; i < NP; i++) {
        const u16x2 a = __builtin_bit_cast(u16x2, m1), k = __builtin_bit_cast(u16x2, K[i]);
        const u32 t = __builtin_bit_cast(u32, __builtin_elementwise_max(a, k));
        m1 = pk_min(m1, K[i]);
        m2 = pk_min(m2, t);
    }
    // the two halves of (m1, m2) are two sorted pairs: merge them
    const u32 a1 = m1 & 0xFFFFu, b1 = m1 >> 16, a2 = m2 & 0xFFFFu, b2 = m2 >> 16;
    const u32 k1 = min(a1, b1), k2 = min(max(a1, b1), min(a2, b2));
    // 32-bit keys comparable across the pixel's lanes: value << 8 | disparity (first minimum wins on ties)
    u32 g1 = ((k1 >> 4) << 8) | (u32)(dbase + (int)(k1 & 15u));
    u32 g2 = ((k2 >> 4) << 8) | (u32)(dbase + (int)(k2 & 15u));
#define VPPX_TOP2_STEP(CTRL)                                    \
    {                                                           \
        const u32 p1 = dpp_ror<CTRL>(g1), p2 = dpp_ror<CTRL>(g2); \
        const u32 t = max(g1, p1);                              \
        g1 = min(g1, p1);                                       \
        g2 = min(min(g2, p2), t);                               \
    }
    VPPX_TOP2_STEP(0x128) VPPX_TOP2_STEP(0x124) VPPX_TOP2_STEP(0x122) VPPX_TOP2_STEP(0x121)
#undef VPPX_TOP2_STEP
    const int best = (int)(g1 & 0xFFu);
    const u32 minc = g1 >> 8;
    u32 sec = g2 >> 8;
    sec = sec >= 4095u ? 65535u : sec; // masked keys carry 4095: no second candidate
    const u32 cm1 = best >= 1 ? nb(best - 1) : 0u, cp1 = best + 1 < D ? nb(best + 1) : 0u;
    bool ok = (1024u * minc <= sec * factor_uniq);
    if (!ok) {
        if (best > 0 && cm1 == sec) ok = true;
        if (best + 1 < n && cp1 == sec) ok = true;
    }
    float disp = ok ? (float)best : INVALID_DISP;
    if (do_subpixel && x_interior && disp > 0.0f && best >= 1 && best <= D - 2) {
        const int c0 = (int)cm1, c1 = (int)minc, c2 = (int)cp1;
        const int den = (c2 < c0) ? c0 - c1 : c2 - c1;
        if (den != 0) disp = (float)best + __fdiv_rn((float)(c0 - c2), __fmul_rn(2.0f, (float)den));
    }
    return disp;
}

// The same decision in two parts (round 3).  Every lane of a pixel ends the reduction with the same (best, second) keys, so
// whatever follows -- the uniqueness test, the sub-pixel step with its float division, the store -- is the same work done
// GL times over, and on gfx950 this path is bound by instruction issue.  top2_reduce stops where the lanes agree (and
// fetches S[best -+ 1], which must be read while the tile is in the ring); the caller lets ONE lane per round keep the
// result of its pixel and runs top2_final once every GL rounds, for 64 different pixels per wave instruction.
// GL lanes per pixel (4, 8 or 16), DPL disparities per lane; keys are 16 bits: S << IB | index in the lane.
struct Top2 {
    u32 g1, g2, nbv; // value << 8 | disparity of the minimum and of the runner-up; S[best-1] | S[best+1] << 16
};
template <int DPL, int GL, typename NB>
__device__ __forceinline__ Top2 top2_reduce(const u32 (&S)[DPL / 2], int dbase, int n /*valid d: d < n*/, int D, NB nb)
{
    constexpr int NP = DPL / 2;
    constexpr int IB = DPL <= 16 ? 4 : 5; // bits of the index in the lane
    static_assert(DPL <= 32, "5-bit local index");
    u32 K[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) K[i] = (S[i] << IB) | (u32)((2 * i) | ((2 * i + 1) << 16));
    if (__builtin_amdgcn_ballot_w64(n < D) != 0) { // some pixel of this wave has a clipped search range
        const int c = n - dbase;                   // local indices >= c are not candidates
#pragma unroll
        for (int i = 0; i < NP; i++) K[i] |= (2 * i >= c ? 0x0000FFFFu : 0u) | (2 * i + 1 >= c ? 0xFFFF0000u : 0u);
    }
    u32 m1 = K[0], m2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 1; i < NP; i++) {
        const u16x2 a = __builtin_bit_cast(u16x2, m1), k = __builtin_bit_cast(u16x2, K[i]);
        const u32 t = __builtin_bit_cast(u32, __builtin_elementwise_max(a, k));
        m1 = pk_min(m1, K[i]);
        m2 = pk_min(m2, t);
    }
    const u32 a1 = m1 & 0xFFFFu, b1 = m1 >> 16, a2 = m2 & 0xFFFFu, b2 = m2 >> 16;
    const u32 k1 = min(a1, b1), k2 = min(max(a1, b1), min(a2, b2));
    // 32-bit keys comparable across the pixel's lanes: value << 8 | disparity (first minimum wins on ties); a masked key
    // keeps the largest value 16 - IB bits hold
    u32 g1 = ((k1 >> IB) << 8) | (u32)(dbase + (int)(k1 & ((1u << IB) - 1u)));
    u32 g2 = ((k2 >> IB) << 8) | (u32)(dbase + (int)(k2 & ((1u << IB) - 1u)));
#define VPPX_TOP2_STEP(CTRL)                                    \
    {                                                           \
        const u32 p1 = dpp_ror<CTRL>(g1), p2 = dpp_ror<CTRL>(g2); \
        const u32 t = max(g1, p1);                              \
        g1 = min(g1, p1);                                       \
        g2 = min(min(g2, p2), t);                               \
    }
    if constexpr (GL == 16) {
        VPPX_TOP2_STEP(0x128) VPPX_TOP2_STEP(0x124) VPPX_TOP2_STEP(0x122) VPPX_TOP2_STEP(0x121)
    } else { // partners l ^ 1, l ^ 2 (quad permutations), then 7 - l (row_half_mirror): disjoint sets at every step
        VPPX_TOP2_STEP(0xB1) VPPX_TOP2_STEP(0x4E)
        if constexpr (GL == 8) VPPX_TOP2_STEP(0x141)
    }
#undef VPPX_TOP2_STEP
    const int best = (int)(g1 & 0xFFu);
    const u32 cm1 = best >= 1 ? nb(best - 1) : 0u, cp1 = best + 1 < D ? nb(best + 1) : 0u;
    Top2 r;
    r.g1 = g1;
    r.g2 = g2;
    r.nbv = cm1 | (cp1 << 16);
    return r;
}
template <int DPL>
__device__ __forceinline__ float top2_final(u32 g1, u32 g2, u32 nbv, int n, int D, u32 factor_uniq, bool do_subpixel, bool x_interior)
{
    constexpr int IB = DPL <= 16 ? 4 : 5;
    const int best = (int)(g1 & 0xFFu);
    const u32 minc = g1 >> 8;
    u32 sec = g2 >> 8;
    sec = sec >= (0xFFFFu >> IB) ? 65535u : sec; // masked keys: no second candidate
    const u32 cm1 = nbv & 0xFFFFu, cp1 = nbv >> 16;
    bool ok = (1024u * minc <= sec * factor_uniq);
    if (!ok) {
        if (best > 0 && cm1 == sec) ok = true;
        if (best + 1 < n && cp1 == sec) ok = true;
    }
    float disp = ok ? (float)best : INVALID_DISP;
    if (do_subpixel && x_interior && disp > 0.0f && best >= 1 && best <= D - 2) {
        const int c0 = (int)cm1, c1 = (int)minc, c2 = (int)cp1;
        const int den = (c2 < c0) ? c0 - c1 : c2 - c1;
        if (den != 0) disp = (float)best + __fdiv_rn((float)(c0 - c2), __fmul_rn(2.0f, (float)den));
    }
    return disp;
}

// A block owns 64 consecutive pixels of one row (4 rounds of 16 pixels).  Besides the left
// disparity it writes the aggregated volume in one or both of two layouts:
//   S  [y][x][d]  ("xyd", the reference's dsiAgg layout; stage API only)
//   ST [y][d][x]  (d-major per row): right-view WTA reads ST[y][d][xr+d], which is coalesced
//                 across consecutive xr; the 64 x 192 tile is transposed through LDS.
template <int DPL, bool EXACT, typename IT>
__global__ void __launch_bounds__(256) sum_wta_kernel(const IT *__restrict__ paths, size_t vol_elems, u16 *__restrict__ S,
                                                      u16 *__restrict__ ST, float *__restrict__ disp, int Hp, int Wp,
                                                      int D, u32 factor_uniq, int do_subpixel)
// (always sums the 8 per-path volumes: general / stage-API form)
{
    constexpr int NP = DPL / 2;
    extern __shared__ __attribute__((aligned(16))) u16 tile[]; // [D][66]
    const int x0 = blockIdx.x * 64, y = blockIdx.y, f = blockIdx.z;
    const size_t rowpix = ((size_t)f * Hp + y) * Wp;
    const int l16 = threadIdx.x & 15;
    const int dbase = DPL * l16;
    for (int it = 0; it < 4; it++) {
        const int p = it * 16 + (threadIdx.x >> 4);
        const int x = x0 + p;
        if (x >= Wp) continue; // uniform per 16-lane group
        const size_t pix = rowpix + x;
        u32 acc[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) acc[i] = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const IT *vp = paths + (size_t)k * vol_elems + pix * D + dbase;
            if (sizeof(IT) == 2) {
                u32 w[NP];
                if (EXACT) {
                    load_words<NP>((const u32 *)vp, w);
                } else {
#pragma unroll
                    for (int i = 0; i < NP; i++) w[i] = (dbase + 2 * i < D) ? ((const u32 *)vp)[i] : 0u;
                }
#pragma unroll
                for (int i = 0; i < NP; i++) acc[i] = pk_adds(acc[i], w[i]);
            } else {
                u32 w[NP / 2];
                load_words<NP / 2>((const u32 *)vp, w);
#pragma unroll
                for (int i = 0; i < NP / 2; i++) {
                    acc[2 * i] = pk_adds(acc[2 * i], __builtin_amdgcn_perm(w[i], w[i], 0x0c010c00u));
                    acc[2 * i + 1] = pk_adds(acc[2 * i + 1], __builtin_amdgcn_perm(w[i], w[i], 0x0c030c02u));
                }
            }
        }
        if (S) {
            u32 *sp = (u32 *)(S + pix * D + dbase);
            if (EXACT) {
                store_words<NP>(sp, acc);
            } else {
#pragma unroll
                for (int i = 0; i < NP; i++)
                    if (dbase + 2 * i < D) sp[i] = acc[i];
            }
        }
        if (ST) {
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const int d0 = dbase + 2 * i;
                if (EXACT || d0 < D) {
                    tile[d0 * 66 + p] = (u16)(acc[i] & 0xFFFFu);
                    tile[(d0 + 1) * 66 + p] = (u16)(acc[i] >> 16);
                }
            }
        }
        if (disp) {
            const int n = (x < D - 1 ? x : D - 1) + 1;
            const float dv = wta_rows<DPL>(acc, dbase, n, D, factor_uniq, do_subpixel != 0, x >= 1 && x <= Wp - 2);
            if (l16 == 0) disp[pix] = dv;
        }
    }
    if (ST) {
        __syncthreads();
        const u32 *t32 = (const u32 *)tile;
        u32 *o32 = (u32 *)(ST + ((size_t)f * Hp + y) * D * Wp);
        for (int idx = threadIdx.x; idx < D * 32; idx += 256) {
            const int d = idx >> 5, j = idx & 31;
            if (x0 + 2 * j < Wp) o32[((size_t)d * Wp + x0) / 2 + j] = t32[d * 33 + j];
        }
    }
}

template <int DPL>
static int launch_sum_wta_t(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const void *paths, int elem_bytes, u16 *S,
                            u16 *ST, float *disp, u32 fu, int sub)
{
    dim3 grid((Wp + 63) / 64, Hp, B);
    const size_t vol = (size_t)B * Hp * Wp * D;
    const bool exact = (D == 16 * DPL);
    const size_t lds = ST ? (size_t)D * 66 * sizeof(u16) : 0;
    if (elem_bytes == 1) {
        sum_wta_kernel<DPL, true, u8><<<grid, 256, lds, ctx->stream>>>((const u8 *)paths, vol, S, ST, disp, Hp, Wp, D, fu, sub);
    } else if (exact) {
        sum_wta_kernel<DPL, true, u16><<<grid, 256, lds, ctx->stream>>>((const u16 *)paths, vol, S, ST, disp, Hp, Wp, D, fu, sub);
    } else {
        sum_wta_kernel<DPL, false, u16><<<grid, 256, lds, ctx->stream>>>((const u16 *)paths, vol, S, ST, disp, Hp, Wp, D, fu, sub);
    }
    VPPX_CHECK_LAUNCH();
    return 0;
}

int rsgm_launch_sum_wta(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const void *paths, int elem_bytes, u16 *S,
                        u16 *ST, float *disp_l, u32 factor_uniq, int do_subpixel)
{
    const int dpl = dpl_for(D);
    if (D != 16 * dpl) elem_bytes = 2;
    switch (dpl) {
    case 4: return launch_sum_wta_t<4>(ctx, B, Hp, Wp, D, paths, elem_bytes, S, ST, disp_l, factor_uniq, do_subpixel);
    case 8: return launch_sum_wta_t<8>(ctx, B, Hp, Wp, D, paths, elem_bytes, S, ST, disp_l, factor_uniq, do_subpixel);
    case 12: return launch_sum_wta_t<12>(ctx, B, Hp, Wp, D, paths, elem_bytes, S, ST, disp_l, factor_uniq, do_subpixel);
    default: return launch_sum_wta_t<16>(ctx, B, Hp, Wp, D, paths, elem_bytes, S, ST, disp_l, factor_uniq, do_subpixel);
    }
}

// ---------------------------------------------------------------------------------------
// Fused sum + left WTA + right WTA (fast path, D = 64/128/192): one block walks one image row
// in rounds of 64 pixels and keeps the last D/64+1 aggregated 64 x D tiles in an LDS ring, so
// the right view (d_R(x) = argmin_d S[x+d][d], needs x .. x+D-1) is read from LDS instead of a
// second pass over HBM: the path volumes are read once and S is never written.
// ---------------------------------------------------------------------------------------
struct VolPtrs {
    const void *v[8];
};

// T = pixels per round (= 16-lane groups per block, block = 16*T threads).  SPARE: one extra ring
// slot lets a single barrier per round suffice; without it (D = 256: LDS is tight) a second barrier
// separates the right-view reads of round k from the tile write of round k+1.
// A block walks `rows_per_block` consecutive rows as ONE stream of rounds (tile j of the stream is
// tile j % ntiles of row j / ntiles): the operand prefetch and the right view's NR-1 round lag run
// across row boundaries, so the fill/drain rounds are paid once per block instead of once per row.
// SW4: byte volumes whose values are <= 63 (4 * max <= 255): four volumes are added as packed bytes with plain
// 32-bit adds (no carry can cross a byte) before the sums are widened to u16 pairs
// GL = lanes per pixel (16; 8 for the byte volumes of D = 128 / 192, round 3: per-pixel overhead -- cross-lane merges, ring
// addressing, cursors -- is paid per lane, and the kernel is bound by instruction issue).  FAST variants finalise a pixel
// once, not GL times (top2_reduce / top2_final): a lane keeps the keys of ONE pixel per GL rounds.
// FULLW: the padded width is a multiple of the round (no partial tile: the per-pixel "inside the row" tests and the divergent
// regions they open disappear).
template <int DPL, typename IT, int NV, int T, bool SPARE, bool SW4, bool FAST = false, int GL = 16, bool FULLW = false>
__global__ void __launch_bounds__(GL * T) sum_wta_lr_kernel(VolPtrs vols, float *__restrict__ disp_l,
                                                          float *__restrict__ disp_r, int Hp, int Wp, u32 factor_uniq,
                                                          int do_subpixel, int rows_per_block)
{
    constexpr int NP = DPL / 2;
    constexpr int D = GL * DPL;
    static_assert(GL == 16 || FAST, "fewer lanes per pixel: byte-volume variants only");
    constexpr int NR = (T + D - 2) / T + 1; // tiles a right pixel can touch (x .. x+D-1)
    constexpr int NT = NR + (SPARE ? 1 : 0);
    constexpr int TW = T + 2;               // tile row pitch in u16 (2 pad)
    constexpr int TS = D * TW;              // u16 per tile
    constexpr int NWD = (sizeof(IT) == 2) ? NP : NP / 2; // dwords per lane per volume
    extern __shared__ __attribute__((aligned(16))) u16 ring_alloc[]; // T guard elements, then [NT][D][TW]
    u16 *const ring = ring_alloc + T; // (the FAST gather forms base addresses up to T elements below a tile row's start)
    const int f = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int nrows = (Hp - r0 < rows_per_block) ? Hp - r0 : rows_per_block;
    const size_t framepix = (size_t)f * Hp * Wp;
    const int g = threadIdx.x / GL, l16 = threadIdx.x % GL; // pixel of the round, lane of the pixel
    const int dbase = DPL * l16;
    // deferred finalisation (FAST): the keys of the pixel this lane keeps, per view {g1, g2, S[best-+1], n | interior << 9 |
    // valid << 10, pixel index in the frame}
    u32 kl[5] = {0, 0, 0, 0, 0}, kr[5] = {0, 0, 0, 0, 0};
    auto flush = [&]() {
        if (kl[3]) {
            const float dv = top2_final<DPL>(kl[0], kl[1], kl[2], (int)(kl[3] & 511u), D, factor_uniq, do_subpixel != 0, (kl[3] >> 9) & 1u);
            disp_l[framepix + kl[4]] = dv;
        }
        if (kr[3]) {
            const float dv = top2_final<DPL>(kr[0], kr[1], kr[2], (int)(kr[3] & 511u), D, factor_uniq, false, false);
            disp_r[framepix + kr[4]] = dv;
        }
        kl[3] = 0;
        kr[3] = 0;
    };
    const int ntiles = (Wp + T - 1) / T;
    const int total = nrows * ntiles;
    // stream cursors (uniform): fetch runs two rounds ahead of the left view, the right view NR-1 behind
    int f_row = r0, f_k = 0;
    int l_row = r0, l_k = 0, l_slot = 0;
    int r_row = r0, r_k = 0, r_slot = 0;
    // operands are fetched two rounds ahead into a ping-pong register set: the loads of round j+2
    // are issued before round j is summed, so two rounds of HBM latency are covered
    // four volumes (fused vertical layout) leave room for a third operand set: three rounds of HBM latency covered
    constexpr int PF = (NV == 4 && sizeof(IT) == 1) ? 3 : 2;
    u32 wa[NV][NWD], wb[NV][NWD], wc[PF == 3 ? NV : 1][PF == 3 ? NWD : 1];
    // (always issued, also past the end of the block's stream, where it re-reads the last row: a fetch under a condition makes
    // the compiler shuffle whole operand sets between registers at the join)
    auto fetch = [&](u32 (&w)[NV][NWD]) {
        int x = T * f_k + g;
        if (!FULLW) x = x < Wp ? x : Wp - 1;
        const int frow = f_row < r0 + nrows ? f_row : r0 + nrows - 1;
        const size_t rowoff = (framepix + (size_t)frow * Wp) * D; // uniform: scalar base + 32-bit lane offset
        // FAST: cells with d > x + 1 have no reader (left WTA: d <= x, its sub-pixel step d <= x + 1; the right view reads
        // S[x][d] for the right pixel x - d >= 0): such a lane re-reads the pixel's first chunk (same cache lines as its
        // neighbours' loads: no extra HBM sectors, no branch) -- 9 % fewer bytes fetched at 960 x 192
        const u32 laneoff = (u32)x * D + ((!FAST || dbase <= x + 1) ? dbase : 0);
#pragma unroll
        for (int v = 0; v < NV; v++) load_words_nt<NWD>((const u32 *)((const IT *)vols.v[v] + rowoff + laneoff), w[v]);
        if (++f_k == ntiles) {
            f_k = 0;
            f_row++;
        }
    };
    auto round = [&](int j, u32 (&w)[NV][NWD]) {
        if (j < total) {
            const int x = T * l_k + g;
            u32 acc[NP];
#pragma unroll
            for (int i = 0; i < NP; i++) acc[i] = 0;
            if constexpr (SW4 && sizeof(IT) == 1 && NV == 4) {
                // fused-path layout: two single-path volumes (W, E; their byte sum cannot carry) and two volumes that
                // hold three summed paths each (<= 255): widen the three operands, plain 32-bit adds on the u16 pairs
#pragma unroll
                for (int i = 0; i < NP / 2; i++) {
                    const u32 ab = w[0][i] + w[1][i];
                    acc[2 * i] = __builtin_amdgcn_perm(ab, ab, 0x0c010c00u) + __builtin_amdgcn_perm(w[2][i], w[2][i], 0x0c010c00u) +
                                 __builtin_amdgcn_perm(w[3][i], w[3][i], 0x0c010c00u);
                    acc[2 * i + 1] = __builtin_amdgcn_perm(ab, ab, 0x0c030c02u) + __builtin_amdgcn_perm(w[2][i], w[2][i], 0x0c030c02u) +
                                     __builtin_amdgcn_perm(w[3][i], w[3][i], 0x0c030c02u);
                }
            } else if constexpr (SW4 && sizeof(IT) == 1 && NV == 8) {
#pragma unroll
                for (int i = 0; i < NP / 2; i++) {
                    const u32 ab = (w[0][i] + w[1][i]) + (w[2][i] + w[3][i]);
                    acc[2 * i] = __builtin_amdgcn_perm(ab, ab, 0x0c010c00u);
                    acc[2 * i + 1] = __builtin_amdgcn_perm(ab, ab, 0x0c030c02u);
                    // sums stay below 2^16: plain 32-bit adds (full rate) on the pairs
                    const u32 cd = (w[4][i] + w[5][i]) + (w[6][i] + w[7][i]);
                    acc[2 * i] += __builtin_amdgcn_perm(cd, cd, 0x0c010c00u);
                    acc[2 * i + 1] += __builtin_amdgcn_perm(cd, cd, 0x0c030c02u);
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    if (sizeof(IT) == 2) {
#pragma unroll
                        for (int i = 0; i < NP; i++) acc[i] = pk_adds(acc[i], w[v][i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < NP / 2; i++) {
                            acc[2 * i] = pk_adds(acc[2 * i], __builtin_amdgcn_perm(w[v][i], w[v][i], 0x0c010c00u));
                            acc[2 * i + 1] = pk_adds(acc[2 * i + 1], __builtin_amdgcn_perm(w[v][i], w[v][i], 0x0c030c02u));
                        }
                    }
                }
            }
            fetch(w); // this register set is free again
            if (FULLW || x < Wp) { // uniform per 16-lane group
                u16 *t = ring + l_slot * TS + g;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    t[(dbase + 2 * i) * TW] = (u16)(acc[i] & 0xFFFFu);
                    t[(dbase + 2 * i + 1) * TW] = (u16)(acc[i] >> 16);
                }
                const int n = (x < D - 1 ? x : D - 1) + 1;
                if constexpr (FAST) { // the pixel's column was just written by this wave (LDS keeps a wave's accesses in order)
                    const Top2 r = top2_reduce<DPL, GL>(acc, dbase, n, D, [&](int d) -> u32 { return t[d * TW]; });
                    if (l16 == (j & (GL - 1))) { // this lane keeps this round's pixel
                        asm volatile("; keep (left view)");
                        kl[0] = r.g1; kl[1] = r.g2; kl[2] = r.nbv;
                        kl[3] = (u32)n | ((x >= 1 && x <= Wp - 2) ? 512u : 0u) | 1024u;
                        kl[4] = (u32)(l_row * Wp + x);
                    }
                } else {
                    const float dv = wta_rows<DPL>(acc, dbase, n, D, factor_uniq, do_subpixel != 0, x >= 1 && x <= Wp - 2);
                    if (l16 == 0) disp_l[framepix + (size_t)l_row * Wp + x] = dv;
                }
            }
            l_slot = (l_slot + 1 == NT) ? 0 : l_slot + 1;
            if (++l_k == ntiles) {
                l_k = 0;
                l_row++;
            }
        }
        __syncthreads();
        if (j >= NR - 1) {
            const int xr = T * r_k + g;
            if (FULLW || xr < Wp) {
                const int n = (Wp - 1 - xr < D - 1 ? Wp - 1 - xr : D - 1) + 1;
                u32 sr[NP];
                if constexpr (FAST) {
                    // S[xr+d][d] for the lane's DPL consecutive d: the column g + d crosses a tile border at most once, so two
                    // base addresses (this tile / the next, column wrapped) + a compile-time stride of one tile row + 1 serve
                    // all elements: one select per element, the rest sits in the instruction's offset field
                    const u32 a0 = (u32)(g + dbase);
                    u32 sA = r_slot + a0 / T, sB = sA + 1;
                    sA = sA >= NT ? sA - NT : sA;
                    sB = sB >= NT ? sB - NT : sB;
                    sB = sB >= NT ? sB - NT : sB;
                    const u32 lane_base = (u32)dbase * TW + (a0 % T);
                    // (24-bit multiplies: v_mul_lo_u32 is quarter rate)
                    const int RA = (int)(__umul24(sA, (u32)TS) + lane_base), RB = (int)(__umul24(sB, (u32)TS) + lane_base) - T; // RB >= -T: guard elements
                    const int e_cross = T - (int)(a0 % T); // first element that lies in the next tile
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        const u32 lo = ring[(2 * i < e_cross ? RA : RB) + (2 * i) * (TW + 1)];
                        const u32 hi = ring[(2 * i + 1 < e_cross ? RA : RB) + (2 * i + 1) * (TW + 1)];
                        sr[i] = lo | (hi << 16);
                    }
                } else
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    // S[xr+d][d]: stream tile (g+d)/T past the pixel's own (disparities beyond the row end
                    // land in the next row's tiles and are masked by n)
                    const u32 a0 = (u32)(g + dbase + 2 * i), a1 = a0 + 1;
                    u32 s0 = r_slot + a0 / T, s1 = r_slot + a1 / T;
                    s0 = s0 >= NT ? s0 - NT : s0;
                    s1 = s1 >= NT ? s1 - NT : s1;
                    const u32 lo = ring[s0 * TS + (dbase + 2 * i) * TW + (a0 % T)];
                    const u32 hi = ring[s1 * TS + (dbase + 2 * i + 1) * TW + (a1 % T)];
                    sr[i] = lo | (hi << 16);
                }
                if constexpr (FAST) {
                    const Top2 r = top2_reduce<DPL, GL>(sr, dbase, n, D, [&](int d) -> u32 {
                        const u32 a = (u32)(g + d);
                        u32 sl = r_slot + a / T;
                        sl = sl >= NT ? sl - NT : sl;
                        return ring[__umul24(sl, (u32)TS) + __umul24((u32)d, (u32)TW) + (a % T)];
                    });
                    if (l16 == (j & (GL - 1))) {
                        asm volatile("; keep (right view)");
                        kr[0] = r.g1; kr[1] = r.g2; kr[2] = r.nbv;
                        kr[3] = (u32)n | 1024u;
                        kr[4] = (u32)(r_row * Wp + xr);
                    }
                } else {
                    const float dv = wta_rows<DPL>(sr, dbase, n, D, factor_uniq, false, false);
                    if (l16 == 0) disp_r[framepix + (size_t)r_row * Wp + xr] = dv;
                }
            }
            r_slot = (r_slot + 1 == NT) ? 0 : r_slot + 1;
            if (++r_k == ntiles) {
                r_k = 0;
                r_row++;
            }
        }
        if constexpr (FAST) {
            if ((j & (GL - 1)) == GL - 1) flush(); // every lane of the wave holds a pixel of its own now
        }
        if (!SPARE) __syncthreads();
    };
    fetch(wa);
    fetch(wb);
    if constexpr (PF == 3) {
        fetch(wc);
        for (int j = 0; j < total + NR - 1; j += 3) {
            round(j, wa);
            if (j + 1 < total + NR - 1) round(j + 1, wb);
            if (j + 2 < total + NR - 1) round(j + 2, wc);
        }
    } else {
        for (int j = 0; j < total + NR - 1; j += 2) {
            round(j, wa);
            if (j + 1 < total + NR - 1) round(j + 1, wb);
        }
    }
    if constexpr (FAST) flush();
}

// ---------------------------------------------------------------------------------------
// Round 5: the fused sum / WTA kernel for D = 256 in the fused layout (sum_wta_trap_kernel).
//
// TRAPEZOID RING.  The right view of pixel xr reads S[xr + d][d]: of the tile k tiles ahead only the disparities
// (k - 1) T < d < (k + 1) T.  The low disparities of a tile are dead long before the high ones are first needed, yet the
// uniform ring of sum_wta_lr_kernel keeps all D rows of a tile for NR rounds: at D = 256 that is 9 tiles of 32 pixels =
// 153 KB, a 512-thread block per CU (2 waves per SIMD, two barriers per round).  Here the disparity axis is cut into four
// bands of D / 4 (the disparities of 4 lanes) and band q keeps only the NR - (q * D / 4) / T tiles somebody can still ask
// for: with 64-pixel rounds 5 + 4 + 3 + 2 = 14 band slots of 8.4 KB (18 with a spare slot per band: one barrier per
// round), so that D = 256 runs the 1024-thread, 64-pixel shape of D = 192.
//
// BANKS.  ds_read_u16 / ds_write_b16 see 32 banks of 4 bytes and serve a wave in two halves of 32 lanes = 2 pixels x 16
// lanes (MI355X_MICROARCH.md, LDS).  In the uniform layout [d][T + 2] a lane's 16 rows put the lanes 528 dwords apart:
// = 16 (mod 32), two banks for 16 lanes, an 8-way conflict on every ring write and 4-way on the diagonal gather (that,
// not HBM, bounded the 32-pixel kernel: 3.1 TB/s).  A band slot here is lane-major with a lane pitch = 2 (mod 4) dwords
// (the 16 lanes of a pixel then sit on 16 distinct even banks, its neighbour pixel in the same dwords or on the odd
// banks), the slot pitch is a multiple of 32 dwords and the band bases continue the lane pitch: conflict-free writes
// and gathers.  (The same count for D = 192: lane pitch 396 = 12 (mod 32): 2-way writes, which the 4-cycle operand
// transfer of a store hides, and a conflict-free gather.)
// ---------------------------------------------------------------------------------------
template <int DPL, int T, int SP>
struct TrapRing {
    static constexpr int D = 16 * DPL, BW = 4 * DPL;    // four bands of four lanes
    static constexpr int NR = (T + D - 2) / T + 1;      // tiles a right pixel can touch (x .. x + D - 1)
    static constexpr int TW = T + 2;                    // row pitch in u16
    static constexpr int LPD0 = DPL * TW / 2;           // dwords of a lane's DPL rows
    static constexpr int LP = DPL * TW + 2 * ((2 - LPD0 % 4 + 4) % 4);  // lane pitch in u16: = 2 (mod 4) dwords
    static constexpr int BS = 4 * LP + 2 * ((32 - (2 * LP) % 32) % 32); // band slot pitch in u16: a multiple of 32 dwords
    static constexpr int nslots(int q) { return NR - (q * BW) / T + SP; }
    static constexpr int base(int q)                    // u16 offset of band q's ring; = q * 4 lane pitches (mod 32 dwords)
    {
        int s = 0;
        for (int i = 0; i < q; i++) s += nslots(i) * BS;
        return s + 2 * ((q * 2 * LP) % 32);
    }
    static constexpr int total = base(3) + nslots(3) * BS;
    static_assert(DPL * TW % 2 == 0 && (LP / 2) % 4 == 2 && (BS / 2) % 32 == 0, "bank layout");
};

template <int DPL, int T, bool SPARE>
__global__ void __launch_bounds__(16 * T) sum_wta_trap_kernel(VolPtrs vols, float *__restrict__ disp_l, float *__restrict__ disp_r,
                                                             int Hp, int Wp, u32 factor_uniq, int do_subpixel, int rows_per_block)
{
    constexpr int GL = 16, NV = 4, NP = DPL / 2, D = GL * DPL, NWD = NP / 2;
    typedef TrapRing<DPL, T, SPARE ? 1 : 0> RG;
    constexpr int NR = RG::NR, TW = RG::TW, BW = RG::BW, BS = RG::BS, LP = RG::LP;
    extern __shared__ __attribute__((aligned(16))) u16 ring_alloc[]; // T guard elements, then the band rings
    u16 *const ring = ring_alloc + T;
    const int f = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int nrows = (Hp - r0 < rows_per_block) ? Hp - r0 : rows_per_block;
    const size_t framepix = (size_t)f * Hp * Wp;
    const int g = threadIdx.x / GL, l16 = threadIdx.x % GL;
    const int dbase = DPL * l16;
    const int q = l16 >> 2, l4 = l16 & 3;                // this lane's band and its place in it
    int my_n = RG::nslots(0), my_base = RG::base(0);
    if (q == 1) { my_n = RG::nslots(1); my_base = RG::base(1); }
    if (q == 2) { my_n = RG::nslots(2); my_base = RG::base(2); }
    if (q == 3) { my_n = RG::nslots(3); my_base = RG::base(3); }
    my_base += l4 * LP;
    // uniform ring cursors per band: slot of the tile the left view writes now, slot of the right view's own tile
    int wcur[4] = {0, 0, 0, 0}, rcur[4] = {0, 0, 0, 0};
    auto sel4 = [&](int b, u32 v0, u32 v1, u32 v2, u32 v3) -> u32 {
        u32 v = v0;
        v = b == 1 ? v1 : v;
        v = b == 2 ? v2 : v;
        v = b == 3 ? v3 : v;
        return v;
    };
    // offset of disparity d inside a band slot (lane-major, then the lane's rows)
    auto in_slot = [&](int d, int b) -> u32 {
        const int r = d - b * BW;
        return (u32)((r / DPL) * LP + (r % DPL) * TW);
    };
    u32 kl[5] = {0, 0, 0, 0, 0}, kr[5] = {0, 0, 0, 0, 0};
    auto flush = [&]() {
        if (kl[3]) {
            const float dv = top2_final<DPL>(kl[0], kl[1], kl[2], (int)(kl[3] & 511u), D, factor_uniq, do_subpixel != 0, (kl[3] >> 9) & 1u);
            disp_l[framepix + kl[4]] = dv;
        }
        if (kr[3]) {
            const float dv = top2_final<DPL>(kr[0], kr[1], kr[2], (int)(kr[3] & 511u), D, factor_uniq, false, false);
            disp_r[framepix + kr[4]] = dv;
        }
        kl[3] = 0;
        kr[3] = 0;
    };
    const int ntiles = (Wp + T - 1) / T;
    const int total = nrows * ntiles;
    // stream cursors (uniform): fetch runs three rounds ahead of the left view, the right view NR-1 behind
    int f_row = r0, f_k = 0;
    int l_row = r0, l_k = 0;
    int r_row = r0, r_k = 0;
    u32 wa[NV][NWD], wb[NV][NWD], wc[NV][NWD];
    auto fetch = [&](u32 (&w)[NV][NWD]) {
        int x = T * f_k + g;
        x = x < Wp ? x : Wp - 1;
        const int frow = f_row < r0 + nrows ? f_row : r0 + nrows - 1;
        const size_t rowoff = (framepix + (size_t)frow * Wp) * D; // uniform: scalar base + 32-bit lane offset
        // cells with d > x + 1 have no reader: such a lane re-reads the pixel's first chunk (sum_wta_lr_kernel)
        const u32 laneoff = (u32)x * D + (dbase <= x + 1 ? dbase : 0);
#pragma unroll
        for (int v = 0; v < NV; v++) load_words_nt<NWD>((const u32 *)((const u8 *)vols.v[v] + rowoff + laneoff), w[v]);
        if (++f_k == ntiles) {
            f_k = 0;
            f_row++;
        }
    };
    auto round = [&](int j, u32 (&w)[NV][NWD]) {
        if (j < total) {
            const int x = T * l_k + g;
            u32 acc[NP];
            // fused layout: W + E as packed bytes (cannot carry), widened, + the two three-path sums (<= 255 each) widened
#pragma unroll
            for (int i = 0; i < NP / 2; i++) {
                const u32 ab = w[0][i] + w[1][i];
                acc[2 * i] = __builtin_amdgcn_perm(ab, ab, 0x0c010c00u) + __builtin_amdgcn_perm(w[2][i], w[2][i], 0x0c010c00u) +
                             __builtin_amdgcn_perm(w[3][i], w[3][i], 0x0c010c00u);
                acc[2 * i + 1] = __builtin_amdgcn_perm(ab, ab, 0x0c030c02u) + __builtin_amdgcn_perm(w[2][i], w[2][i], 0x0c030c02u) +
                                 __builtin_amdgcn_perm(w[3][i], w[3][i], 0x0c030c02u);
            }
            fetch(w); // this register set is free again
            if (x < Wp) { // uniform per 16-lane group
                const int wslot = (int)sel4(q, (u32)wcur[0], (u32)wcur[1], (u32)wcur[2], (u32)wcur[3]);
                u16 *t = ring + my_base + wslot * BS + g;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    t[(2 * i) * TW] = (u16)(acc[i] & 0xFFFFu);
                    t[(2 * i + 1) * TW] = (u16)(acc[i] >> 16);
                }
                const int n = (x < D - 1 ? x : D - 1) + 1;
                // the pixel's column was just written by this wave (LDS keeps a wave's accesses in order)
                const u32 wb0 = RG::base(0) + wcur[0] * BS, wb1 = RG::base(1) + wcur[1] * BS, wb2 = RG::base(2) + wcur[2] * BS,
                          wb3 = RG::base(3) + wcur[3] * BS;
                const Top2 r = top2_reduce<DPL, GL>(acc, dbase, n, D, [&](int d) -> u32 {
                    const int b = d / BW;
                    return ring[sel4(b, wb0, wb1, wb2, wb3) + in_slot(d, b) + (u32)g];
                });
                if (l16 == (j & (GL - 1))) { // this lane keeps this round's pixel
                    asm volatile("; keep (left view)");
                    kl[0] = r.g1; kl[1] = r.g2; kl[2] = r.nbv;
                    kl[3] = (u32)n | ((x >= 1 && x <= Wp - 2) ? 512u : 0u) | 1024u;
                    kl[4] = (u32)(l_row * Wp + x);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) wcur[b] = (wcur[b] + 1 == RG::nslots(b)) ? 0 : wcur[b] + 1;
            if (++l_k == ntiles) {
                l_k = 0;
                l_row++;
            }
        }
        __syncthreads();
        if (j >= NR - 1) {
            const int xr = T * r_k + g;
            if (xr < Wp) {
                const int n = (Wp - 1 - xr < D - 1 ? Wp - 1 - xr : D - 1) + 1;
                u32 sr[NP];
                // S[xr+d][d] for the lane's DPL consecutive d: the column g + d crosses a tile border at most once, so two base
                // addresses (this tile / the next, column wrapped) + a compile-time stride of one row + 1 serve all elements
                const u32 a0 = (u32)(g + dbase);
                int sA = (int)sel4(q, (u32)rcur[0], (u32)rcur[1], (u32)rcur[2], (u32)rcur[3]) + (int)(a0 / T);
                sA = sA >= my_n ? sA - my_n : sA;
                sA = sA >= my_n ? sA - my_n : sA;
                int sB = sA + 1;
                sB = sB >= my_n ? sB - my_n : sB;
                const int lane_base = my_base + (int)(a0 % T);
                const int RA = (int)__umul24((u32)sA, (u32)BS) + lane_base, RB = (int)__umul24((u32)sB, (u32)BS) + lane_base - T; // RB >= -T: guard elements
                const int e_cross = T - (int)(a0 % T); // first element that lies in the next tile
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const u32 lo = ring[(2 * i < e_cross ? RA : RB) + (2 * i) * (TW + 1)];
                    const u32 hi = ring[(2 * i + 1 < e_cross ? RA : RB) + (2 * i + 1) * (TW + 1)];
                    sr[i] = lo | (hi << 16);
                }
                // an arbitrary disparity's band, packed: base | slots << 20 | the right view's cursor << 24
                const u32 pb0 = (u32)RG::base(0) | ((u32)RG::nslots(0) << 20) | ((u32)rcur[0] << 24);
                const u32 pb1 = (u32)RG::base(1) | ((u32)RG::nslots(1) << 20) | ((u32)rcur[1] << 24);
                const u32 pb2 = (u32)RG::base(2) | ((u32)RG::nslots(2) << 20) | ((u32)rcur[2] << 24);
                const u32 pb3 = (u32)RG::base(3) | ((u32)RG::nslots(3) << 20) | ((u32)rcur[3] << 24);
                const Top2 r = top2_reduce<DPL, GL>(sr, dbase, n, D, [&](int d) -> u32 {
                    const int b = d / BW;
                    const u32 pk = sel4(b, pb0, pb1, pb2, pb3);
                    const int nb_ = (int)((pk >> 20) & 15u);
                    const u32 a = (u32)(g + d);
                    int sl = (int)(pk >> 24) + (int)(a / T);
                    sl = sl >= nb_ ? sl - nb_ : sl;
                    sl = sl >= nb_ ? sl - nb_ : sl;
                    return ring[(pk & 0xFFFFFu) + __umul24((u32)sl, (u32)BS) + in_slot(d, b) + (a % T)];
                });
                if (l16 == (j & (GL - 1))) {
                    asm volatile("; keep (right view)");
                    kr[0] = r.g1; kr[1] = r.g2; kr[2] = r.nbv;
                    kr[3] = (u32)n | 1024u;
                    kr[4] = (u32)(r_row * Wp + xr);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) rcur[b] = (rcur[b] + 1 == RG::nslots(b)) ? 0 : rcur[b] + 1;
            if (++r_k == ntiles) {
                r_k = 0;
                r_row++;
            }
        }
        if ((j & (GL - 1)) == GL - 1) flush(); // every lane of the wave holds a pixel of its own now
        if (!SPARE) __syncthreads();
    };
    fetch(wa);
    fetch(wb);
    fetch(wc);
    for (int j = 0; j < total + NR - 1; j += 3) {
        round(j, wa);
        if (j + 1 < total + NR - 1) round(j + 1, wb);
        if (j + 2 < total + NR - 1) round(j + 2, wc);
    }
    flush();
}

template <int DPL, int T, bool SPARE>
static int launch_trap(vppx_ctx *ctx, const VolPtrs &vp, int B, int Hp, int Wp, float *disp_l, float *disp_r, u32 fu, int sub)
{
    typedef TrapRing<DPL, T, SPARE ? 1 : 0> RG;
    const size_t lds = ((size_t)RG::total + T) * sizeof(u16);
    static bool attr_set[VPPX_MAX_DEVICES] = {}; // function attributes are per device
    static int ncu_of[VPPX_MAX_DEVICES] = {};
    const int dv = ctx->device & (VPPX_MAX_DEVICES - 1);
    if (!attr_set[dv]) {
        VPPX_HIP(hipFuncSetAttribute((const void *)sum_wta_trap_kernel<DPL, T, SPARE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipDeviceProp_t prop;
        VPPX_HIP(hipGetDeviceProperties(&prop, ctx->device));
        ncu_of[dv] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_set[dv] = true;
    }
    // one block per CU is resident (LDS): the rows-per-block split with the shortest makespan (launch_lr_t2)
    const int ncu = ncu_of[dv], ntiles = (Wp + T - 1) / T;
    int chunks = 1;
    long best = -1;
    for (int c = 1; c <= Hp && c <= 256; c++) {
        const long rows = (Hp + c - 1) / c;
        const long nb = (long)((Hp + rows - 1) / rows) * B;
        const long cost = ((nb + ncu - 1) / ncu) * (rows * ntiles + RG::NR + 2);
        if (best < 0 || cost < best) { best = cost; chunks = c; }
    }
    const int rpb = (Hp + chunks - 1) / chunks;
    chunks = (Hp + rpb - 1) / rpb;
    sum_wta_trap_kernel<DPL, T, SPARE><<<dim3(chunks, B), 16 * T, lds, ctx->stream>>>(vp, disp_l, disp_r, Hp, Wp, fu, sub, rpb);
    VPPX_CHECK_LAUNCH();
    return 0;
}

template <int DPL, typename IT, int NV, int T, bool SPARE, bool SW4, bool FAST = false, int GL = 16, bool FULLW = false>
static int launch_lr_t2(vppx_ctx *ctx, const VolPtrs &vp, int B, int Hp, int Wp, float *disp_l, float *disp_r, u32 fu, int sub)
{
    if constexpr (FAST && !FULLW) { // whole tiles only: the variant without per-pixel row-end tests
        if (Wp % T == 0) return launch_lr_t2<DPL, IT, NV, T, SPARE, SW4, FAST, GL, true>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub);
    }
    constexpr int D = GL * DPL;
    constexpr int NR = (T + D - 2) / T + 1;
    constexpr int NT = NR + (SPARE ? 1 : 0);
    const size_t lds = ((size_t)NT * D * (T + 2) + T) * sizeof(u16);
    static bool attr_set[VPPX_MAX_DEVICES] = {}; // function attributes are per device
    const int dv = ctx->device & (VPPX_MAX_DEVICES - 1);
    if (!attr_set[dv]) {
        VPPX_HIP(hipFuncSetAttribute((const void *)sum_wta_lr_kernel<DPL, IT, NV, T, SPARE, SW4, FAST, GL, FULLW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dv] = true;
    }
    // one block per CU is resident (LDS): pick the rows-per-block split with the shortest makespan,
    // (waves of blocks over the CUs) x (rounds per block incl. its fill/drain rounds)
    const int forced = ctx->knobs.sum_blocks;
    static int ncu_of[VPPX_MAX_DEVICES] = {};
    if (!ncu_of[dv]) {
        hipDeviceProp_t prop;
        VPPX_HIP(hipGetDeviceProperties(&prop, ctx->device));
        ncu_of[dv] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int ncu = ncu_of[dv];
    const int ntiles = (Wp + T - 1) / T;
    int chunks = 1;
    if (forced > 0) {
        chunks = (forced + B - 1) / B;
        chunks = chunks > Hp ? Hp : chunks;
    } else {
        long best = -1;
        for (int c = 1; c <= Hp && c <= 256; c++) {
            const long rows = (Hp + c - 1) / c;
            const long nb = (long)((Hp + rows - 1) / rows) * B;
            const long cost = ((nb + ncu - 1) / ncu) * (rows * ntiles + NR + 2);
            if (best < 0 || cost < best) { best = cost; chunks = c; }
        }
    }
    const int rpb = (Hp + chunks - 1) / chunks;
    chunks = (Hp + rpb - 1) / rpb;
    sum_wta_lr_kernel<DPL, IT, NV, T, SPARE, SW4, FAST, GL, FULLW><<<dim3(chunks, B), GL * T, lds, ctx->stream>>>(vp, disp_l, disp_r, Hp, Wp, fu, sub, rpb);
    VPPX_CHECK_LAUNCH();
    return 0;
}

template <int DPL, typename IT, int NV, int T, bool SPARE>
static int launch_lr_t(vppx_ctx *ctx, const VolPtrs &vp, int B, int Hp, int Wp, float *disp_l, float *disp_r, u32 fu, int sub,
                       bool sw4)
{
    if constexpr (sizeof(IT) == 1) {
        const int fast = !ctx->knobs.sum_general;
        // VPPX_VARIANT sum_gl8: 8 lanes per pixel with twice the disparities per lane for D = 128 / 192.  A fifth fewer
        // instructions, but one 512-thread block per CU (the LDS ring) is 2 waves per SIMD between two block-wide barriers
        // per round: 2.89 ms per B=32 launch against 2.44 with 16 lanes per pixel (4 waves per SIMD).  Bit-exact all the same.
        const int gl = ctx->knobs.sum_gl8 ? 8 : 16;
        if constexpr (DPL == 8 || DPL == 12) {
            if (sw4 && fast && gl == 8) return launch_lr_t2<2 * DPL, IT, NV, T, SPARE, true, true, 8>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub);
        }
        if (sw4 && fast) return launch_lr_t2<DPL, IT, NV, T, SPARE, true, true>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub);
        if (sw4) return launch_lr_t2<DPL, IT, NV, T, SPARE, true>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub);
    }
    return launch_lr_t2<DPL, IT, NV, T, SPARE, false>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub);
}

template <typename IT, int NV>
static int launch_lr_d(vppx_ctx *ctx, const VolPtrs &vp, int B, int Hp, int Wp, int D, float *disp_l, float *disp_r, u32 fu,
                       int sub, bool sw4)
{
    // D = 256 in the fused layout: the trapezoid ring (64-pixel rounds, 1024 threads) with a spare slot per band, i.e. one barrier
    // per round (155 KB of LDS; measured at 8 x 1536x2048: 5.1 ms against 5.5 without the spare slots -- VPPX_VARIANT sum_trap1,
    // 120 KB, two barriers -- and 8.2 on the uniform ring with 32-pixel rounds, sum_trap0)
    if constexpr (sizeof(IT) == 1 && NV == 4) {
        const int trap = ctx->knobs.sum_trap, fast = !ctx->knobs.sum_general;
        if (D == 256 && trap && fast && sw4) {
            if (trap == 1) return launch_trap<16, 64, false>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub);
            return launch_trap<16, 64, true>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub);
        }
    }
    // LDS ring: (tiles) x D x (T+2) u16 must fit 160 KiB
    if (D == 64) return launch_lr_t<4, IT, NV, 64, true>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub, sw4);
    if (D == 128) return launch_lr_t<8, IT, NV, 64, true>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub, sw4);
    if (D == 192) return launch_lr_t<12, IT, NV, 64, true>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub, sw4);
    return launch_lr_t<16, IT, NV, 32, false>(ctx, vp, B, Hp, Wp, disp_l, disp_r, fu, sub, sw4); // D = 256: 153 KiB
}

// LDS of one block of the fused sum / WTA kernel (its tile ring; launch_lr_d / launch_lr_t2): what a kernel that runs next to it
// cannot have
size_t rsgm_sum_lds_bytes(const vppx_ctx *ctx, int D, int nvol)
{
    if (D == 256 && nvol == 4 && ctx->knobs.sum_trap && !ctx->knobs.sum_general) // trapezoid ring (launch_lr_d)
        return ((size_t)(ctx->knobs.sum_trap == 1 ? TrapRing<16, 64, 0>::total : TrapRing<16, 64, 1>::total) + 64) * sizeof(u16);
    const int T = D == 256 ? 32 : 64;
    const int NT = (T + D - 2) / T + 1 + (D == 256 ? 0 : 1);
    return ((size_t)NT * D * (T + 2) + T) * sizeof(u16);
}

int rsgm_launch_sum_wta_lr(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const void *const *vols, int nvol, int elem_bytes,
                           float *disp_l, float *disp_r, u32 fu, int sub, int max_path_value)
{
    // byte arithmetic in the sum: 8 single-path volumes are added four at a time as packed bytes; 4 volumes (fused
    // vertical paths: W, E, and two three-path sums) add W + E as bytes and widen the rest
    const bool sw4 = elem_bytes == 1 && max_path_value > 0 && (nvol == 8 ? 4 * max_path_value <= 255 : 3 * max_path_value <= 255);
    if ((D != 64 && D != 128 && D != 192 && D != 256) || (nvol != 8 && nvol != 4)) return 1; // caller falls back
    VolPtrs vp;
    for (int i = 0; i < 8; i++) vp.v[i] = i < nvol ? vols[i] : nullptr;
    if (elem_bytes == 1) {
        if (nvol == 8) return launch_lr_d<u8, 8>(ctx, vp, B, Hp, Wp, D, disp_l, disp_r, fu, sub, sw4);
        return launch_lr_d<u8, 4>(ctx, vp, B, Hp, Wp, D, disp_l, disp_r, fu, sub, sw4);
    }
    if (nvol == 8) return launch_lr_d<u16, 8>(ctx, vp, B, Hp, Wp, D, disp_l, disp_r, fu, sub, false);
    return 1;
}

// ---------------------------------------------------------------------------------------
// stand-alone left WTA / sub-pixel on a materialised S (stage API: rsgm.py:141-142)
// ---------------------------------------------------------------------------------------
template <int DPL, bool SUBPIX_ONLY>
__global__ void __launch_bounds__(256) wta_left_kernel(const u16 *__restrict__ S, float *__restrict__ disp, int Hp, int Wp,
                                                       int D, u32 factor_uniq)
{
    constexpr int NP = DPL / 2;
    const size_t npix = (size_t)Hp * Wp;
    const size_t pl = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int f = blockIdx.z;
    if (pl >= npix) return;
    const size_t pix = (size_t)f * npix + pl;
    const int x = (int)(pl % Wp);
    const int l16 = threadIdx.x & 15;
    const int dbase = DPL * l16;
    u32 acc[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) acc[i] = (dbase + 2 * i < D) ? ((const u32 *)(S + pix * D + dbase))[i] : 0xFFFFFFFFu;
    if (!SUBPIX_ONLY) {
        const int n = (x < D - 1 ? x : D - 1) + 1;
        const float dv = wta_rows<DPL>(acc, dbase, n, D, factor_uniq, false, false);
        if (l16 == 0) disp[pix] = dv;
    } else {
        // subPixelRefine(dsi, disp, ..., method 0): refine an existing integer disparity
        const float dv = disp[pix];
        if (x >= 1 && x <= Wp - 2 && dv > 0.0f) {
            const int best = (int)dv;
            u32 cm1 = 0, c1 = 0, cp1 = 0;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const int d0 = dbase + 2 * i, d1 = d0 + 1;
                const u32 s0 = acc[i] & 0xFFFFu, s1 = acc[i] >> 16;
                if (d0 == best - 1) cm1 = s0;
                if (d1 == best - 1) cm1 = s1;
                if (d0 == best) c1 = s0;
                if (d1 == best) c1 = s1;
                if (d0 == best + 1) cp1 = s0;
                if (d1 == best + 1) cp1 = s1;
            }
            cm1 = row_or_u32(cm1); c1 = row_or_u32(c1); cp1 = row_or_u32(cp1);
            if (best >= 1 && best <= D - 2) {
                const int c0 = (int)cm1, cc = (int)c1, c2 = (int)cp1;
                const int den = (c2 < c0) ? c0 - cc : c2 - cc;
                if (den != 0 && l16 == 0)
                    disp[pix] = (float)best + __fdiv_rn((float)(c0 - c2), __fmul_rn(2.0f, (float)den));
            }
        }
    }
}

int rsgm_launch_wta_left(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *S, float *disp, u32 fu)
{
    const size_t npix = (size_t)Hp * Wp;
    dim3 grid((unsigned)((npix + 15) / 16), 1, B);
    switch (dpl_for(D)) {
    case 4: wta_left_kernel<4, false><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, fu); break;
    case 8: wta_left_kernel<8, false><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, fu); break;
    case 12: wta_left_kernel<12, false><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, fu); break;
    default: wta_left_kernel<16, false><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, fu); break;
    }
    VPPX_CHECK_LAUNCH();
    return 0;
}

int rsgm_launch_subpixel(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *S, float *disp)
{
    const size_t npix = (size_t)Hp * Wp;
    dim3 grid((unsigned)((npix + 15) / 16), 1, B);
    switch (dpl_for(D)) {
    case 4: wta_left_kernel<4, true><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, 0); break;
    case 8: wta_left_kernel<8, true><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, 0); break;
    case 12: wta_left_kernel<12, true><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, 0); break;
    default: wta_left_kernel<16, true><<<grid, 256, 0, ctx->stream>>>(S, disp, Hp, Wp, D, 0); break;
    }
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// right-view WTA (call site rsgm.py:170): d_R(y,x) = argmin_d S[y, x+d, d], same uniqueness rule.
// Two readers:
//   wta_right_t_kernel : from the d-major volume ST[y][d][x] (fused path): one thread per right
//                        pixel, ST[y][d][xr+d] is coalesced across xr; single pass over d.
//   wta_right_kernel   : from the reference-layout S[y][x][d] (stage API): diagonal band staged
//                        through LDS.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wta_right_t_kernel(const u16 *__restrict__ ST, float *__restrict__ disp, int Hp, int Wp,
                                                          int D, u32 factor_uniq)
{
    const int xr = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (xr >= Wp) return;
    const u16 *row = ST + ((size_t)f * Hp + y) * D * Wp + xr; // element d at row[d*Wp + d]
    const int n = (Wp - 1 - xr < D - 1 ? Wp - 1 - xr : D - 1) + 1;
    u32 minc = row[0], sec = 65535u, prev = 0, cm1 = 0, cp1 = 0;
    int best = 0;
    bool want_next = true; // the candidate after a new best is its d+1 neighbour
    for (int d = 1; d < n; d++) {
        const u32 c = row[(size_t)d * Wp + d];
        if (want_next) { cp1 = c; want_next = false; }
        if (c < minc) {       // strictly smaller: first minimum wins ties
            sec = min(sec, minc);
            cm1 = (d == 1) ? row[0] : prev;
            minc = c;
            best = d;
            want_next = true;
        } else {
            sec = min(sec, c);
        }
        prev = c;
    }
    bool ok = (1024u * minc <= sec * factor_uniq);
    if (!ok) {
        if (best > 0 && cm1 == sec) ok = true;
        if (best + 1 < n && cp1 == sec) ok = true;
    }
    disp[((size_t)f * Hp + y) * Wp + xr] = ok ? (float)best : INVALID_DISP;
}

int rsgm_launch_wta_right_t(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *ST, float *disp, u32 fu)
{
    dim3 grid((Wp + 255) / 256, Hp, B);
    wta_right_t_kernel<<<grid, 256, 0, ctx->stream>>>(ST, disp, Hp, Wp, D, fu);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// stage-API reader (reference layout):
// A block owns XR consecutive right pixels of one row.  The diagonal band of S it needs is
// staged through LDS with coalesced reads: for source pixel xs the needed disparities are
// d in [xs-x0-XR+1, xs-x0] (contiguous), so every S element is read exactly once overall.
// LDS image: band[xs_local][j], j = d - (xs_local - XR + 1), i.e. band[xs_local][j] belongs
// to right pixel xr_local = XR-1-j.
// ---------------------------------------------------------------------------------------
#define XR 64
__global__ void __launch_bounds__(256) wta_right_kernel(const u16 *__restrict__ S, float *__restrict__ disp, int Hp, int Wp,
                                                        int D, u32 factor_uniq)
{
    extern __shared__ __attribute__((aligned(16))) u16 band[]; // [(XR + D - 1)][XR]
    const int x0 = blockIdx.x * XR;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    const size_t rowbase = ((size_t)f * Hp + y) * Wp;
    const int nsrc = XR + D - 1;
    // stage: XR entries per source pixel
    for (int e = threadIdx.x; e < nsrc * XR; e += blockDim.x) {
        const int xsl = e / XR, j = e % XR;
        const int xs = x0 + xsl;
        const int d = xsl - XR + 1 + j;
        u16 v = 0xFFFF;
        if (xs < Wp && d >= 0 && d < D) v = S[(rowbase + xs) * D + d];
        band[e] = v;
    }
    __syncthreads();
    // 4 threads per right pixel: thread q handles d = q, q+4, ...
    const int xrl = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int xr = x0 + xrl;
    const int n = xr < Wp ? ((Wp - 1 - xr < D - 1 ? Wp - 1 - xr : D - 1) + 1) : 0;
    // cost of (xr, d) = band[xrl + d][XR-1-xrl]
    const int j = XR - 1 - xrl;
    u32 key = 0xFFFFFFFFu;
    for (int d = q; d < n; d += 4) {
        const u32 c = band[(xrl + d) * XR + j];
        key = min(key, (c << 8) | (u32)d);
    }
    // reduce over the 4 threads (quad_perm DPP)
    key = min(key, (u32)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xF, 0xF, false)); // quad_perm [1,0,3,2]
    key = min(key, (u32)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xF, 0xF, false)); // quad_perm [2,3,0,1]
    const int best = (int)(key & 0xFFu);
    const u32 minc = key >> 8;
    u32 sec = 65535u;
    for (int d = q; d < n; d += 4) {
        if (d != best) sec = min(sec, (u32)band[(xrl + d) * XR + j]);
    }
    sec = min(sec, (u32)__builtin_amdgcn_update_dpp(0, (int)sec, 0xB1, 0xF, 0xF, false));
    sec = min(sec, (u32)__builtin_amdgcn_update_dpp(0, (int)sec, 0x4E, 0xF, 0xF, false));
    if (q == 0 && xr < Wp) {
        bool ok = (1024u * minc <= sec * factor_uniq);
        if (!ok) {
            if (best > 0 && (u32)band[(xrl + best - 1) * XR + j] == sec) ok = true;
            if (best + 1 < n && (u32)band[(xrl + best + 1) * XR + j] == sec) ok = true;
        }
        disp[rowbase + xr] = ok ? (float)best : INVALID_DISP;
    }
}

int rsgm_launch_wta_right(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *S, float *disp, u32 fu)
{
    dim3 grid((Wp + XR - 1) / XR, Hp, B);
    const size_t lds = (size_t)(XR + D - 1) * XR * sizeof(u16);
    wta_right_kernel<<<grid, 256, lds, ctx->stream>>>(S, disp, Hp, Wp, D, fu);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// median3x3 on float32 (call sites rsgm.py:145,173): exact median inside, border copied.
// ---------------------------------------------------------------------------------------
// Median of nine as order statistics of the three sorted columns: med3(max of the minima, med of the medians, min of the
// maxima) -- 13 three-operand min / med / max instructions (v_min3_f32, v_med3_f32, v_max3_f32) instead of a 19-exchange
// network's 38.  The median is one of the nine values whatever the method, so the bits are the same.
__device__ __forceinline__ float median9(const float (&v)[9])
{
    float lo[3], md[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float a = v[c], b = v[3 + c], d = v[6 + c];
        lo[c] = fminf(fminf(a, b), d);
        md[c] = __builtin_amdgcn_fmed3f(a, b, d);
        hi[c] = fmaxf(fmaxf(a, b), d);
    }
    return __builtin_amdgcn_fmed3f(fmaxf(fmaxf(lo[0], lo[1]), lo[2]), __builtin_amdgcn_fmed3f(md[0], md[1], md[2]),
                                   fminf(fminf(hi[0], hi[1]), hi[2]));
}
__global__ void __launch_bounds__(256) median3x3_kernel(const float *__restrict__ src, float *__restrict__ dst, int Hp, int Wp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (x >= Wp) return;
    const float *s = src + (size_t)f * Hp * Wp;
    float out;
    if (y == 0 || y == Hp - 1 || x == 0 || x == Wp - 1) {
        out = s[(size_t)y * Wp + x];
    } else {
        float v[9];
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) v[(dy + 1) * 3 + dx + 1] = s[(size_t)(y + dy) * Wp + x + dx];
        out = median9(v);
    }
    dst[((size_t)f * Hp + y) * Wp + x] = out;
}

int rsgm_launch_median(vppx_ctx *ctx, int B, int Hp, int Wp, const float *src, float *dst)
{
    dim3 grid((Wp + 255) / 256, Hp, B);
    median3x3_kernel<<<grid, 256, 0, ctx->stream>>>(src, dst, Hp, Wp);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// _linear_interpolate(dmap, 15, 3) (rsgm.py:67-113) + np.clip(.,0,None) (:151,179).
// The reference walks each row in place.  A fill only happens for a gap (maximal run of
// values <= 0 between two valid pixels xl < xr of the same row) with xr-xl <= 14, and it is
// triggered by the first hole pixel, in raster order, that sees both ends inside its +-7
// window: x_t = max(xl+1, xr-7).  The line (m, q) is formed in float64 relative to x_t and
// every pixel of [xl, xr] is overwritten with (float)(m*(x-x_t)+q); the two end points come
// back bit-identical (the float64 error is far below half a float32 ulp), so gaps are
// independent and every hole pixel can evaluate its own value: one thread per pixel.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) linear_interp_clip_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                 int Hp, int Wp)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (x >= Wp) return;
    const float *row = src + ((size_t)f * Hp + y) * Wp;
    float v = row[x];
    if (v <= 0) {
        int xl = -1, xr = -1;
        for (int k = 1; k <= 13; k++)
            if (x - k >= 0 && row[x - k] > 0) { xl = x - k; break; }
        if (xl >= 0) {
            for (int k = 1; k <= 14 - (x - xl); k++)
                if (x + k < Wp && row[x + k] > 0) { xr = x + k; break; }
        }
        if (xl >= 0 && xr >= 0) { // xr - xl <= 14 by construction
            const double n_left = (double)row[xl], n_right = (double)row[xr];
            if (fabs(n_left - n_right) < 3.0) {
                const int xt = max(xl + 1, xr - 7);
                const int n_leftx = xl - xt, n_rightx = xr - xt;
                const double m = __ddiv_rn(n_right - n_left, (double)(n_rightx - n_leftx));
                const double q = __dsub_rn(n_left, __dmul_rn(m, (double)n_leftx));
                v = (float)__dadd_rn(__dmul_rn(m, (double)(x - xt)), q);
            }
        }
    }
    dst[((size_t)f * Hp + y) * Wp + x] = (v >= 0) ? v : 0.0f; // np.clip(., 0, None)
}

int rsgm_launch_linear_interp_clip(vppx_ctx *ctx, int B, int Hp, int Wp, const float *src, float *dst)
{
    dim3 grid((Wp + 255) / 256, Hp, B);
    linear_interp_clip_kernel<<<grid, 256, 0, ctx->stream>>>(src, dst, Hp, Wp);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// Fused median3x3 -> _linear_interpolate -> clip for the left and the right view (rsgm.py:145-151,
// 173-179): one block per row; the row's medians go to LDS, together with a bitmap of the valid (> 0) ones; a hole pixel
// finds its nearest valid neighbours (at most 13 to the left, then as far to the right as the reference's window of 14
// allows) with two bit scans of a 27-bit window of that bitmap instead of two loops over LDS -- the pad columns make every
// row end in holes, and one hole keeps its whole wave in those loops.  (Same arithmetic as the two stand-alone kernels
// above, which the stage API keeps using.)
__global__ void __launch_bounds__(256) median_interp_clip_kernel(ImgSet io, int B, int Hp, int Wp)
{
    extern __shared__ __attribute__((aligned(16))) float s_med[]; // [Wp] medians, then the bitmap: [Wp / 64 + 3] 64-bit words
    const int nwords = (Wp + 63) / 64;
    unsigned long long *s_bits = (unsigned long long *)(s_med + ((Wp + 1) & ~1)); // word 0 and the last one stay empty
    const int y = blockIdx.x;
    const int set = blockIdx.y / B, f = blockIdx.y % B;
    const int lane = threadIdx.x & 63;
    const float *s = (const float *)(set == 0 ? io.src[0] : io.src[1]) + (size_t)f * Hp * Wp;
    float *dst = (float *)(set == 0 ? io.dst[0] : io.dst[1]) + ((size_t)f * Hp + y) * Wp;
    if (threadIdx.x == 0) s_bits[0] = 0ull, s_bits[nwords + 1] = 0ull;
    for (int xb = threadIdx.x & ~63; xb < Wp; xb += 256) { // (whole waves: the ballot below wants every lane)
        const int x = xb + lane;
        float out = 0.0f;
        if (x < Wp) {
            if (y == 0 || y == Hp - 1 || x == 0 || x == Wp - 1) {
                out = s[(size_t)y * Wp + x];
            } else {
                float v[9];
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) v[(dy + 1) * 3 + dx + 1] = s[(size_t)(y + dy) * Wp + x + dx];
                out = median9(v);
            }
            s_med[x] = out;
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(x < Wp && out > 0);
        if (lane == 0) s_bits[1 + (xb >> 6)] = bal;
    }
    __syncthreads();
    const float *row = s_med;
    const u32 *b32 = (const u32 *)s_bits;
    for (int x = threadIdx.x; x < Wp; x += 256) {
        float v = row[x];
        if (v <= 0) {
            // bit j of w: is pixel x - 13 + j valid (j = 0..26; pixels outside the row read as invalid)
            const int sb = 64 + x - 13;
            const u32 w = __builtin_amdgcn_alignbit(b32[(sb >> 5) + 1], b32[sb >> 5], (u32)(sb & 31)) & 0x7FFFFFFu;
            const u32 wl = w & 0x1FFFu; // x - 13 .. x - 1
            int xl = -1, xr = -1;
            if (wl) {
                xl = x - 13 + (31 - __builtin_clz(wl));
                // x + k with 1 <= k <= 14 - (x - xl): bits 14 .. 27 - (x - xl)
                const int top = 27 - (x - xl);
                const u32 wr = (w >> 14) & ((1u << (top - 13)) - 1u);
                if (wr) xr = x + 1 + __builtin_ctz(wr);
            }
            if (xl >= 0 && xr >= 0) {
                const double n_left = (double)row[xl], n_right = (double)row[xr];
                if (fabs(n_left - n_right) < 3.0) {
                    const int xt = max(xl + 1, xr - 7);
                    const int n_leftx = xl - xt, n_rightx = xr - xt;
                    const double m = __ddiv_rn(n_right - n_left, (double)(n_rightx - n_leftx));
                    const double q = __dsub_rn(n_left, __dmul_rn(m, (double)n_leftx));
                    v = (float)__dadd_rn(__dmul_rn(m, (double)(x - xt)), q);
                }
            }
        }
        dst[x] = (v >= 0) ? v : 0.0f; // np.clip(., 0, None)
    }
}

int rsgm_launch_median_interp_clip(vppx_ctx *ctx, int B, int Hp, int Wp, const float *src_l, float *dst_l, const float *src_r,
                                   float *dst_r)
{
    ImgSet io;
    io.src[0] = src_l; io.dst[0] = dst_l; io.src[1] = src_r; io.dst[1] = dst_r; io.src[2] = nullptr; io.dst[2] = nullptr;
    const size_t lds = (size_t)((Wp + 1) & ~1) * sizeof(float) + (size_t)((Wp + 63) / 64 + 3) * sizeof(unsigned long long);
    median_interp_clip_kernel<<<dim3(Hp, 2 * B), 256, lds, ctx->stream>>>(io, B, Hp, Wp);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// post-processing of compute_rsgm (rsgm.py:275-292) on the cropped frame:
//   crop -> left/right check (:230-248) -> zero mask==128 -> astype(uint8) -> filterSpeckles
//   (0,200,10) -> astype(float32) -> restore sub-pixel -> _interpolate_background (:185-227)
// ---------------------------------------------------------------------------------------
// Connected components for cv2.filterSpeckles (4-connectivity, |a-b| <= maxDiff, pixels equal to newVal excluded):
// tile-local labelling in LDS, union-find in global memory only across tile borders (kernels below).
__device__ __forceinline__ int uf_find(int *label, int i)
{
    int p = label[i];
    while (p != i) {
        i = p;
        p = label[i];
    }
    return i;
}
// the same with path halving, for the union pass: most of a frame is ONE component (neighbouring disparities differ by
// less than maxDiff almost everywhere), so without it every union walks the chain of that component's runs.  A label
// only ever decreases, and whoever lowers a non-root label with atomicMin goes on to unite the parent it displaced
// (uf_union), so shortcutting i to its grandparent -- with atomicMin as well: never undoes a lower label set meanwhile --
// keeps every pixel in its component.
__device__ __forceinline__ int uf_find_halve(int *label, int i)
{
    int p = label[i];
    while (p != i) {
        const int gp = label[p];
        if (gp != p) atomicMin(&label[i], gp);
        i = p;
        p = gp;
    }
    return i;
}
__device__ __forceinline__ void uf_union(int *label, int a, int b)
{
    while (true) {
        a = uf_find_halve(label, a);
        b = uf_find_halve(label, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; } // a > b : hang a under b
        const int old = atomicMin(&label[a], b);
        if (old == a) return;
        a = old;
    }
}
__device__ __forceinline__ bool px_link(int a, int b, int new_val, int max_diff)
{
    return a != new_val && b != new_val && abs(a - b) <= max_diff;
}

// speckle_tile_kernel: crop + left/right check + the connected components INSIDE a 64 x 32 tile, in LDS (a wave owns whole
// tile rows: horizontal runs by one segmented max-scan per row, vertical links by union-find on the run starts with LDS
// atomics, component sizes by one LDS add per run).  Every pixel gets the frame index of its tile-local root, the root
// pixel the local component's size (0 everywhere else).
// speckle_border_kernel: unions across tile borders only (a few thousand links per frame instead of one per vertical run
// contact); speckle_total_kernel: every local root adds its size to its global root; speckle_apply_kernel: components of at most
// maxSpeckleSize pixels become newVal, float32 again, sub-pixel values restored (rsgm.py:285-290).
// (Rounds 1-2 labelled whole frames in global memory -- row runs, one union per vertical run contact, a size per run: 7 launches,
// 0.52 ms per 32 frames and 0.10 ms for one frame, against 4 launches, 0.25 / 0.04 ms.)
#define SPK_TW 64
#define SPK_TH 32
__device__ __forceinline__ int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_find(const int *lab, int i)
{
    int p = lds_ld(&lab[i]);
    while (p != i) {
        i = p;
        p = lds_ld(&lab[i]);
    }
    return i;
}
__device__ __forceinline__ void lds_union(int *lab, int a, int b)
{
    while (true) {
        a = lds_find(lab, a);
        b = lds_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; } // a > b: hang a under b (labels only ever decrease)
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ void __launch_bounds__(256) speckle_tile_kernel(const float *__restrict__ dl, const float *__restrict__ dr,
                                                           u8 *__restrict__ fd8, int *__restrict__ label, int *__restrict__ size,
                                                           int *__restrict__ roots, int *__restrict__ nroots, int H, int W, int Hp,
                                                           int Wp, int pad_t, int pad_l, int new_val, int max_diff)
{
    __shared__ int s_lab[SPK_TH * SPK_TW];
    __shared__ int s_cnt[SPK_TH * SPK_TW];
    __shared__ u8 s_val[SPK_TH * SPK_TW];
    __shared__ int s_nroot;
    if (threadIdx.x == 0) s_nroot = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tx0 = blockIdx.x * SPK_TW, ty0 = blockIdx.y * SPK_TH, f = blockIdx.z;
    const int x = tx0 + lane;
    constexpr int RPW = SPK_TH / 4; // tile rows per wave
    int vv[RPW], ss[RPW];
    // (the wave's eight left disparities, then the eight right ones they point at: unconditional loads at clamped
    // positions, two round trips through memory instead of sixteen -- hipcc serialises a load behind a branch)
    float dvs[RPW], rvs[RPW];
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int y = min(ty0 + wv * RPW + k, H - 1);
        dvs[k] = dl[((size_t)f * Hp + y + pad_t) * Wp + pad_l + min(x, W - 1)];
    }
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int y = min(ty0 + wv * RPW + k, H - 1);
        const int d = dvs[k] > 0 ? (int)rintf(dvs[k]) : 0; // round half to even (numba round)
        rvs[k] = dr[((size_t)f * Hp + y + pad_t) * Wp + pad_l + min(max(min(x, W - 1) - d, 0), W - 1)];
    }
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wv * RPW + k, y = ty0 + r;
        int v = new_val;
        if (x < W && y < H) {
            // crop + left/right check (rsgm.py:230-248, 275-283)
            const float dv = dvs[k];
            float keep = dv;
            if (dv > 0) {
                const int d = (int)rintf(dv);
                const int xd = x - d;
                if (0 <= xd && xd <= W - 1) {
                    const float rv = rvs[k];
                    if (rv > 0 && fabsf(__fsub_rn(dv, rv)) > 1.0f) keep = 0; // mask 128
                } else {
                    keep = 0; // mask 128
                }
            }
            const u8 k8 = (u8)keep; // astype(np.uint8): truncation, values in [0,256)
            fd8[((size_t)f * H + y) * W + x] = k8;
            v = k8;
        }
        const int vprev = __shfl_up(v, 1);
        const bool in = v != new_val;
        const bool start = in && !(lane > 0 && px_link(vprev, v, new_val, max_diff));
        int s = start ? lane : -1; // inclusive max-scan of run starts over the tile row
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(s, off);
            if (lane >= off) s = max(s, t);
        }
        vv[k] = v;
        ss[k] = s;
        const int p = r * SPK_TW + lane;
        s_val[p] = (u8)v;
        s_lab[p] = in ? r * SPK_TW + s : p;
        s_cnt[p] = 0;
    }
    __syncthreads();
    // vertical links: one union per place where a link between two runs begins
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wv * RPW + k, p = r * SPK_TW + lane;
        if (r == 0) continue;
        const int a = s_val[p - SPK_TW], b = vv[k];
        if (!px_link(a, b, new_val, max_diff)) continue;
        if (lane > 0) {
            const int a0 = s_val[p - SPK_TW - 1], b0 = s_val[p - 1];
            if (px_link(a0, b0, new_val, max_diff) && px_link(a0, a, new_val, max_diff) && px_link(b0, b, new_val, max_diff)) continue;
        }
        lds_union(s_lab, lds_ld(&s_lab[p - SPK_TW]), r * SPK_TW + ss[k]);
    }
    __syncthreads();
    // sizes: the last pixel of a run adds the run's length to the root
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wv * RPW + k, p = r * SPK_TW + lane;
        const int v = vv[k];
        if (v == new_val) continue;
        const int vnext = lane < 63 ? (int)s_val[p + 1] : new_val;
        if (lane < 63 && px_link(v, vnext, new_val, max_diff)) continue;
        atomicAdd(&s_cnt[lds_find(s_lab, r * SPK_TW + ss[k])], lane - ss[k] + 1);
    }
    __syncthreads();
    // labels for every pixel; the size only AT the local roots, which also go on the tile's root list (speckle_total_kernel
    // walks those lists instead of a whole image of mostly-zero sizes)
    const int tile = (f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    int *rl = roots + (size_t)tile * (SPK_TW * SPK_TH);
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = wv * RPW + k, y = ty0 + r, p = r * SPK_TW + lane;
        const bool inside = x < W && y < H;
        const int v = vv[k];
        const int root = v != new_val ? lds_find(s_lab, r * SPK_TW + ss[k]) : p;
        const size_t o = ((size_t)f * H + min(y, H - 1)) * W + min(x, W - 1);
        const bool is_root = inside && v != new_val && root == p;
        if (inside) label[o] = (ty0 + root / SPK_TW) * W + tx0 + root % SPK_TW;
        if (is_root) size[o] = s_cnt[p];
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(is_root);
        if (bal) {
            const int leader = __builtin_ctzll(bal);
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_nroot, (int)__popcll(bal));
            base = __builtin_amdgcn_readlane(base, leader);
            if (is_root) rl[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = y * W + x;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) nroots[tile] = s_nroot;
}

__global__ void __launch_bounds__(256) speckle_border_kernel(const u8 *__restrict__ img, int *__restrict__ label, int H, int W,
                                                             int new_val, int max_diff)
{
    const int f = blockIdx.y;
    const u8 *im = img + (size_t)f * H * W;
    int *lb = label + (size_t)f * H * W;
    const int nby = (H - 1) / SPK_TH, nbx = (W - 1) / SPK_TW; // tile borders with pixels on both sides
    int idx = blockIdx.x * 256 + threadIdx.x;
    int ia, ib, ja = -1, jb = -1; // the pair, and the pair before it along the border (same two tiles only)
    if (idx < nby * W) {
        const int x = idx % W, y = SPK_TH * (idx / W + 1) - 1;
        ia = y * W + x;
        ib = ia + W;
        if (x % SPK_TW != 0) { ja = ia - 1; jb = ib - 1; }
    } else {
        idx -= nby * W;
        if (idx >= nbx * H) return;
        const int y = idx % H, x = SPK_TW * (idx / H + 1) - 1;
        ia = y * W + x;
        ib = ia + 1;
        if (y % SPK_TH != 0) { ja = ia - W; jb = ib - W; }
    }
    const int a = im[ia], b = im[ib];
    if (!px_link(a, b, new_val, max_diff)) return;
    if (ja >= 0) { // the pair before links the same two tile-local components: skip (never across a tile corner: the links
                   // this argument leans on must be ones the tile kernel has already made)
        const int a0 = im[ja], b0 = im[jb];
        if (px_link(a0, b0, new_val, max_diff) && px_link(a0, a, new_val, max_diff) && px_link(b0, b, new_val, max_diff)) return;
    }
    uf_union(lb, lb[ia], lb[ib]);
}

// every tile-local root adds its component's size to its global root (all unions are done: roots are final); one wave per
// tile walks the tile's root list
__global__ void __launch_bounds__(64) speckle_total_kernel(int *__restrict__ label, int *__restrict__ size, const int *__restrict__ roots,
                                                           const int *__restrict__ nroots, size_t npix_frame, int tiles_per_frame)
{
    const int tile = blockIdx.x;
    const size_t base = (size_t)(tile / tiles_per_frame) * npix_frame;
    int *lb = label + base;
    const int n = nroots[tile];
    const int *rl = roots + (size_t)tile * (SPK_TW * SPK_TH);
    for (int e = threadIdx.x; e < n; e += 64) {
        const int me = rl[e];
        const int sz = size[base + me];
        int c = me, p = lb[c];
        while (p != c) { // path halving (shortcuts to an ancestor are benign under concurrency)
            const int gp = lb[p];
            if (gp != p) lb[c] = gp;
            c = gp;
            p = lb[c];
        }
        if (c != me) {
            lb[me] = c;
            atomicAdd(&size[base + c], sz);
        }
    }
}

// speckle filter result (components of at most maxSpeckleSize pixels become newVal, float32 again, sub-pixel values restored,
// rsgm.py:285-290) fused with the row pass of _interpolate_background (rsgm.py:189-214): one block per row, the filtered row
// only ever exists in LDS.  Fills only touch invalid (<= 0) pixels and read only originally valid ones, so per pixel: nearest
// valid to the left (xl) and to the right (xr) of the ORIGINAL row; both -> min(v[xl], v[xr]); only one -> that one (border
// extension); none -> unchanged.  Each thread owns a contiguous segment of the row; the nearest valid index outside the
// segment comes from one wave-level max/min scan (+ 4 wave carries).  A row with one valid pixel ends up valid everywhere;
// row_any[] says which rows are, for the column pass.
template <int SEGMAX> // pixels per thread: 8 covers W <= 2048, 32 covers W <= 8192
__global__ void __launch_bounds__(256) speckle_apply_rows_kernel(const u8 *__restrict__ fd8, const int *__restrict__ label,
                                                                  const int *__restrict__ size, const float *__restrict__ dl_pad,
                                                                  float *__restrict__ out, u8 *__restrict__ row_any, int H, int W, int Hp,
                                                                  int Wp, int pad_t, int pad_l, int new_val, int max_size, int subpixel)
{
    extern __shared__ __attribute__((aligned(16))) float sh_f[]; // [W] filtered row, [W] result
    __shared__ int s_l[4], s_r[4];
    float *val = sh_f, *res = sh_f + W;
    const int y = blockIdx.x, f = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t base = (size_t)f * H * W, rowo = base + (size_t)y * W;
    const int *lb = label + base;
    const float *dlr = dl_pad + ((size_t)f * Hp + y + pad_t) * Wp + pad_l;
    float *g = out + rowo;
    // (four pixels per thread and round, every level of the label chase for all four before the next one: the loads are
    // unconditional -- at clamped positions past the end of the row -- so that they are in flight together)
    for (int xb = threadIdx.x; xb < W; xb += 4 * 256) {
        int v[4], r1[4], r2[4];
        float sub[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int x = min(xb + 256 * u, W - 1);
            v[u] = fd8[rowo + x];
            r1[u] = label[rowo + x];
            sub[u] = dlr[x];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) r2[u] = lb[r1[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int r = r1[u], pr = r2[u];
            while (pr != r) {
                r = pr;
                pr = lb[r];
            }
            r1[u] = r;
        }
        int sz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) sz[u] = size[base + r1[u]]; // (meaningless where v == new_val: not used there)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int x = xb + 256 * u;
            int vv = v[u];
            if (vv != new_val && sz[u] <= max_size) vv = new_val;
            float fv = (float)vv;                      // rsgm.py:286
            if (subpixel && fv != 0.0f) fv = sub[u];   // rsgm.py:289-290
            if (x < W) val[x] = fv;
        }
    }
    __syncthreads();
    const int SEG = (W + 255) / 256; // <= SEGMAX
    const int x0 = threadIdx.x * SEG;
    int lastv = -1, firstv = 0x7FFFFFFF;
    for (int j = 0; j < SEG; j++) {
        const int x = x0 + j;
        if (x < W && val[x] > 0) {
            lastv = x;
            if (firstv == 0x7FFFFFFF) firstv = x;
        }
    }
    int sl = lastv, sr = firstv;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int tl = __shfl_up(sl, off), tr = __shfl_down(sr, off);
        if (lane >= off) sl = max(sl, tl);
        if (lane + off < 64) sr = min(sr, tr);
    }
    if (lane == 63) s_l[wv] = sl;
    if (lane == 0) s_r[wv] = sr;
    __syncthreads();
    if (threadIdx.x == 0) row_any[(size_t)f * H + y] = max(max(s_l[0], s_l[1]), max(s_l[2], s_l[3])) >= 0 ? 1 : 0;
    int cl = __shfl_up(sl, 1), cr = __shfl_down(sr, 1);
    if (lane == 0) cl = -1;
    if (lane == 63) cr = 0x7FFFFFFF;
    for (int w = 0; w < wv; w++) cl = max(cl, s_l[w]);
    for (int w = wv + 1; w < 4; w++) cr = min(cr, s_r[w]);
    int xl[SEGMAX];
#pragma unroll
    for (int j = 0; j < SEGMAX; j++) {
        const int x = x0 + j;
        if (j < SEG && x < W && val[x] > 0) cl = x;
        xl[j] = cl;
    }
#pragma unroll
    for (int j = SEGMAX - 1; j >= 0; j--) {
        const int x = x0 + j;
        if (j >= SEG || x >= W) continue;
        float v = val[x];
        if (v > 0) {
            cr = x;
        } else {
            const bool hl = xl[j] >= 0, hr = cr < W;
            if (hl && hr) {
                const float a = val[xl[j]], b = val[cr];
                v = b < a ? b : a; // Python min(a, b)
            } else if (hr) {
                v = val[cr];
            } else if (hl) {
                v = val[xl[j]];
            }
        }
        res[x] = v;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) g[x] = res[x]; // (res == val where val is valid)
}
// columns (rsgm.py:216-227): per column, rows above the first valid row take its value, rows below the last valid row take
// that one's.  After the row pass a row is valid everywhere or nowhere, so the first and the last valid row are the same
// for every column: the rows above the first non-empty row become copies of it, the rows below the last one copies of
// that (usually there are none).  One block per frame.
__global__ void __launch_bounds__(256) interp_bg_edge_rows_kernel(float *__restrict__ dm, const u8 *__restrict__ row_any, int H, int W)
{
    __shared__ int s_first[4], s_last[4];
    const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *g = dm + (size_t)f * H * W;
    int first = 0x7FFFFFFF, last = -1;
    for (int v = threadIdx.x; v < H; v += 256)
        if (row_any[(size_t)f * H + v]) { first = min(first, v); last = max(last, v); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        first = min(first, __shfl_xor(first, off));
        last = max(last, __shfl_xor(last, off));
    }
    if (lane == 0) s_first[wv] = first, s_last[wv] = last;
    __syncthreads();
    first = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));
    last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    if (last < 0) return;
    for (size_t i = threadIdx.x; i < (size_t)first * W; i += 256) g[i] = g[(size_t)first * W + i % W];
    for (size_t i = (size_t)(last + 1) * W + threadIdx.x; i < (size_t)H * W; i += 256) g[i] = g[(size_t)last * W + i % W];
}

// A fused aggregation launch that lost its lock step leaves void path volumes; the launches behind it cannot know and
// would turn them into plausible-looking disparities.  Queued right behind the post stage of such a call, this kernel
// overwrites the call's output with NaN when the device-side error word carries the call's launch serial: whatever
// consumes the result later on the stream (a gather, torch ops, a copy to the host) sees values no disparity map has.
__global__ void __launch_bounds__(256) void_if_lost_kernel(float *__restrict__ out, size_t n, const unsigned *__restrict__ err_dev,
                                                           unsigned serial)
{
    if (__hip_atomic_load(err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != serial) return;
    const float nan = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = nan;
}

int rsgm_launch_void_if_lost(vppx_ctx *ctx, float *out, size_t n, const unsigned *err_dev, unsigned serial)
{
    const unsigned nblk = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
    void_if_lost_kernel<<<nblk < 1 ? 1 : nblk, 256, 0, ctx->stream>>>(out, n, err_dev, serial);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int rsgm_launch_post(vppx_ctx *ctx, const RsgmGeom &g, const float *dl_pad, const float *dr_pad, int subpixel, float *out)
{
    const size_t n = (size_t)g.B * g.H * g.W, npf = (size_t)g.H * g.W;
    if (g.W > 8192) { vppx_set_error("frame width %d > 8192 is not supported by the post-processing kernels", g.W); return VPPX_E_UNSUPPORTED; }
    const int ntx = (g.W + SPK_TW - 1) / SPK_TW, nty = (g.H + SPK_TH - 1) / SPK_TH;
    const size_t ntiles = (size_t)ntx * nty * g.B;
    u8 *fd8, *row_any;
    int *label, *size, *roots, *nroots;
    int rc;
    if ((rc = ws_get(ctx, WS_FD8, n, &fd8))) return rc;
    if ((rc = ws_get(ctx, WS_LABEL, n, &label))) return rc;
    if ((rc = ws_get(ctx, WS_LCOUNT, n, &size))) return rc;
    if ((rc = ws_get(ctx, WS_SPK_ROOTS, ntiles * (SPK_TW * SPK_TH), &roots))) return rc;
    if ((rc = ws_get(ctx, WS_SPK_NROOTS, ntiles, &nroots))) return rc;
    if ((rc = ws_get(ctx, WS_ROW_ANY, (size_t)g.B * g.H, &row_any))) return rc;
    speckle_tile_kernel<<<dim3(ntx, nty, g.B), 256, 0, ctx->stream>>>(dl_pad, dr_pad, fd8, label, size, roots, nroots, g.H, g.W, g.Hp, g.Wp,
                                                                      g.pad_t, g.pad_l, 0, 10);
    VPPX_CHECK_LAUNCH();
    const int nby = (g.H - 1) / SPK_TH, nbx = (g.W - 1) / SPK_TW;
    const int nlinks = nby * g.W + nbx * g.H;
    if (nlinks > 0) {
        speckle_border_kernel<<<dim3((nlinks + 255) / 256, g.B), 256, 0, ctx->stream>>>(fd8, label, g.H, g.W, 0, 10);
        VPPX_CHECK_LAUNCH();
    }
    speckle_total_kernel<<<(unsigned)ntiles, 64, 0, ctx->stream>>>(label, size, roots, nroots, npf, ntx * nty);
    VPPX_CHECK_LAUNCH();
    const size_t lds = (size_t)g.W * 2 * sizeof(float);
    if (g.W <= 2048)
        speckle_apply_rows_kernel<8><<<dim3(g.H, g.B), 256, lds, ctx->stream>>>(fd8, label, size, dl_pad, out, row_any, g.H, g.W, g.Hp, g.Wp,
                                                                               g.pad_t, g.pad_l, 0, 200, subpixel);
    else
        speckle_apply_rows_kernel<32><<<dim3(g.H, g.B), 256, lds, ctx->stream>>>(fd8, label, size, dl_pad, out, row_any, g.H, g.W, g.Hp, g.Wp,
                                                                                g.pad_t, g.pad_l, 0, 200, subpixel);
    VPPX_CHECK_LAUNCH();
    interp_bg_edge_rows_kernel<<<g.B, 256, 0, ctx->stream>>>(out, row_any, g.H, g.W);
    VPPX_CHECK_LAUNCH();
    return 0;
}
