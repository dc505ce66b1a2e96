// vpp_kernels.hip -- virtual pattern projection on gfx950, hand-written HIP.
//
// Replaces the reference's sequential scan kernels for the hot path:
//   vpp_core/vpp_core_opt.pyx:53-131  virtual_projection_scan_rnd      (bit-exact target)
//   vpp_core/vpp_core_opt.pyx:133-341 virtual_projection_scan_max_dist (bit-exact target)
//   vpp_standalone.py:243-369 / :14-232 numba twins (extra gates only)
//   filter.py:246-292 occlusion_heuristic (producer of g_occ)
//
// The reference scan is strictly sequential (every hint blends a patch into BOTH images in
// place and later hints read what earlier ones wrote).  The parallel formulation used here
// (verified against the oracle): the total order of side effects is
// (hint rank in scan order, channel, yw, xw, slot); channels are independent; a write to an
// R pixel reads only that R pixel; a write to an L pixel reads only that L pixel plus, for
// occluded hints, R "as of that instant".  So every output pixel replays, in key order,
// exactly the ops that touch it:
//   1. compact_kernel : per row, hints in scan order + prefix sums of their rand() draws + a bitmap of the
//                       scan positions that hold a hint
//   2. rowscan_kernel : prefix over rows -> absolute position of every draw in the stream
//   3. rand_kernel    : glibc TYPE_3 rand() stream generated in parallel by polynomial
//                       jump-ahead (x_n = x_{n-3} + x_{n-31} mod 2^32 is linear)
//   4. apply_l_bits_kernel  : L pixels whose window holds a hint (found in the bitmaps, compacted through LDS);
//      apply_l_heavy_kernel : the ones with an occluded hint (they replay R sub-chains), a pair of lanes each
//   5. r_rows_kernel : per-R-pixel hint lists of a row in LDS, then the touched R pixels (compacted through LDS)
// Mixed float32/float64 blend arithmetic follows the C that Cython generates;
// the library is compiled with -ffp-contract=off and uses explicit _rn intrinsics.
#include "vppx_internal.h"

#include <math.h>
#include <string.h>
#include <mutex>
#include <stdlib.h>

#define LG 992 // draws per generator thread (32 passes over the 31-word ring)
#define RLCAP 8 // per-R-pixel hint list capacity (overflow falls back to the row scan)

struct HintRec { // 16 bytes, one per hint, row-compacted in scan order
    int x;
    float g;
    u32 base;  // first draw of this hint relative to the row's first draw
    u32 flags; // bit0 occluded, bits 8..15 patch radius n_k, bits 16..31 cnt (draws per channel, non-uniform)
};

__host__ __device__ inline int vpp_bits_words(int W) { return (W + 63) / 64 + 2; }

struct VppK {
    int B, H, W, C;
    int n, direction, uniform, discard, interp, use_dist, use_bil;
    float c, c_occ, dmin, dmax;
    const float2 *range; // use_distance_patch with per_frame_range: [B] {dmin, dmax} of every frame's own hints (hint_range_kernel), else null
    double inv_gamma;
    u8 *l;
    u8 *r;
    const u8 *r_src; // un-patterned right image for the L side's R sub-chains (= r unless the caller kept the original)
    const float *g;
    const u8 *occ;       // may be null
    const float *filled; // may be null
    HintRec *rec;        // [B][H][W]
    u32 *rng;            // [B][H][W] per hint: R-target column range lo | hi<<16 (int16 each)
    uint4 *dense;        // [B][H][W] at hint pixels: {idx in row, base in row, flags, bits of the hint value}
    unsigned long long *bits; // [B][H][vpp_bits_words(W)] which scan positions of a row hold a hint (one zero word before and after)
    int *rcnt;           // [B][H][W] number of hints that touch an R pixel
    u32 *rlist;          // [B][H][W][RLCAP] their (row << 16 | idx in row), unordered
    int *row_count;      // [B][H]
    u32 *row_draws;      // [B][H]
    u32 *row_base;       // [B][H]
    unsigned long long *frame_tot; // [B][2] {draws, hints}
    const u8 *rnd;       // [B][rnd_cap]
    size_t rnd_cap;
    u32 *lwork;          // L pixels deferred to the second pass: {frame, y << 16 | x} per entry
    int *lwork_cnt;      // their number
};

// ---------------------------------------------------------------------------------------
// per-hint helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int patch_radius(const VppK &k, int f, float gv)
{
    if (!k.use_dist) return k.n;
    float dmin = k.dmin, dmax = k.dmax;
    if (k.range) { // the frame's own range, as vpp() takes it per call (vpp_standalone.py:410-411)
        const float2 r = k.range[f];
        dmin = r.x; dmax = r.y;
        if (!(dmax > dmin)) return k.n; // one hint value only: the reference divides by zero there (include/vppx.h)
    }
    // vpp_standalone.py:7-11 with numba typing: float32 ratio, float64 pow, round half-to-even
    const float num = __fsub_rn(gv, dmin);
    const float den = __fsub_rn(dmax, dmin);
    const float ratio = __fdiv_rn(num, den);
    const double gw = pow((double)ratio, k.inv_gamma);
    const double ws = rint(__dadd_rn(__dmul_rn(gw, (double)(2 * k.n)), 1.0)); // patch_size-1 == 2n for odd wsize
    long long wsize = (long long)ws;
    long long a = wsize - 1;
    long long nn = a >= 0 ? a / 2 : -((-a + 1) / 2);
    if (nn < 0) nn = -1; // empty patch
    if (nn > k.n) nn = k.n;
    return (int)nn;
}

__device__ __forceinline__ bool gate_pass(const VppK &k, int f, float gv, int yy, int xx)
{
    // vpp_standalone.py:154,335: abs(g[y,x] - filled_g[y+yw,x+xw]) < 0.1 in float64
    const float fg = k.filled[((size_t)f * k.H + yy) * k.W + xx];
    return fabs((double)gv - (double)fg) < 0.1;
}

// number of rand() draws one channel of this hint consumes (non-uniform colour) and, for a
// patch pixel, its rank among the drawing pixels
__device__ __forceinline__ int hint_cnt(const VppK &k, int f, int y, int x, float gv, int nk)
{
    if (nk < 0) return 0;
    const int ywmin = max(-nk, -y), ywmax = min(nk, k.H - 1 - y);
    const int xwmin = max(-nk, -x), xwmax = min(nk, k.W - 1 - x);
    if (!k.use_bil) return (ywmax - ywmin + 1) * (xwmax - xwmin + 1);
    int c = 0;
    for (int yw = ywmin; yw <= ywmax; yw++)
        for (int xw = xwmin; xw <= xwmax; xw++) c += gate_pass(k, f, gv, y + yw, x + xw) ? 1 : 0;
    return c;
}
__device__ __forceinline__ int hint_idx(const VppK &k, int f, int y, int x, float gv, int nk, int yw0, int xw0)
{
    const int ywmin = max(-nk, -y);
    const int xwmin = max(-nk, -x), xwmax = min(nk, k.W - 1 - x);
    if (!k.use_bil) return (yw0 - ywmin) * (xwmax - xwmin + 1) + (xw0 - xwmin);
    int c = 0;
    for (int yw = ywmin; yw <= yw0; yw++)
        for (int xw = xwmin; xw <= xwmax; xw++) {
            if (yw == yw0 && xw >= xw0) break;
            c += gate_pass(k, f, gv, y + yw, x + xw) ? 1 : 0;
        }
    return c;
}

// ---------------------------------------------------------------------------------------
// 1. row compaction in scan order (vpp_core_opt.pyx:77-81,129; gt_reshape :352-371)
// ---------------------------------------------------------------------------------------
// (one wave per row, four rows per block: the hints of 64 columns are numbered by a ballot and a lane-mask popcount, their
// draws by one wave scan -- no block barriers; round 3: stage 0.16 -> 0.14 ms per 32 frames, two 50 MB copies included)
// min / max of a frame's positive hints (vpp_standalone.py:410-411: gt[gt > 0].min() / .max()), one block per frame.  Positive
// floats order like their bit patterns; NaN and values <= 0 are no hints (vpp_core_opt.pyx:81).  No positive value: {0, 0}, unused.
__global__ void __launch_bounds__(1024) hint_range_kernel(const float *__restrict__ g, size_t npx, float2 *__restrict__ range)
{
    __shared__ u32 s_lo, s_hi;
    if (threadIdx.x == 0) s_lo = 0xFFFFFFFFu, s_hi = 0u;
    __syncthreads();
    const float *gf = g + (size_t)blockIdx.x * npx;
    u32 lo = 0xFFFFFFFFu, hi = 0u;
    for (size_t i = threadIdx.x; i < npx; i += 1024) {
        const float v = gf[i];
        if (v > 0) {
            const u32 b = __float_as_uint(v);
            lo = min(lo, b);
            hi = max(hi, b);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (u32)__shfl_xor((int)lo, off));
        hi = max(hi, (u32)__shfl_xor((int)hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_lo, lo);
        atomicMax(&s_hi, hi);
    }
    __syncthreads();
    if (threadIdx.x == 0) range[blockIdx.x] = s_hi == 0u ? make_float2(0.f, 0.f) : make_float2(__uint_as_float(s_lo), __uint_as_float(s_hi));
}

#define CK_NG 16
template <bool DIST> // use_distance_patch: the radius of a patch is a float64 pow() of its hint (registers the common case does not pay for)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) compact_kernel(VppK k)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.x * 4 + wv, f = blockIdx.y;
    if (y >= k.H) return;
    const size_t rowoff = ((size_t)f * k.H + y) * k.W;
    u32 run_cnt = 0, run_drw = 0; // wave-uniform running totals of the row
    unsigned long long *brow = k.bits + ((size_t)f * k.H + y) * vpp_bits_words(k.W);
    if (lane == 0) brow[0] = 0ull, brow[vpp_bits_words(k.W) - 1] = 0ull;
    // (CK_NG chunks of 64 columns per round -- a whole row of up to 1024 columns: their hint values and mask bytes are
    // loaded together, unconditionally.  Next to the sum / WTA kernel of the previous part, whose one block per CU leaves
    // room for a wave or two per SIMD, a round trip through memory takes several microseconds: they are what this
    // kernel's time is made of there.)
    for (int pq = 0; pq < k.W; pq += 64 * CK_NG) {
    float gvq[CK_NG];
    u32 ocq[CK_NG / 4]; // the mask bytes, four to a register
#pragma unroll
    for (int u = 0; u < CK_NG / 4; u++) ocq[u] = 0;
#pragma unroll
    for (int u = 0; u < CK_NG; u++) {
        const int pc = min(pq + 64 * u + lane, k.W - 1);
        gvq[u] = k.g[rowoff + (k.direction ? pc : k.W - 1 - pc)];
    }
    if (k.occ) {
#pragma unroll
        for (int u = 0; u < CK_NG; u++) {
            const int pc = min(pq + 64 * u + lane, k.W - 1);
            ocq[u / 4] |= (u32)k.occ[rowoff + (k.direction ? pc : k.W - 1 - pc)] << (8 * (u % 4));
        }
    }
#pragma unroll 1 // (one copy of the body -- patch_radius alone is 170 float64 instructions --: the chunk's values are picked out of the registers)
    for (int u = 0; u < CK_NG; u++) {
        const int p0 = pq + 64 * u;
        if (p0 >= k.W) break; // uniform
        const int p = p0 + lane;
        const int x = k.direction ? p : k.W - 1 - p;
        float gsel = gvq[0];
        u32 osel = ocq[0];
#pragma unroll
        for (int j = 1; j < CK_NG; j++) gsel = u == j ? gvq[j] : gsel;
#pragma unroll
        for (int j = 1; j < CK_NG / 4; j++) osel = (u >> 2) == j ? ocq[j] : osel;
        osel = (osel >> (8 * (u & 3))) & 0xFFu;
        const float gv = p < k.W ? gsel : 0.f;
        const bool is = p < k.W && gv > 0; // NaN and <= 0 are skipped (vpp_core_opt.pyx:81)
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(is);
        if (lane == 0) brow[1 + (p0 >> 6)] = mk; // bit = scan position: the L side finds the hints of a pixel's window here
        if (mk == 0) continue; // uniform
        int nk = 0;
        u32 cnt = 0, drw = 0;
        if (is) {
            nk = DIST ? patch_radius(k, f, gv) : k.n;
            cnt = (u32)hint_cnt(k, f, y, x, gv, nk);
            drw = k.uniform ? (u32)k.C : (u32)k.C * cnt;
        }
        // inclusive wave scan of the draws
        u32 vd = drw;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 td = __shfl_up(vd, off);
            if (lane >= off) vd += td;
        }
        const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0u)); // hints in lower lanes
        if (is) {
            const u32 my_idx = run_cnt + below;
            const u32 my_base = run_drw + vd - drw;
            HintRec r;
            r.x = x;
            r.g = gv;
            r.base = my_base;
            const u32 occ = osel != 0 ? 1u : 0u;
            r.flags = occ | ((u32)(nk & 0xFF) << 8) | ((cnt & 0xFFFFu) << 16);
            k.rec[rowoff + my_idx] = r;
            k.dense[rowoff + x] = make_uint4(my_idx, my_base, r.flags, __float_as_uint(gv));
            // columns of R this hint can touch: [xd0-1-n_k, xd0+n_k] (a negative lo also means the
            // Python-style wraparound write/read of column W-1, SURVEY C-1/C-2)
            int lo = x - (int)floorf(gv) - 1 - nk, hi = x - (int)floorf(gv) + nk;
            lo = lo < -32768 ? -32768 : lo;
            hi = hi < -32768 ? -32768 : (hi > 32767 ? 32767 : hi);
            k.rng[rowoff + my_idx] = ((u32)lo & 0xFFFFu) | ((u32)hi << 16);
        }
        run_cnt += (u32)__popcll(mk);
        run_drw += (u32)__builtin_amdgcn_readlane((int)vd, 63);
    }
    }
    if (lane == 0) {
        k.row_count[(size_t)f * k.H + y] = (int)run_cnt;
        k.row_draws[(size_t)f * k.H + y] = run_drw;
    }
}

// 2. exclusive scan of the per-row draw counts (one block per frame)
__global__ void __launch_bounds__(256) rowscan_kernel(VppK k, long long *n_hints_out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && k.lwork_cnt) *k.lwork_cnt = 0; // work list of the L side's second pass (apply_l_heavy_kernel)
    __shared__ unsigned long long s_d[4], s_c[4];
    __shared__ unsigned long long s_run_d, s_run_c;
    const int f = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_run_d = 0; s_run_c = 0; }
    __syncthreads();
    for (int y0 = 0; y0 < k.H; y0 += 256) {
        const int y = y0 + threadIdx.x;
        unsigned long long d = 0, c = 0;
        if (y < k.H) {
            d = k.row_draws[(size_t)f * k.H + y];
            c = (unsigned long long)k.row_count[(size_t)f * k.H + y];
        }
        unsigned long long vd = d, vc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long td = __shfl_up(vd, off), tc = __shfl_up(vc, off);
            if (lane >= off) { vd += td; vc += tc; }
        }
        if (lane == 63) { s_d[wv] = vd; s_c[wv] = vc; }
        __syncthreads();
        unsigned long long pd = s_run_d, pc = s_run_c;
        for (int w = 0; w < wv; w++) { pd += s_d[w]; pc += s_c[w]; }
        if (y < k.H) k.row_base[(size_t)f * k.H + y] = (u32)(pd + vd - d);
        __syncthreads();
        if (threadIdx.x == 255) { s_run_d = pd + vd; s_run_c = pc + vc; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        k.frame_tot[2 * f + 0] = s_run_d;
        k.frame_tot[2 * f + 1] = s_run_c;
        if (n_hints_out) n_hints_out[f] = (long long)s_run_c;
    }
}

// ---------------------------------------------------------------------------------------
// 3. glibc rand() stream (srand: vpp_core_opt.pyx:33-35; rand()%256: :93,102).
//    y_n = x_{n+3}: y_n = y_{n-3} + y_{n-31} (mod 2^32, n >= 31), basis v_i = y_i (i<31) from
//    srand's Schrage sequence; output o_k = y_{k+341} >> 1.  y_N = <coef(z^N mod P), v> with
//    P(z) = z^31 - z^28 - 1.  Generator thread (frame f, block b) covers LG draws starting at
//    k0 = rand_offset + b*LG: p = z^(rand_offset+310) * z^(b*LG) mod P gives its 31-word ring,
//    then LG sequential steps.
// ---------------------------------------------------------------------------------------
struct Poly31 { u32 c[31]; };

__host__ __device__ inline void poly_mulmod(const u32 *a, const u32 *b, u32 *out)
{
    u32 t[61];
    for (int i = 0; i < 61; i++) t[i] = 0;
    for (int i = 0; i < 31; i++)
        for (int j = 0; j < 31; j++) t[i + j] += a[i] * b[j];
    for (int kk = 60; kk >= 31; kk--) { // z^k = z^(k-3) + z^(k-31)
        t[kk - 3] += t[kk];
        t[kk - 31] += t[kk];
    }
    for (int i = 0; i < 31; i++) out[i] = t[i];
}

__device__ __forceinline__ void srand_basis(u32 seed, u32 (&v)[31])
{
    // glibc __srandom_r (TYPE_3): r[0]=seed (0 -> 1); r[i] = 16807*r[i-1] % 2147483647 (Schrage, int32)
    int word = (int)(seed == 0 ? 1u : seed);
    int r[31];
    r[0] = word;
#pragma unroll
    for (int i = 1; i < 31; i++) {
        const long long hi = word / 127773, lo = word % 127773;
        long long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int)w;
        r[i] = word;
    }
    // v_i = y_i = x_{i+3}; x_31..33 = x_0..2
#pragma unroll
    for (int i = 0; i < 31; i++) v[i] = (u32)r[(i + 3) % 31];
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) rand_kernel(u8 *__restrict__ out, size_t cap, const unsigned long long *__restrict__ frame_tot,
                                                  const u32 *__restrict__ seeds, u32 seed0, Poly31 qoff,
                                                  const u32 *__restrict__ tab /*[nblk][31]*/, int nblk)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int f = blockIdx.y;
    if (b >= nblk) return;
    const unsigned long long tot = frame_tot ? frame_tot[2 * f] : (unsigned long long)cap;
    if ((unsigned long long)b * LG >= tot) return;
    const u32 seed = seeds ? seeds[f] : seed0 + (u32)f;
    u32 v[31], p[31], s[31];
    srand_basis(seed, v);
    {
        u32 tb[31];
#pragma unroll
        for (int i = 0; i < 31; i++) tb[i] = tab[(size_t)b * 31 + i];
        // p = qoff * tb mod P
        u32 t[61];
#pragma unroll
        for (int i = 0; i < 61; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 31; i++)
#pragma unroll
            for (int j = 0; j < 31; j++) t[i + j] += qoff.c[i] * tb[j];
#pragma unroll
        for (int kk = 60; kk >= 31; kk--) {
            t[kk - 3] += t[kk];
            t[kk - 31] += t[kk];
        }
#pragma unroll
        for (int i = 0; i < 31; i++) p[i] = t[i];
    }
#pragma unroll
    for (int i = 0; i < 31; i++) {
        u32 acc = 0;
#pragma unroll
        for (int j = 0; j < 31; j++) acc += p[j] * v[j];
        s[i] = acc;
        // p = p * z mod P
        const u32 top = p[30];
#pragma unroll
        for (int j = 30; j >= 1; j--) p[j] = p[j - 1];
        p[0] = top;
        p[28] += top;
    }
    // ring: slot i holds y_{n-31}, y_{n-3} is slot (i+28)%31
    u32 *o = (u32 *)(out + (size_t)f * cap + (size_t)b * LG);
    for (int it = 0; it < LG / 124; it++) {
        u32 wv = 0;
        int nb = 0, wi = 0;
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
#pragma unroll
            for (int i = 0; i < 31; i++) {
                s[i] += s[(i + 28) % 31];
                wv |= ((s[i] >> 1) & 0xFFu) << (8 * nb);
                if (++nb == 4) {
                    o[it * 31 + wi] = wv;
                    wi++;
                    wv = 0;
                    nb = 0;
                }
            }
        }
    }
}

// full 31-bit outputs (vppx_rand_stream): same generator, int32 output
__global__ void __launch_bounds__(64) rand_full_kernel(int *__restrict__ out, long long n, u32 seed, Poly31 qoff,
                                                       const u32 *__restrict__ tab, int nblk)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblk || (long long)b * LG >= n) return;
    u32 v[31], p[31], s[31], tb[31];
    srand_basis(seed, v);
    for (int i = 0; i < 31; i++) tb[i] = tab[(size_t)b * 31 + i];
    poly_mulmod(qoff.c, tb, p);
    for (int i = 0; i < 31; i++) {
        u32 acc = 0;
        for (int j = 0; j < 31; j++) acc += p[j] * v[j];
        s[i] = acc;
        const u32 top = p[30];
        for (int j = 30; j >= 1; j--) p[j] = p[j - 1];
        p[0] = top;
        p[28] += top;
    }
    int ri = 0;
    for (int t = 0; t < LG; t++) {
        s[ri] += s[(ri + 28) % 31];
        const long long idx = (long long)b * LG + t;
        if (idx < n) out[idx] = (int)(s[ri] >> 1);
        ri = ri == 30 ? 0 : ri + 1;
    }
}

// host-side tables: z^(b*LG) mod P (device resident, grown on demand) and z^N mod P
static std::mutex g_tab_mutex;
static std::vector<u32> g_tab_host; // [nblk][31]
static u32 *g_tab_dev[VPPX_MAX_DEVICES] = {}; // one copy per device a context lives on
static int g_tab_dev_n[VPPX_MAX_DEVICES] = {};

static void poly_pow_z(unsigned long long N, u32 *out)
{
    u32 res[31] = {0}, base[31] = {0}, tmp[31];
    res[0] = 1;
    base[1] = 1; // z
    while (N) {
        if (N & 1) { poly_mulmod(res, base, tmp); memcpy(res, tmp, sizeof(tmp)); }
        N >>= 1;
        if (N) { poly_mulmod(base, base, tmp); memcpy(base, tmp, sizeof(tmp)); }
    }
    memcpy(out, res, 31 * sizeof(u32));
}

static int ensure_rand_table(vppx_ctx *ctx, int nblk, const u32 **tab_out)
{
    std::lock_guard<std::mutex> lk(g_tab_mutex);
    const int have = (int)(g_tab_host.size() / 31);
    if (have < nblk) {
        u32 step[31];
        poly_pow_z(LG, step);
        g_tab_host.resize((size_t)nblk * 31);
        if (have == 0) {
            memset(&g_tab_host[0], 0, 31 * sizeof(u32));
            g_tab_host[0] = 1;
        }
        for (int b = (have == 0 ? 1 : have); b < nblk; b++)
            poly_mulmod(&g_tab_host[(size_t)(b - 1) * 31], step, &g_tab_host[(size_t)b * 31]);
    }
    const int dv = ctx->device & (VPPX_MAX_DEVICES - 1);
    if (g_tab_dev_n[dv] < nblk) {
        // grow: the old (shorter, prefix-identical) table is left alive on purpose -- kernels enqueued by other
        // contexts / streams and captured graphs may still reference it (a few MB per growth at most)
        g_tab_dev[dv] = nullptr;
        g_tab_dev_n[dv] = 0;
        VPPX_HIP(hipMalloc((void **)&g_tab_dev[dv], (size_t)nblk * 31 * sizeof(u32)));
        VPPX_HIP(hipMemcpy(g_tab_dev[dv], g_tab_host.data(), (size_t)nblk * 31 * sizeof(u32), hipMemcpyHostToDevice));
        g_tab_dev_n[dv] = nblk;
    }
    *tab_out = g_tab_dev[dv];
    return 0;
}

// ---------------------------------------------------------------------------------------
// blend arithmetic (SURVEY A.2 / generated C of vpp_core_opt.pyx:107-124)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u8 to_u8(double v) { return (u8)(int)v; } // (uint8_t)(double): truncation

// (u8)( V*cc + P*(1.0-cc) )                                   pyx:107,113,121,124
__device__ __forceinline__ u8 blend1(u8 rv, float cc, u8 P)
{
    const double a = (double)__fmul_rn((float)rv, cc);
    const double b = __dmul_rn((double)P, __dsub_rn(1.0, (double)cc));
    return to_u8(__dadd_rn(a, b));
}
// (u8)( (V*cc + R*(1.0-cc)) * (1.0-beta) + R*beta )            pyx:109,116  (R*beta is float32)
__device__ __forceinline__ u8 blend_r0(u8 rv, float cc, u8 R, float beta)
{
    const double a = (double)__fmul_rn((float)rv, cc);
    const double b = __dmul_rn((double)R, __dsub_rn(1.0, (double)cc));
    const double t = __dmul_rn(__dadd_rn(a, b), __dsub_rn(1.0, (double)beta));
    const double u = (double)__fmul_rn((float)R, beta);
    return to_u8(__dadd_rn(t, u));
}
// (u8)( (V*cc + R*(1.0-cc)) * beta + R*(1.0-beta) )            pyx:111,118
__device__ __forceinline__ u8 blend_r1(u8 rv, float cc, u8 R, float beta)
{
    const double a = (double)__fmul_rn((float)rv, cc);
    const double b = __dmul_rn((double)R, __dsub_rn(1.0, (double)cc));
    const double t = __dmul_rn(__dadd_rn(a, b), (double)beta);
    const double u = __dmul_rn((double)R, __dsub_rn(1.0, (double)beta));
    return to_u8(__dadd_rn(t, u));
}
// (u8)( (R0*(1.0-beta) + R1*beta) * c + L*(1.0-c) )            pyx:119  (R1*beta float32)
__device__ __forceinline__ u8 blend_l_occ(u8 R0, u8 R1, float beta, float c, u8 L)
{
    const double a = __dmul_rn((double)R0, __dsub_rn(1.0, (double)beta));
    const double b = (double)__fmul_rn((float)R1, beta);
    const double t = __dmul_rn(__dadd_rn(a, b), (double)c);
    const double u = __dmul_rn((double)L, __dsub_rn(1.0, (double)c));
    return to_u8(__dadd_rn(t, u));
}
// (u8)( Rd*c + L*(1.0-c) )                                     pyx:122  (Rd*c float32)
__device__ __forceinline__ u8 blend_l_occ_ni(u8 Rd, float c, u8 L)
{
    const double a = (double)__fmul_rn((float)Rd, c);
    const double u = __dmul_rn((double)L, __dsub_rn(1.0, (double)c));
    return to_u8(__dadd_rn(a, u));
}

// (Loads inside `for (j < k.C)` loops or behind `if (j >= k.C) break` are serialised by hipcc: one branch + load + wait per
// channel.  The replay kernels are chains of dependent round trips through memory, so the channels' draws and the pixel's
// bytes are loaded UNCONDITIONALLY -- a channel that does not exist re-reads channel 0 -- and used afterwards.)
struct HintGeo {
    int x, d0, d1, d, nk, occ, cnt;
    float g, beta;
    u32 base;
};
__device__ __forceinline__ void decode_hint(int x, float gv, u32 base, u32 flags, HintGeo &h)
{
    h.x = x;
    h.g = gv;
    h.d = (int)roundf(gv);  // round half away from zero (libc round, pyx:82)
    h.d0 = (int)floorf(gv); // pyx:83
    h.d1 = (int)ceilf(gv);  // pyx:84
    h.beta = __fsub_rn(gv, (float)h.d0);
    h.base = base;
    h.occ = (int)(flags & 1u);
    h.nk = (int)(signed char)((flags >> 8) & 0xFFu);
    h.cnt = (int)(flags >> 16);
}

__device__ __forceinline__ u8 draw(const VppK &k, const u8 *rnd, const HintGeo &h, int j, int idx)
{
    return k.uniform ? rnd[h.base + (u32)j] : rnd[h.base + (u32)j * (u32)h.cnt + (u32)idx];
}
__device__ __forceinline__ void draw4(const VppK &k, const u8 *rnd, const HintGeo &h, int idx, u8 (&dv)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) dv[j] = draw(k, rnd, h, j < k.C ? j : 0, idx);
}
__device__ __forceinline__ void load_px4(const u8 *px, int C, u8 (&v)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = px[j < C ? j : 0];
}

// All ops of one hint of row yy (scan order) that touch R pixel (yp, q), for every channel, with xw <= xw_lim.
// R[] carries the pixel through the chain.  Three phases, so that a caller can put the draws of several hints in flight
// together: r_prep decides which of the (at most two) patch columns act on the pixel and where their draws are, r_draws
// loads them -- unconditionally: a column that does not act re-reads the first byte of the stream --, r_blend applies them.
struct RCand {
    bool on[2], hit0[2], hit1[2];
    const u8 *dp[2];
    u32 stride[2];
    float cc, beta;
};
__device__ __forceinline__ void r_prep(const VppK &k, int f, int yp, int yy, const HintRec rec, int q, const u8 *rnd, int xw_lim, RCand &c)
{
    const int W = k.W;
    HintGeo h;
    decode_hint(rec.x, rec.g, rec.base, rec.flags, h);
    const int yw = yp - yy;
    const bool row_ok = !(yw < -h.nk || yw > h.nk) && !(h.occ && k.discard);
    c.cc = h.occ ? k.c_occ : k.c;
    c.beta = h.beta;
    const int xd0 = h.x - h.d0, xd1 = h.x - h.d1, xd = h.x - h.d;
    const int xw_hi = min(h.nk, xw_lim);
    // The reference walks xw = -n_k .. n_k; only the (at most two) offsets whose target is column q act on this pixel:
    // xd0 + xw == q, xd1 + xw == q (one apart at most, pyx:110,117), or without interpolation xd + xw == q / q - W (the
    // wraparound write, pyx:113,121).  They are visited in ascending order, as the loop would.
    const int xa = k.interp ? q - xd0 : q - W - xd;
    const int xb = k.interp ? q - xd1 : q - xd;
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        const int xw = s2 == 0 ? xa : xb;
        const int xx = h.x + xw;
        bool on = row_ok && !(s2 == 1 && xb == xa) && !(xw < -h.nk || xw > xw_hi) && !(xx < 0 || xx > W - 1); // pyx:99
        if (k.use_bil && on) on = gate_pass(k, f, h.g, yp, xx);
        on = on && (0 <= xd0 + xw && xd0 + xw <= W - 1);    // pyx:104 (else-branch touches L only)
        bool hit0, hit1 = false;
        if (k.interp) {
            hit0 = (xd0 + xw == q);
            hit1 = (xd1 + xw >= 0) && (xd1 + xw == q);       // pyx:110,117
        } else {
            const int tq = xd + xw;
            hit0 = ((tq < 0 ? tq + W : tq) == q);            // pyx:113,121 (wraparound)
        }
        on = on && (hit0 || hit1);
        int idx = 0;
        if (!k.uniform && (on || !k.use_bil)) idx = hint_idx(k, f, yy, h.x, h.g, h.nk, yw, xw); // (loads only with the bilateral gate)
        c.on[s2] = on;
        c.hit0[s2] = hit0;
        c.hit1[s2] = hit1;
        c.dp[s2] = on ? rnd + h.base + (u32)idx : k.rnd;
        c.stride[s2] = on && !k.uniform ? (u32)h.cnt : (on ? 1u : 0u);
    }
}
__device__ __forceinline__ void r_draws(const VppK &k, const RCand &c, u8 (&dv)[2][4])
{
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 4; j++) dv[s2][j] = c.dp[s2][(u32)(j < k.C ? j : 0) * c.stride[s2]];
}
__device__ __forceinline__ void r_blend(const VppK &k, const RCand &c, const u8 (&dv)[2][4], u8 (&R)[4])
{
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        if (!c.on[s2]) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j >= k.C) break;
            const u8 rv = dv[s2][j];
            if (k.interp) {
                if (c.hit0[s2]) R[j] = blend_r0(rv, c.cc, R[j], c.beta);
                if (c.hit1[s2]) R[j] = blend_r1(rv, c.cc, R[j], c.beta);
            } else {
                R[j] = blend1(rv, c.cc, R[j]);
            }
        }
    }
}
__device__ __forceinline__ void r_apply_rec(const VppK &k, int f, int yp, int yy, const HintRec rec, int q, u8 (&R)[4], const u8 *rnd,
                                            int xw_lim)
{
    RCand c;
    u8 dv[2][4];
    r_prep(k, f, yp, yy, rec, q, rnd, xw_lim, c);
    r_draws(k, c, dv);
    r_blend(k, c, dv, R);
}
// two hints, the second one only if `two`: all four sets of draws in flight together
__device__ __forceinline__ void r_apply_rec2(const VppK &k, int f, int yp, int q, u8 (&R)[4], int yy0, const HintRec rec0, const u8 *rnd0,
                                             int lim0, bool two, int yy1, const HintRec rec1, const u8 *rnd1, int lim1)
{
    RCand c0, c1;
    u8 dv0[2][4], dv1[2][4];
    r_prep(k, f, yp, yy0, rec0, q, rnd0, lim0, c0);
    r_prep(k, f, yp, yy1, rec1, q, rnd1, lim1, c1);
    c1.on[0] = c1.on[0] && two;
    c1.on[1] = c1.on[1] && two;
    r_draws(k, c0, dv0);
    r_draws(k, c1, dv1);
    r_blend(k, c0, dv0, R);
    r_blend(k, c1, dv1, R);
}

__device__ __forceinline__ void r_apply_hint(const VppK &k, int f, int yp, int yy, int i, int q, u8 (&R)[4], const u8 *rnd,
                                             int xw_lim)
{
    r_apply_rec(k, f, yp, yy, k.rec[((size_t)f * k.H + yy) * k.W + i], q, R, rnd, xw_lim);
}

__device__ __forceinline__ bool rng_hit(u32 rg, int q, int W)
{
    // targets lie in [lo, hi] (or wrap -1 -> W-1)
    const int lo = (int)(short)(rg & 0xFFFFu), hi = (int)(short)(rg >> 16);
    return (q >= lo && q <= hi) || (q == W - 1 && lo < 0);
}

// Replay the chain of R pixel (yp, q) for all channels up to and including the op with key
// (lim_y, lim_i, lim_xw).  R[] enters with the original pixel.  (Used by the L kernel for the
// occluded-hint branch, which needs R "as of that instant".)
__device__ void r_chain(const VppK &k, int f, int yp, int q, u8 (&R)[4], int lim_y, int lim_i, int lim_xw)
{
    const int W = k.W, H = k.H;
    // the pixel's hint list (r_rows_kernel has written it out before every caller) names exactly the hints that can touch it:
    // replay those up to the limit key instead of scanning the hint rows (a few entries instead of ~3 rows of hints)
    if (k.rcnt != nullptr && lim_y < H) {
        const size_t pidx = ((size_t)f * H + yp) * W + q;
        const int n = k.rcnt[pidx];
        if (n <= RLCAP) {
            if (n == 0) return;
            static_assert(RLCAP == 8, "list record = 2 x uint4");
            u32 ids[RLCAP];
            const uint4 a = ((const uint4 *)k.rlist)[pidx * 2], b = ((const uint4 *)k.rlist)[pidx * 2 + 1];
            ids[0] = a.x; ids[1] = a.y; ids[2] = a.z; ids[3] = a.w;
            ids[4] = b.x; ids[5] = b.y; ids[6] = b.z; ids[7] = b.w;
#pragma unroll
            for (int i = 0; i < RLCAP; i++) ids[i] = i < n ? ids[i] : 0xFFFFFFFFu;
#define CS(a, b) { const u32 lo_ = min(ids[a], ids[b]), hi_ = max(ids[a], ids[b]); ids[a] = lo_; ids[b] = hi_; }
            if (n == 2) {
                CS(0, 1)
            } else if (n > 2) { // most lists hold one or two hints
                CS(0, 1) CS(2, 3) CS(4, 5) CS(6, 7) CS(0, 2) CS(1, 3) CS(4, 6) CS(5, 7) CS(1, 2) CS(5, 6)
                CS(0, 4) CS(3, 7) CS(1, 5) CS(2, 6) CS(1, 4) CS(3, 6) CS(2, 4) CS(3, 5) CS(3, 4)
            }
#undef CS
            const u32 lim = ((u32)lim_y << 16) | (u32)min(lim_i, 0xFFFF);
            const u8 *rnd_f = k.rnd + (size_t)f * k.rnd_cap;
            // the records and row bases of the first two entries (most lists hold one or two) are fetched together
            const int yy0 = (int)(ids[0] >> 16), yy1 = n > 1 ? (int)(ids[1] >> 16) : yy0;
            const int hi0 = (int)(ids[0] & 0xFFFFu), hi1 = n > 1 ? (int)(ids[1] & 0xFFFFu) : hi0;
            const HintRec rec0 = k.rec[((size_t)f * H + yy0) * W + hi0], rec1 = k.rec[((size_t)f * H + yy1) * W + hi1];
            const u32 rb0 = k.row_base[(size_t)f * H + yy0], rb1 = k.row_base[(size_t)f * H + yy1];
            if (ids[0] > lim) return;
            const bool two = n > 1 && ids[1] <= lim;
            r_apply_rec2(k, f, yp, q, R, yy0, rec0, rnd_f + rb0, ids[0] == lim ? lim_xw : 0x7FFFFFFF, two, yy1, rec1, rnd_f + rb1,
                         ids[1] == lim ? lim_xw : 0x7FFFFFFF);
            if (!two) return;
            for (int i = 2; i < n; i++) { // (a loop, not unrolled copies of the replay: the lists are short)
                const u32 id = i == 2 ? ids[2] : i == 3 ? ids[3] : i == 4 ? ids[4] : i == 5 ? ids[5] : i == 6 ? ids[6] : ids[7];
                if (id > lim) break;
                const int yy = (int)(id >> 16), hi = (int)(id & 0xFFFFu);
                r_apply_hint(k, f, yp, yy, hi, q, R, rnd_f + k.row_base[(size_t)f * H + yy], id == lim ? lim_xw : 0x7FFFFFFF);
            }
            return;
        }
    }
    const int ylo = max(0, yp - k.n), yhi = min(min(H - 1, yp + k.n), lim_y);
    const u8 *rnd_f = k.rnd + (size_t)f * k.rnd_cap;
    for (int yy = ylo; yy <= yhi; yy++) {
        const size_t rowoff = ((size_t)f * H + yy) * W;
        int cntrow = k.row_count[(size_t)f * H + yy];
        if (yy == lim_y && lim_i + 1 < cntrow) cntrow = lim_i + 1;
        const u8 *rnd = rnd_f + k.row_base[(size_t)f * H + yy];
        for (int i = 0; i < cntrow; i++) {
            if (!rng_hit(k.rng[rowoff + i], q, W)) continue;
            const bool at_limit = (yy == lim_y && i == lim_i);
            r_apply_hint(k, f, yp, yy, i, q, R, rnd, at_limit ? lim_xw : 0x7FFFFFFF);
        }
    }
}

// ---------------------------------------------------------------------------------------
// 5. R pixels: one thread per touched pixel replays the hints of its list in scan order; untouched
// pixels are not even read.  A list that overflowed (> RLCAP hints on one pixel: very dense
// hints) falls back to scanning the hint rows.
// ---------------------------------------------------------------------------------------
// replay of one R pixel from its (complete, n <= RLCAP) list; ids[] holds the list (entries >= n ignored)
__device__ __forceinline__ void r_replay_list(const VppK &k, int f, int yp, int q, size_t pidx, int n, u32 (&ids)[RLCAP])
{
    const int H = k.H;
    u8 *px = k.r + pidx * k.C;
    u8 R[4];
    load_px4(px, k.C, R);
    const u8 *rnd_f = k.rnd + (size_t)f * k.rnd_cap;
#pragma unroll
    for (int i = 0; i < RLCAP; i++) ids[i] = i < n ? ids[i] : 0xFFFFFFFFu;
    // scan order = ascending (row, idx in row): sorting network for 8 keys (most lists hold one or two hints)
#define CS(a, b) { const u32 lo_ = min(ids[a], ids[b]), hi_ = max(ids[a], ids[b]); ids[a] = lo_; ids[b] = hi_; }
    if (n == 2) {
        CS(0, 1)
    } else if (n > 2) {
        CS(0, 1) CS(2, 3) CS(4, 5) CS(6, 7) CS(0, 2) CS(1, 3) CS(4, 6) CS(5, 7) CS(1, 2) CS(5, 6)
        CS(0, 4) CS(3, 7) CS(1, 5) CS(2, 6) CS(1, 4) CS(3, 6) CS(2, 4) CS(3, 5) CS(3, 4)
    }
#undef CS
    // (a loop, not eight unrolled copies of the replay: eight copies of the float64 blends of four channels made this kernel
    // 75 KB of code -- more than the instruction cache two CUs share.)  The records and row bases of the first two entries --
    // most lists hold one or two -- are fetched together, ahead of the loop.
    const int yy0 = (int)(ids[0] >> 16), yy1 = n > 1 ? (int)(ids[1] >> 16) : yy0;
    const int hi0 = (int)(ids[0] & 0xFFFFu), hi1 = n > 1 ? (int)(ids[1] & 0xFFFFu) : hi0;
    const HintRec rec0 = k.rec[((size_t)f * H + yy0) * k.W + hi0], rec1 = k.rec[((size_t)f * H + yy1) * k.W + hi1];
    const u32 rb0 = k.row_base[(size_t)f * H + yy0], rb1 = k.row_base[(size_t)f * H + yy1];
    // (callers hand a wave pixels with lists of the same length where they can: a wave of one-entry lists skips the second
    // record's half of the work)
    if (__builtin_amdgcn_ballot_w64(n > 1) == 0) r_apply_rec(k, f, yp, yy0, rec0, q, R, rnd_f + rb0, 0x7FFFFFFF);
    else r_apply_rec2(k, f, yp, q, R, yy0, rec0, rnd_f + rb0, 0x7FFFFFFF, n > 1, yy1, rec1, rnd_f + rb1, 0x7FFFFFFF);
#pragma unroll 1
    for (int i = 2; i < n; i++) {
        const u32 id = i == 2 ? ids[2] : i == 3 ? ids[3] : i == 4 ? ids[4] : i == 5 ? ids[5] : i == 6 ? ids[6] : ids[7];
        const int yy = (int)(id >> 16), hi = (int)(id & 0xFFFFu);
        r_apply_hint(k, f, yp, yy, hi, q, R, rnd_f + k.row_base[(size_t)f * H + yy], 0x7FFFFFFF);
    }
    for (int j = 0; j < k.C; j++) px[j] = R[j];
}
// List build and replay of (a part of) one R row in ONE block, the lists in LDS.  The hints that can touch row yp are those
// of rows yp - n .. yp + n: their (hint, target column) items append to the row's lists with LDS atomics -- as global
// atomics, one L2 operation per item, they were what bounded the list kernel of rounds 1-3 (14 M per 16 frames) --, the
// touched pixels are compacted and replayed from LDS; no count image to clear, to fill and to read back, no lists through
// memory.
// Only the second L pass (apply_l_heavy_kernel, apply_l_wide_kernel: r_chain) reads lists of R pixels, and only of pixels an
// OCCLUDED hint writes (the two pixels it blends into, pyx:114-122) or of column W-1 (the unguarded read of pyx:119): with
// `lists` those go to memory, and the count of every pixel of the row.
// IDT: a list entry is (row offset << IB) | index in row: 16 bits when W <= 2048 (IB = 11), else 32 (IB = 16).
// mode bit 0: replay, bit 1: write the lists out.
template <typename IDT, int IB, int NT>
__global__ void __launch_bounds__(NT) r_rows_kernel(VppK k, int mode, int SW)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];
    __shared__ int s_off[33];
    __shared__ int s_total, s_n2, s_n3;
    const int W = k.W, H = k.H, n = k.n;
    // a block owns the columns [q0, q0 + SW) of row yp (SW = W, or a part of a row too wide for one block's LDS)
    const int nseg = (W + SW - 1) / SW;
    const int yp = blockIdx.x / nseg, q0 = (blockIdx.x % nseg) * SW, f = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63;
    u32 *s_cnt = s_dyn;                                  // [SW] low half: hints on the pixel, high half: occluded ones among them
    IDT *s_list = (IDT *)(s_dyn + SW);                   // [SW][RLCAP]
    // touched columns by list length: one entry from the front of [0, SW), three and more from its back, two in [SW, 2 SW) --
    // replayed in that order, so that nearly every wave holds lists of one length
    unsigned short *s_touch = (unsigned short *)(s_list + (size_t)SW * RLCAP);
    const int span = 2 * n + 1, tw = 2 * n + 2;          // source rows, columns of a hint's target range
    for (int q = t; q < SW; q += NT) s_cnt[q] = 0;
    if (t == 0) s_total = 0, s_n2 = 0, s_n3 = 0;
    if (t < 64) { // items per source row and their prefix (span <= 31)
        const int yy = yp - n + t;
        int c = (t < span && yy >= 0 && yy <= H - 1) ? k.row_count[(size_t)f * H + yy] * tw : 0;
        int inc = c;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int v = __shfl_up(inc, off);
            if (lane >= off) inc += v;
        }
        if (t <= 32) s_off[t] = inc - c; // exclusive; s_off[span] = total
    }
    __syncthreads();
    const int total_items = s_off[span];
    const u32 tw_magic = 0xFFFFFFFFu / (u32)tw + 1u; // loc / tw == umulhi(loc, magic) for loc * tw < 2^32 (loc < 2^16 * 32)
    for (int t0 = t; t0 < total_items; t0 += NT * 4) {
        int rr[4], ii[4], tqv[4];
        u32 rgs[4], fl[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int it = min(t0 + NT * u, total_items - 1);
            int r = 0;
            for (int j = 1; j < span; j++) r += s_off[j] <= it ? 1 : 0; // (s_off is non-decreasing)
            const int loc = it - s_off[r];
            rr[u] = r;
            ii[u] = (int)__umulhi((u32)loc, tw_magic);
            tqv[u] = loc - ii[u] * tw;
            const size_t ro = ((size_t)f * H + (yp - n + r)) * W + ii[u];
            rgs[u] = k.rng[ro];
            fl[u] = k.rec[ro].flags;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (t0 + NT * u >= total_items) continue;
            const int lo = (int)(short)(rgs[u] & 0xFFFFu), hi = (int)(short)(rgs[u] >> 16);
            const u32 add = 1u + ((fl[u] & 1u) << 16);
            const IDT id = (IDT)(((u32)rr[u] << IB) | (u32)ii[u]);
            const int q = lo + tqv[u];
            if (q >= 0 && q <= hi && q <= W - 1 && q >= q0 && q < q0 + SW) {
                const u32 slot = atomicAdd(&s_cnt[q - q0], add) & 0xFFFFu;
                if (slot < RLCAP) s_list[(size_t)(q - q0) * RLCAP + slot] = id;
            }
            // Python-style wraparound target, column W-1 (SURVEY C-1/C-2): only the un-interpolated write
            // r[.., xd+xw] with xd = xd0-1 and xd0+xw == 0 can index -1 (pyx:113,121), i.e. lo < 0 <= hi
            if (tqv[u] == 0 && !k.interp && lo < 0 && hi >= 0 && !(W - 1 >= max(lo, 0) && W - 1 <= hi) && W - 1 < q0 + SW) {
                const u32 slot = atomicAdd(&s_cnt[W - 1 - q0], add) & 0xFFFFu;
                if (slot < RLCAP) s_list[(size_t)(W - 1 - q0) * RLCAP + slot] = id;
            }
        }
    }
    __syncthreads();
    const size_t prow = ((size_t)f * H + yp) * W;
    for (int qb = t & ~63; qb < SW; qb += NT) { // (whole waves; q is the column inside the block's part of the row)
        const int q = qb + lane;
        const bool qin = q < SW && q0 + q < W;
        const u32 c = qin ? s_cnt[q] : 0u;
        if ((mode & 2) && qin) k.rcnt[prow + q0 + q] = (int)(c & 0xFFFFu);
        const u32 cn = c & 0xFFFFu;
#pragma unroll
        for (int cls = 0; cls < 3; cls++) {
            const bool hit = cls == 0 ? cn == 1 : (cls == 1 ? cn == 2 : cn > 2);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            if (bal) {
                const int leader = __builtin_ctzll(bal);
                int base = 0;
                if (lane == leader) base = atomicAdd(cls == 0 ? &s_total : (cls == 1 ? &s_n2 : &s_n3), (int)__popcll(bal));
                base = __builtin_amdgcn_readlane(base, leader);
                const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                if (hit) s_touch[cls == 0 ? slot : (cls == 1 ? SW + slot : SW - 1 - slot)] = (unsigned short)q;
            }
        }
    }
    __syncthreads();
    const int n1 = s_total, n2 = s_n2, total = n1 + n2 + s_n3;
    for (int e = t; e < total; e += NT) {
        const int ql = s_touch[e < n1 ? e : (e < n1 + n2 ? SW + (e - n1) : SW - 1 - (e - n1 - n2))];
        const u32 c = s_cnt[ql];
        const int cnt = (int)(c & 0xFFFFu);
        const int q = q0 + ql;
        const size_t pidx = prow + q;
        u32 ids[RLCAP];
#pragma unroll
        for (int i = 0; i < RLCAP; i++) {
            const u32 pk = (u32)s_list[(size_t)ql * RLCAP + i];
            ids[i] = ((u32)(yp - n + (int)(pk >> IB)) << 16) | (pk & ((1u << IB) - 1u));
        }
        if ((mode & 2) && ((c >> 16) != 0 || q == W - 1)) {
            uint4 *dst = (uint4 *)k.rlist + pidx * 2;
            dst[0] = make_uint4(ids[0], ids[1], ids[2], ids[3]);
            dst[1] = make_uint4(ids[4], ids[5], ids[6], ids[7]);
        }
        if (!(mode & 1)) continue;
        if (cnt <= RLCAP) {
            r_replay_list(k, f, yp, q, pidx, cnt, ids);
        } else { // overflowed list (typically column W-1, the wraparound target of every hint near the left border)
            u8 *px = k.r + pidx * k.C;
            u8 R[4] = {0, 0, 0, 0};
            for (int j = 0; j < k.C; j++) R[j] = px[j];
            r_chain(k, f, yp, q, R, H, 0x7FFFFFFF, 0x7FFFFFFF);
            for (int j = 0; j < k.C; j++) px[j] = R[j];
        }
    }
}
#define RR_NT 256 // (192 .. 256 threads per row measure the same, 320 and more lose a third)
static size_t r_rows_lds(int W, bool wide_ids) { return (size_t)W * (4 + RLCAP * (wide_ids ? 4 : 2) + 4) + 16; }

// ---------------------------------------------------------------------------------------
// 4. L pixels (they replay R chains from the ORIGINAL right image: before the R replay unless the caller kept a copy)
// ---------------------------------------------------------------------------------------
// one hint at (yy, xx) acting on L pixel (yp, xp)
// `defer` != nullptr: the caller cannot afford the occluded-hint branch (it replays R sub-chains: ten times the work of a
// plain blend, and one such lane holds its whole wave up); instead of taking it, set *defer and return.
// (`gv_in` < 0: the hint's value comes with its record in `dense`, one load instead of two dependent ones)
// LIGHT: the caller always defers (`defer` is not null); the occluded-hint branch and the three R sub-chain replays it
// inlines -- nine tenths of the code, half of the registers -- are not compiled into such a kernel at all.
template <bool LIGHT = false>
// `side` >= 0: the pixel is replayed by a PAIR of lanes (side = lane & 1) that walk it in step; each replays one of the two
// R sub-chains of an occluded hint and they swap the results, half the dependent trips through memory per hint.
__device__ __forceinline__ bool l_apply_hint(const VppK &k, int f, int yp, int xp, int yy, int xx, float gv_in, u8 (&L)[4],
                                             const u8 *rnd_f, bool *defer = nullptr, int side = -1)
{
    const int W = k.W, H = k.H;
    const size_t rowoff = ((size_t)f * H + yy) * W;
    const uint4 dn = k.dense[rowoff + xx];
    const float gv = gv_in < 0 ? __uint_as_float(dn.w) : gv_in;
    HintGeo h;
    decode_hint(xx, gv, dn.y, dn.z, h);
    const int yw = yp - yy, xw = xp - xx;
    if (yw < -h.nk || yw > h.nk || xw < -h.nk || xw > h.nk) return false;
    if (k.use_bil && !gate_pass(k, f, gv, yp, xp)) return false;
    const u8 *rnd = rnd_f + k.row_base[(size_t)f * H + yy];
    const int idx = k.uniform ? 0 : hint_idx(k, f, yy, xx, gv, h.nk, yw, xw);
    const int xd0 = xx - h.d0, xd1 = xx - h.d1, xd = xx - h.d;
    if (0 <= xd0 + xw && xd0 + xw <= W - 1) {                       // pyx:104
        if (!h.occ) {                                                 // pyx:106-107
            u8 dv[4];
            draw4(k, rnd, h, idx, dv);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j < k.C) L[j] = blend1(dv[j], k.c, L[j]);
            return true;
        } else if (!k.discard) {                                      // pyx:114-122
            if (LIGHT || defer) {
                *defer = true;
                return false;
            }
            if (k.interp) {
                const int q0 = xd0 + xw;
                int q1 = xd1 + xw;
                q1 = q1 < 0 ? q1 + W : q1;                           // unguarded read, pyx:119
                u8 R0[4], R1[4];
                if (side >= 0) {
                    const int qm = side ? q1 : q0;
                    u8 Rm[4];
                    load_px4(k.r_src + (((size_t)f * H + yp) * W + qm) * k.C, k.C, Rm);
                    r_chain(k, f, yp, qm, Rm, yy, (int)dn.x, xw);
                    const u32 mine = (u32)Rm[0] | ((u32)Rm[1] << 8) | ((u32)Rm[2] << 16) | ((u32)Rm[3] << 24);
                    const u32 other = (u32)__shfl_xor((int)mine, 1);
                    const u32 v0 = side ? other : mine, v1 = side ? mine : other;
#pragma unroll
                    for (int j = 0; j < 4; j++) R0[j] = (u8)(v0 >> (8 * j)), R1[j] = (u8)(v1 >> (8 * j));
                } else {
                const u8 *r0p = k.r_src + (((size_t)f * H + yp) * W + q0) * k.C;
                const u8 *r1p = k.r_src + (((size_t)f * H + yp) * W + q1) * k.C;
                load_px4(r0p, k.C, R0); // (unconditional loads; channels that do not exist are never used)
                load_px4(r1p, k.C, R1);
                r_chain(k, f, yp, q0, R0, yy, (int)dn.x, xw);
                if (q1 == q0) {
                    for (int j = 0; j < 4; j++) R1[j] = R0[j];
                } else {
                    r_chain(k, f, yp, q1, R1, yy, (int)dn.x, xw);
                }
                }
                for (int j = 0; j < k.C; j++) L[j] = blend_l_occ(R0[j], R1[j], h.beta, k.c, L[j]);
            } else {
                int qd = xd + xw;
                qd = qd < 0 ? qd + W : qd;
                u8 Rd[4];
                const u8 *rdp = k.r_src + (((size_t)f * H + yp) * W + qd) * k.C;
                load_px4(rdp, k.C, Rd);
                r_chain(k, f, yp, qd, Rd, yy, (int)dn.x, xw);
                for (int j = 0; j < k.C; j++) L[j] = blend_l_occ_ni(Rd[j], k.c, L[j]);
            }
            return true;
        }
        return false;
    }
    u8 dv[4];
    draw4(k, rnd, h, idx, dv);
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < k.C) L[j] = blend1(dv[j], k.c, L[j]); // pyx:123-124
    return true;
}

// What l_apply_hint decides before it needs the draws (no bilateral gate): does the hint blend plainly into the pixel
// (pyx:106-107,123-124), is it an occluded hint that needs R (pyx:114-122), or neither; and where its draws are.
struct LPrep {
    bool plain, occl;
    const u8 *dp;
    u32 stride;
};
__device__ __forceinline__ void l_prep(const VppK &k, int f, int yp, int xp, int yy, int xx, const uint4 dn, const u8 *rnd, LPrep &o)
{
    HintGeo h;
    decode_hint(xx, __uint_as_float(dn.w), dn.y, dn.z, h);
    const int yw = yp - yy, xw = xp - xx;
    const bool act = !(yw < -h.nk || yw > h.nk || xw < -h.nk || xw > h.nk);
    const int idx = k.uniform ? 0 : hint_idx(k, f, yy, xx, h.g, h.nk, yw, xw); // (arithmetic only without the gate)
    const int t0 = xx - h.d0 + xw;
    const bool inr = 0 <= t0 && t0 <= k.W - 1;                    // pyx:104
    o.plain = act && (!inr || !h.occ);
    o.occl = act && inr && h.occ && !k.discard;
    o.dp = o.plain ? rnd + h.base + (u32)idx : k.rnd;              // (a hint that does not blend re-reads the first byte of the stream)
    o.stride = o.plain ? (k.uniform ? 1u : (u32)h.cnt) : 0u;
}

// The replay of one L pixel: the hints of its window in scan order.  Returns false when the pixel was deferred.
template <int NWIN, bool LIGHT = false>
__device__ __forceinline__ bool l_replay_pixel(const VppK &k, int f, int yp, int xp, unsigned long long mask, const float *gf,
                                               const u8 *rnd_f, bool may_defer,
                                               int side = -1 /* >= 0: one of a pair of lanes on this pixel (l_apply_hint) */)
{
    constexpr int n = (NWIN - 1) / 2;
    const int W = k.W, H = k.H;
    u8 *px = k.l + (((size_t)f * H + yp) * W + xp) * k.C;
    u8 L[4];
    load_px4(px, k.C, L);
    bool touched = false, defer = false;
    // The first two hints of the window at once (most windows hold one or two): both records, both row bases and then both
    // sets of draws are in flight together, two round trips through memory instead of four.  Only where every hint is a
    // plain blend or a deferral (no occlusion mask, or the pass that defers) and the patch has no bilateral gate.
    if (!gf && !k.use_bil && (LIGHT || k.occ == nullptr) && mask) {
        const int q0 = __ffsll((long long)mask) - 1;
        const unsigned long long m1 = mask & (mask - 1);
        const bool two = m1 != 0;
        const int q1 = two ? __ffsll((long long)m1) - 1 : q0;
        LPrep p0, p1;
        const int yy0 = yp - n + q0 / NWIN, yy1 = yp - n + q1 / NWIN;
        const int xx0 = k.direction ? xp - n + q0 % NWIN : xp + n - q0 % NWIN, xx1 = k.direction ? xp - n + q1 % NWIN : xp + n - q1 % NWIN;
        const uint4 dn0 = k.dense[((size_t)f * H + yy0) * W + xx0], dn1 = k.dense[((size_t)f * H + yy1) * W + xx1];
        const u32 rb0 = k.row_base[(size_t)f * H + yy0], rb1 = k.row_base[(size_t)f * H + yy1];
        l_prep(k, f, yp, xp, yy0, xx0, dn0, rnd_f + rb0, p0);
        l_prep(k, f, yp, xp, yy1, xx1, dn1, rnd_f + rb1, p1);
        if (p0.occl || (two && p1.occl)) return false; // (only with LIGHT: the pixel goes to the second pass as it is)
        u8 d0[4], d1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) d0[j] = p0.dp[(u32)(j < k.C ? j : 0) * p0.stride];
#pragma unroll
        for (int j = 0; j < 4; j++) d1[j] = p1.dp[(u32)(j < k.C ? j : 0) * p1.stride];
        if (p0.plain) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j < k.C) L[j] = blend1(d0[j], k.c, L[j]);
        }
        if (two && p1.plain) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j < k.C) L[j] = blend1(d1[j], k.c, L[j]);
        }
        touched = p0.plain || (two && p1.plain);
        mask = two ? m1 & (m1 - 1) : 0ull;
    }
    while (mask) {
        const int q = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int yy = yp - n + q / NWIN;
        const int xx = k.direction ? xp - n + q % NWIN : xp + n - q % NWIN;
        touched |= l_apply_hint<LIGHT>(k, f, yp, xp, yy, xx, gf ? gf[(size_t)yy * W + xx] : -1.0f, L, rnd_f, (LIGHT || may_defer) ? &defer : nullptr, side);
        if (defer) return false;
    }
    if (touched && side <= 0) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j < k.C) px[j] = L[j];
    }
    return true;
}

// the window of hint values around an L pixel as a bit mask in scan order (pyx:78,129)
// (The loads are UNCONDITIONAL, at clamped coordinates, and the "inside the frame" test selects afterwards: hipcc never
// speculates a load, so `in ? g[...] : 0` becomes a branch + load + s_waitcnt vmcnt(0) per window position -- nine serialised
// round trips through memory where one was meant.)
template <int NWIN>
__device__ __forceinline__ unsigned long long l_window_mask(const VppK &k, const float *gf, int yp, int xp)
{
    constexpr int n = (NWIN - 1) / 2, NP = NWIN * NWIN;
    const int W = k.W, H = k.H;
    unsigned long long mask = 0;
    float gw[NP];
#pragma unroll
    for (int wa = 0; wa < NWIN; wa++)
#pragma unroll
        for (int wb = 0; wb < NWIN; wb++) {
            const int yc = min(max(yp - n + wa, 0), H - 1);
            const int xc = min(max(k.direction ? xp - n + wb : xp + n - wb, 0), W - 1);
            gw[wa * NWIN + wb] = gf[(size_t)yc * W + xc];
        }
#pragma unroll
    for (int wa = 0; wa < NWIN; wa++)
#pragma unroll
        for (int wb = 0; wb < NWIN; wb++) {
            const int yy = yp - n + wa;
            const int xx = k.direction ? xp - n + wb : xp + n - wb;
            const bool in = yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1;
            mask |= (in && gw[wa * NWIN + wb] > 0) ? (1ull << (wa * NWIN + wb)) : 0ull;
        }
    return mask;
}

// second pass: the deferred pixels, a pair of lanes each (grid-stride over the work list); the pair replays the pixel in
// step and splits the two R sub-chains of every occluded hint between its lanes
template <int NWIN>
__global__ void __launch_bounds__(64) apply_l_heavy_kernel(VppK k)
{
    const int cnt = *k.lwork_cnt;
    const int side = threadIdx.x & 1;
    for (int e = blockIdx.x * 32 + (threadIdx.x >> 1); e < cnt; e += gridDim.x * 32) {
        const int f = (int)k.lwork[2 * (size_t)e];
        const u32 yx = k.lwork[2 * (size_t)e + 1];
        const int yp = (int)(yx >> 16), xp = (int)(yx & 0xFFFFu);
        const float *gf = k.g + (size_t)f * k.H * k.W;
        const unsigned long long mask = l_window_mask<NWIN>(k, gf, yp, xp);
        (void)l_replay_pixel<NWIN>(k, f, yp, xp, mask, gf, k.rnd + (size_t)f * k.rnd_cap, false, side);
    }
}

// The pixel-driven form of the same replay, for every frame (sparse or dense) with n <= 3.  A block owns ROWS rows x 1024
// scan positions.  Discovery: the hints of a pixel's window come from the row bitmaps compact_kernel left (bit = scan
// position, so a window row is NWIN consecutive bits and the mask is in scan order as it stands); the bitmap rows of the
// block sit in LDS, a pixel costs a few LDS reads and shifts instead of NWIN^2 scattered float loads.  Three pixels in
// four have an empty window.  Compaction: the others are appended to a list in LDS (one LDS atomic per wave and item), so
// that the replay -- a few hundred instructions of float64 blends per hint -- runs with full waves instead of a quarter of
// the lanes; no wave loops over the hints of a row, a single frame has thousands of independent waves.
#define LB_CW 18 // bitmap words a block needs per row: 16 + one before + one after
// bits [p - n, p + n] of window row `wa` of the pixel at block row r, chunk c, lane: they start at bit 64 + lane - n of the
// three 64-bit words c, c + 1, c + 2 of the staged row (c = the word before the chunk's own), i.e. in 32-bit word
// (64 + lane - n) >> 5 of the six: one two-word LDS read and one v_alignbit
template <int NWIN>
__device__ __forceinline__ u32 lb_row_bits(const unsigned long long (*s_w)[LB_CW], int row, int c, int lane)
{
    constexpr int n = (NWIN - 1) / 2;
    const int sb = 64 + lane - n;
    const u32 *w32 = (const u32 *)s_w[row] + 2 * c + (sb >> 5);
    return __builtin_amdgcn_alignbit(w32[1], w32[0], (u32)(sb & 31)) & ((1u << NWIN) - 1u);
}
template <int NWIN>
__device__ __forceinline__ bool lb_any(const unsigned long long (*s_w)[LB_CW], int r, int c, int lane)
{
    u32 any = 0;
#pragma unroll
    for (int wa = 0; wa < NWIN; wa++) any |= lb_row_bits<NWIN>(s_w, r + wa, c, lane);
    return any != 0;
}
template <int NWIN>
__device__ __forceinline__ unsigned long long lb_mask(const unsigned long long (*s_w)[LB_CW], int r, int c, int lane)
{
    unsigned long long mask = 0;
#pragma unroll
    for (int wa = 0; wa < NWIN; wa++) mask |= (unsigned long long)lb_row_bits<NWIN>(s_w, r + wa, c, lane) << (wa * NWIN);
    return mask;
}
template <int NWIN, int ROWS>
__global__ void __launch_bounds__(256) apply_l_bits_kernel(VppK k)
{
    constexpr int n = (NWIN - 1) / 2;
    __shared__ unsigned long long s_w[ROWS + 2 * n][LB_CW];
    __shared__ unsigned short s_list[ROWS * 1024], s_dfr[ROWS * 1024];
    __shared__ int s_total, s_ndfr;
    const int W = k.W, H = k.H;
    const int y0 = blockIdx.y * ROWS, f = blockIdx.z, c0 = blockIdx.x * 16;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nw = vpp_bits_words(W);
    if (t == 0) s_total = 0, s_ndfr = 0;
    for (int i = t; i < (ROWS + 2 * n) * LB_CW; i += 256) { // rows outside the frame and words past the row read as empty
        const int yy = y0 - n + i / LB_CW, gw = c0 + i % LB_CW;
        const unsigned long long v = k.bits[((size_t)f * H + min(max(yy, 0), H - 1)) * nw + min(gw, nw - 1)];
        s_w[i / LB_CW][i % LB_CW] = (yy >= 0 && yy <= H - 1 && gw < nw) ? v : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = j * 4 + wv; // chunk of 64 scan positions inside the block
            const int pp = c * 64 + lane;
            const bool hit = y0 + r <= H - 1 && c0 * 64 + pp < W && lb_any<NWIN>(s_w, r, c, lane);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            if (bal) {
                const int leader = __builtin_ctzll(bal);
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_total, (int)__popcll(bal));
                base = __builtin_amdgcn_readlane(base, leader);
                if (hit) s_list[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (unsigned short)(r * 1024 + pp);
            }
        }
    __syncthreads();
    const int total = s_total;
    const bool may_defer = k.occ != nullptr && !k.discard && k.lwork != nullptr;
    const u8 *rnd_f = k.rnd + (size_t)f * k.rnd_cap;
    for (int e = t; e < total; e += 256) {
        const int id = s_list[e], r = id >> 10, pp = id & 1023;
        const int yp = y0 + r, p = c0 * 64 + pp;
        const int xp = k.direction ? p : W - 1 - p;
        unsigned long long mask = lb_mask<NWIN>(s_w, r, pp >> 6, pp & 63);
        const bool dfr = !l_replay_pixel<NWIN, true>(k, f, yp, xp, mask, nullptr, rnd_f, may_defer);
        // deferred pixels: collected in LDS (in place: slot <= e, and every entry up to e has been consumed), one LDS atomic
        // per wave; the block then takes its share of the global work list with ONE atomic (atomics on a single address
        // queue up in one L2 channel: one per wave and round made them the longest part of this kernel)
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(dfr);
        if (dm) {
            const int leader = __builtin_ctzll(dm);
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_ndfr, (int)__popcll(dm));
            base = __builtin_amdgcn_readlane(base, leader);
            if (dfr) s_dfr[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(dm >> 32), __builtin_amdgcn_mbcnt_lo((u32)dm, 0u))] = (unsigned short)id;
        }
    }
    if (!may_defer) return; // (uniform)
    __syncthreads();
    const int nd = s_ndfr;
    if (nd == 0) return;
    if (t == 0) s_total = atomicAdd(k.lwork_cnt, nd);
    __syncthreads();
    const int gbase = s_total;
    for (int e = t; e < nd; e += 256) {
        const int id = s_dfr[e], p = c0 * 64 + (id & 1023);
        k.lwork[2 * (size_t)(gbase + e)] = (u32)f;
        k.lwork[2 * (size_t)(gbase + e) + 1] = ((u32)(y0 + (id >> 10)) << 16) | (u32)(k.direction ? p : W - 1 - p);
    }
}

// Patches wider than 7 x 7 (n > 3: the window no longer fits a 64-bit mask): one thread per pixel walks the window's hint
// values in scan order.  (grid: a bounded number of blocks per frame striding over its (row, 256-column block) pairs)
__global__ void __launch_bounds__(256) apply_l_wide_kernel(VppK k)
{
    const int f = blockIdx.y;
    const int W = k.W, H = k.H;
    const u8 *rnd_f = k.rnd + (size_t)f * k.rnd_cap;
    const float *gf = k.g + (size_t)f * H * W;
    const int nxb = (W + 255) / 256;
    for (int t = blockIdx.x; t < nxb * H; t += gridDim.x) {
        const int yp = t / nxb, xp = (t % nxb) * 256 + threadIdx.x;
        if (xp >= W) continue;
        u8 *px = k.l + (((size_t)f * H + yp) * W + xp) * k.C;
        bool touched = false;
        u8 L[4] = {0, 0, 0, 0};
        for (int j = 0; j < k.C; j++) L[j] = px[j];
        for (int yy = max(0, yp - k.n); yy <= min(H - 1, yp + k.n); yy++) {
            const int xa = max(0, xp - k.n), xb = min(W - 1, xp + k.n);
            for (int s2 = 0; s2 <= xb - xa; s2++) {
                const int xx = k.direction ? xa + s2 : xb - s2; // scan order inside the row (pyx:78,129)
                const float gv = gf[(size_t)yy * W + xx];
                if (!(gv > 0)) continue;
                touched |= l_apply_hint(k, f, yp, xp, yy, xx, gv, L, rnd_f);
            }
        }
        if (touched)
            for (int j = 0; j < k.C; j++) px[j] = L[j];
    }
}

// ---------------------------------------------------------------------------------------
// maxDistance colour method (vpp_core_opt.pyx:133-341).  The colour of every patch pixel is
// found by a greedy, order-dependent walk over a (wsize_agg_y x wsize_agg_x) window of the
// CURRENT left and right images, so hints are truly serial (SURVEY C-12); only channels and
// frames are independent.  One wave owns one (frame, channel) chain and walks the hints in
// scan order; inside a colour search the 64 lanes hold 64 window columns and the sequential
// walk jumps from one interval-shrinking sample to the next with ballot + ffs, which visits
// exactly the samples the reference's loop acts on, in the same order.
// All image accesses of a chain go through L2-coherent (agent-scope relaxed atomic) byte
// loads/stores: the chain reads bytes it wrote itself a few instructions earlier.
// ---------------------------------------------------------------------------------------
struct MdK {
    VppK k;
    int n_agg_x, n_agg_y;
};

__device__ __forceinline__ u32 md_ld(const u8 *p) { return (u32)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void md_st(u8 *p, u8 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Where a chain keeps the pixels it reads and writes.
//   MemGlobal: in place in HBM/L2 (every access an L2 round trip; any parameters).
//   MemLds   : the rows a hint row can touch ([y-rad, y+rad], rad = patch radius + wsize_agg_y/2) of the
//              chain's L and R channel planes live in an LDS ring of NR = 2*rad+1 rows; rows are loaded
//              once and written back once as the chain moves down the image.  LDS accesses of one wave
//              execute in order, so a lane-0 store is seen by the next load of any lane.
struct MemGlobal {
    u8 *lch, *rch;
    int W, C;
    __device__ __forceinline__ u32 ldL(int y, int x) const { return md_ld(lch + ((size_t)y * W + x) * C); }
    __device__ __forceinline__ u32 ldR(int y, int x) const { return md_ld(rch + ((size_t)y * W + x) * C); }
    __device__ __forceinline__ void stL(int y, int x, u8 v) const { md_st(lch + ((size_t)y * W + x) * C, v); }
    __device__ __forceinline__ void stR(int y, int x, u8 v) const { md_st(rch + ((size_t)y * W + x) * C, v); }
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_s_waitcnt(0); }
};
struct MemLds {
    u8 *sl, *sr; // [NR][W]
    int W, NR;
    int row0, slot0; // current hint row and its ring slot (row0 % NR): rows within +-rad map without a division
    __device__ __forceinline__ int off(int y) const
    {
        int sidx = slot0 + (y - row0);
        sidx = sidx < 0 ? sidx + NR : (sidx >= NR ? sidx - NR : sidx);
        return sidx * W;
    }
    __device__ __forceinline__ u32 ldL(int y, int x) const { return sl[off(y) + x]; }
    __device__ __forceinline__ u32 ldR(int y, int x) const { return sr[off(y) + x]; }
    __device__ __forceinline__ void stL(int y, int x, u8 v) const { sl[off(y) + x] = v; }
    __device__ __forceinline__ void stR(int y, int x, u8 v) const { sr[off(y) + x] = v; }
    // program order is enough inside one wave (LDS executes a wave's accesses in order); the fence only
    // stops the compiler from moving LDS accesses across it
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
};

// (u8)( V*cc + P*(1.0-cc) )                                   pyx:318,324,332,335 (V double)
__device__ __forceinline__ u8 mdblend1(double V, float cc, u8 P)
{
    const double a = __dmul_rn(V, (double)cc);
    const double b = __dmul_rn((double)P, __dsub_rn(1.0, (double)cc));
    return to_u8(__dadd_rn(a, b));
}
// (u8)( (V*cc + R*(1.0-cc)) * (1.0-beta) + R*beta )            pyx:320,327
__device__ __forceinline__ u8 mdblend_r0(double V, float cc, u8 R, float beta)
{
    const double a = __dmul_rn(V, (double)cc);
    const double b = __dmul_rn((double)R, __dsub_rn(1.0, (double)cc));
    const double t = __dmul_rn(__dadd_rn(a, b), __dsub_rn(1.0, (double)beta));
    const double u = (double)__fmul_rn((float)R, beta);
    return to_u8(__dadd_rn(t, u));
}
// (u8)( (V*cc + R*(1.0-cc)) * beta + R*(1.0-beta) )            pyx:322,329
__device__ __forceinline__ u8 mdblend_r1(double V, float cc, u8 R, float beta)
{
    const double a = __dmul_rn(V, (double)cc);
    const double b = __dmul_rn((double)R, __dsub_rn(1.0, (double)cc));
    const double t = __dmul_rn(__dadd_rn(a, b), (double)beta);
    const double u = __dmul_rn((double)R, __dsub_rn(1.0, (double)beta));
    return to_u8(__dadd_rn(t, u));
}

// colour search: pyx:216-260 (uniform, bins_inside) / :269-313 (per patch pixel)
template <typename Mem>
__device__ void md_search(const MdK &m, const Mem &mem, int cy, int cx, int rcx, bool occluded,
                          bool bins_inside, u32 *hist /* LDS [256] */, int &pa_out, int &pb_out)
{
    const int W = m.k.W, H = m.k.H, C = m.k.C;
    const int lane = threadIdx.x & 63;
    int pa = 0, pb = 255;
    int zeros = 0;
    const int ncol = 2 * m.n_agg_x + 1;
    if (ncol <= 64 && m.n_agg_y <= 1) {
        // common window (<= 64 columns, <= 3 rows): all samples are fetched before the walk starts
        const int xx = cx - m.n_agg_x + lane, rx = rcx - m.n_agg_x + lane;
        const bool inL = (lane < ncol) && xx >= 0 && xx <= W - 1;
        const bool rin = rx >= 0 && rx <= W - 1;
        const bool cL0 = inL && (!occluded || !rin), cR0 = inL && rin;
        int Lr[3], Rr[3];
        bool rowok[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int yy = cy + r - 1;
            rowok[r] = (r - 1 >= -m.n_agg_y) && (r - 1 <= m.n_agg_y) && yy >= 0 && yy <= H - 1;
            Lr[r] = (rowok[r] && cL0) ? (int)mem.ldL(yy, xx) : 0;
            Rr[r] = (rowok[r] && cR0) ? (int)mem.ldR(yy, rx) : 0;
        }
        // The walk itself runs on scalar state: pa, pb and the "still to come" lane masks are wave-uniform, so an
        // iteration is two vector range tests + two ballots + a handful of scalar ops + one lane read.
        const unsigned long long baseL = __ballot(cL0), baseR = __ballot(cR0);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            if (!rowok[r]) continue; // wave-uniform
            const int Lv = Lr[r], Rv = Rr[r];
            if (!bins_inside) zeros += __popcll(__ballot(cL0 && Lv == 0)) + __popcll(__ballot(cR0 && Rv == 0));
            unsigned long long aliveL = baseL, aliveR = baseR;
            while (true) {
                // pa < v < pb  <=>  (unsigned)(v - pa - 1) < (unsigned)(pb - pa - 1)
                const u32 wdt = (u32)(pb - pa - 1);
                const unsigned long long mL = __ballot((u32)(Lv - pa - 1) < wdt) & aliveL;
                const unsigned long long mR = __ballot((u32)(Rv - pa - 1) < wdt) & aliveR;
                if ((mL | mR) == 0) break;
                const int fL = mL ? __ffsll((long long)mL) - 1 : 64;
                const int fR = mR ? __ffsll((long long)mR) - 1 : 64;
                const bool isL = fL <= fR; // sequence order: left sample of a column before its right sample
                const int fl = __builtin_amdgcn_readfirstlane(isL ? fL : fR);
                const int p = isL ? __builtin_amdgcn_readlane(Lv, fl) : __builtin_amdgcn_readlane(Rv, fl);
                if (p - pa > pb - p) pb = p;
                else if (p - pa < pb - p) pa = p;
                const unsigned long long upto = (fl >= 63) ? ~0ull : ((2ull << fl) - 1); // columns <= fl
                aliveL = baseL & ~upto;
                aliveR = baseR & ~(isL ? (upto >> 1) : upto);                          // right sample of column fl still to come
            }
        }
    } else
    for (int yw = -m.n_agg_y; yw <= m.n_agg_y; yw++) {
        const int yy = cy + yw;
        if (yy < 0 || yy > H - 1) continue;
        for (int c0 = 0; c0 < ncol; c0 += 64) {
            const int i = c0 + lane;
            const int xx = cx - m.n_agg_x + i, rx = rcx - m.n_agg_x + i;
            const bool inL = (i < ncol) && xx >= 0 && xx <= W - 1;
            const bool rin = rx >= 0 && rx <= W - 1;
            const bool condL = inL && (!occluded || !rin);
            const bool condR = inL && rin;
            const int Lv = condL ? (int)mem.ldL(yy, xx) : 0;
            const int Rv = condR ? (int)mem.ldR(yy, rx) : 0;
            if (!bins_inside) {
                zeros += __popcll(__ballot(condL && Lv == 0)) + __popcll(__ballot(condR && Rv == 0));
            }
            int pos = 0; // next sequence position (2*lane = left sample, 2*lane+1 = right sample)
            while (true) {
                const bool candL = condL && Lv > pa && Lv < pb && (2 * lane >= pos);
                const bool candR = condR && Rv > pa && Rv < pb && (2 * lane + 1 >= pos);
                const unsigned long long mk = __ballot(candL || candR);
                if (mk == 0) break;
                // the first candidate lane is wave-uniform: scalar lane reads instead of LDS-crossbar shuffles
                const int fl = __builtin_amdgcn_readfirstlane(__ffsll((long long)mk) - 1);
                const int isL = __builtin_amdgcn_readlane((int)candL, fl);
                const int p = __builtin_amdgcn_readlane(isL ? Lv : Rv, fl);
                if (p - pa > pb - p) pb = p;
                else if (p - pa < pb - p) pa = p;
                pos = 2 * fl + (isL ? 1 : 2);
            }
        }
    }
    if (!bins_inside && zeros == 256) {
        // n_bins == 0 (exactly 256 zero-valued samples): least-used bin, first minimum (pyx:305-313)
        for (int b = lane; b < 256; b += 64) hist[b] = 0;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (int yw = -m.n_agg_y; yw <= m.n_agg_y; yw++) {
            const int yy = cy + yw;
            if (yy < 0 || yy > H - 1) continue;
            for (int c0 = 0; c0 < ncol; c0 += 64) {
                const int i = c0 + lane;
                const int xx = cx - m.n_agg_x + i, rx = rcx - m.n_agg_x + i;
                const bool inL = (i < ncol) && xx >= 0 && xx <= W - 1;
                const bool rin = rx >= 0 && rx <= W - 1;
                if (inL && (!occluded || !rin)) atomicAdd(&hist[mem.ldL(yy, xx)], 1u);
                if (inL && rin) atomicAdd(&hist[mem.ldR(yy, rx)], 1u);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        u32 key = 0xFFFFFFFFu; // count*256 + bin: first minimum wins
        for (int b = lane; b < 256; b += 64) key = min(key, (hist[b] << 8) | (u32)b);
        for (int off = 32; off >= 1; off >>= 1) key = min(key, (u32)__shfl_xor((int)key, off));
        pa = pb = (int)(key & 0xFFu);
    }
    pa_out = pa;
    pb_out = pb;
}

// one hint of row y of chain (f, j): colour search(es) + the patch blends (pyx:200-341)
template <typename Mem>
__device__ __forceinline__ void md_hint(const MdK &m, const Mem &mem, int f, int y, const HintRec &rec, u32 *hist, int &pa, int &pb)
{
    const VppK &k = m.k;
    const int W = k.W, H = k.H;
    const int lane = threadIdx.x & 63;
    {
        HintGeo h;
        decode_hint(rec.x, rec.g, rec.base, rec.flags, h);
        const int x = h.x;
        const bool occ = h.occ != 0;
        const int xd0 = x - h.d0, xd1 = x - h.d1, xd = x - h.d;
        if (k.uniform) md_search(m, mem, y, x, xd, occ, true, hist, pa, pb); // pyx:216-260
        for (int yw = -h.nk; yw <= h.nk; yw++)
            for (int xw = -h.nk; xw <= h.nk; xw++) {
                const int py = y + yw, pxx = x + xw;
                if (py < 0 || py > H - 1 || pxx < 0 || pxx > W - 1) continue;      // pyx:267
                if (k.use_bil && !gate_pass(k, f, h.g, py, pxx)) continue;        // vpp_standalone.py:154
                if (!k.uniform) md_search(m, mem, py, pxx, xd + xw, occ, false, hist, pa, pb); // pyx:269-313
                const double V = __ddiv_rn((double)(pa + pb), 2.0);
                if (0 <= xd0 + xw && xd0 + xw <= W - 1) {                          // pyx:315
                    if (!occ) {                                                    // pyx:317-324
                        const u8 Lo = (u8)mem.ldL(py, pxx);
                        const u8 Ln = mdblend1(V, k.c, Lo);
                        if (k.interp) {
                            const int q0 = xd0 + xw;
                            const u8 R0n = mdblend_r0(V, k.c, (u8)mem.ldR(py, q0), h.beta);
                            if (lane == 0) { mem.stL(py, pxx, Ln); mem.stR(py, q0, R0n); }
                            if (0 <= xd1 + xw && xd1 + xw <= W - 1) {
                                const int q1 = xd1 + xw;
                                mem.sync();
                                const u8 R1n = mdblend_r1(V, k.c, (u8)mem.ldR(py, q1), h.beta);
                                if (lane == 0) mem.stR(py, q1, R1n);
                            }
                        } else {
                            int q = xd + xw;
                            q = q < 0 ? q + W : q;
                            const u8 Rn = mdblend1(V, k.c, (u8)mem.ldR(py, q));
                            if (lane == 0) { mem.stL(py, pxx, Ln); mem.stR(py, q, Rn); }
                        }
                    } else if (!k.discard) {                                       // pyx:325-333
                        if (k.interp) {
                            const int q0 = xd0 + xw;
                            int q1 = xd1 + xw;
                            const bool in1 = q1 >= 0 && q1 <= W - 1;
                            q1 = q1 < 0 ? q1 + W : q1;
                            const u8 R0n = mdblend_r0(V, k.c_occ, (u8)mem.ldR(py, q0), h.beta);
                            if (lane == 0) mem.stR(py, q0, R0n);
                            mem.sync();
                            u8 R1v = (u8)mem.ldR(py, q1); // after the r0 store: r1 may alias r0
                            if (in1) {
                                R1v = mdblend_r1(V, k.c_occ, R1v, h.beta);
                                if (lane == 0) mem.stR(py, q1, R1v);
                                mem.sync();
                            }
                            const u8 R0v = (q1 == q0) ? R1v : R0n;
                            const u8 Ln = blend_l_occ(R0v, R1v, h.beta, k.c, (u8)mem.ldL(py, pxx)); // pyx:330
                            if (lane == 0) mem.stL(py, pxx, Ln);
                        } else {
                            int q = xd + xw;
                            q = q < 0 ? q + W : q;
                            const u8 Rn = mdblend1(V, k.c_occ, (u8)mem.ldR(py, q));
                            const u8 Ln = blend_l_occ_ni(Rn, k.c, (u8)mem.ldL(py, pxx));    // pyx:333
                            if (lane == 0) { mem.stR(py, q, Rn); mem.stL(py, pxx, Ln); }
                        }
                    }
                } else {                                                           // pyx:334-335
                    const u8 Ln = mdblend1(V, k.c, (u8)mem.ldL(py, pxx));
                    if (lane == 0) mem.stL(py, pxx, Ln);
                }
                mem.sync(); // stores of this patch pixel land before the next search reads
            }
    }
}

// all hints of row y of chain (f, j), in scan order
template <typename Mem>
__device__ __forceinline__ void md_row(const MdK &m, const Mem &mem, int f, int y, u32 *hist, int &pa, int &pb)
{
    const VppK &k = m.k;
    const size_t rowoff = ((size_t)f * k.H + y) * k.W;
    const int cnt = k.row_count[(size_t)f * k.H + y];
    for (int i = 0; i < cnt; i++) md_hint(m, mem, f, y, k.rec[rowoff + i], hist, pa, pb);
}

// in-place variant (any parameters)
__global__ void __launch_bounds__(64) maxdist_kernel(MdK m)
{
    __shared__ u32 hist[256];
    const VppK &k = m.k;
    const int f = blockIdx.x / k.C, j = blockIdx.x % k.C;
    MemGlobal mem;
    mem.lch = k.l + (size_t)f * k.H * k.W * k.C + j;
    mem.rch = k.r + (size_t)f * k.H * k.W * k.C + j;
    mem.W = k.W;
    mem.C = k.C;
    int pa = 0, pb = 255; // pyx:196-197
    for (int y = 0; y < k.H; y++) md_row(m, mem, f, y, hist, pa, pb);
}

// LDS-resident variant: `rad` rows above and below the hint row are kept in a ring of NR = 2*rad+1 rows
__global__ void __launch_bounds__(64) maxdist_lds_kernel(MdK m, int rad)
{
    __shared__ u32 hist[256];
    extern __shared__ __attribute__((aligned(16))) u8 md_rows[]; // [2][NR][W]
    const VppK &k = m.k;
    const int f = blockIdx.x / k.C, j = blockIdx.x % k.C;
    const int W = k.W, H = k.H, C = k.C, NR = 2 * rad + 1;
    const int lane = threadIdx.x & 63;
    u8 *lch = k.l + (size_t)f * H * W * C + j;
    u8 *rch = k.r + (size_t)f * H * W * C + j;
    MemLds mem;
    mem.sl = md_rows;
    mem.sr = md_rows + (size_t)NR * W;
    mem.W = W;
    mem.NR = NR;
    auto load_row = [&](int r) {
        const int o = (r % NR) * W;
        for (int x = lane; x < W; x += 64) {
            mem.sl[o + x] = lch[((size_t)r * W + x) * C];
            mem.sr[o + x] = rch[((size_t)r * W + x) * C];
        }
    };
    auto store_row = [&](int r) {
        const int o = (r % NR) * W;
        for (int x = lane; x < W; x += 64) {
            lch[((size_t)r * W + x) * C] = mem.sl[o + x];
            rch[((size_t)r * W + x) * C] = mem.sr[o + x];
        }
    };
    int lo = 0, hi = -1; // rows [lo, hi] are resident
    int pa = 0, pb = 255; // pyx:196-197
    for (int y = 0; y < H; y++) {
        if (k.row_count[(size_t)f * H + y] == 0) continue;
        const int nlo = max(0, y - rad), nhi = min(H - 1, y + rad);
        if (nlo > hi) { // no overlap with what is resident: write everything back, start afresh
            for (int r = lo; r <= hi; r++) store_row(r);
            lo = nlo;
            hi = nlo - 1;
        }
        for (; lo < nlo; lo++) store_row(lo); // rows the chain has left behind
        for (; hi < nhi;) load_row(++hi);
        __builtin_amdgcn_s_waitcnt(0);
        mem.row0 = y;
        mem.slot0 = y % NR;
        md_row(m, mem, f, y, hist, pa, pb);
    }
    for (int r = lo; r <= hi; r++) store_row(r);
}

// ---------------------------------------------------------------------------------------
// Row wavefront (SURVEY C-12).  Hints of one (frame, channel) chain conflict only when their footprints overlap: a
// hint of row y reads rows y +- rad (rad = patch radius n + vertical half window) and writes rows y +- n, so rows
// further apart than dep = n + rad never interact, and inside that band only hints whose column ranges overlap do:
//   left image : |x1 - x2| <= 2n + n_agg_x
//   right image: [x - ceil(g) - n - n_agg_x, x - floor(g) + n + n_agg_x] of the two hints overlap
// (plus everything that indexes column -1 -> W-1, SURVEY C-1/C-2: such a hint waits for whole rows).
// One workgroup of MD_NW waves owns a chain; wave w walks rows w, w + MD_NW, ...  Before a hint runs, its wave waits
// until each of the dep rows above has completed its LAST hint that conflicts with it (the lanes test the hints of
// such a row in parallel; a row's hints complete in index order, so one progress counter per row suffices).  Progress
// counters, the image rows the chain works on (ring of MD_RING rows of both channel planes) and the completion order
// live in LDS.
// Scan direction x ascending only (direction = 1, what vpp() uses); other cases take the one-wave kernels.
// ---------------------------------------------------------------------------------------
#define MD_NW 16
#define MD_RING 32
#define MD_PRE_ROWS 4 // rows above with precomputed wait targets (dep = 2n + agg_y/2: 3 for the defaults)
struct MemRing {
    u8 *sl, *sr; // [MD_RING][W]
    int W;
    __device__ __forceinline__ int off(int y) const { return (y & (MD_RING - 1)) * W; }
    __device__ __forceinline__ u32 ldL(int y, int x) const { return sl[off(y) + x]; }
    __device__ __forceinline__ u32 ldR(int y, int x) const { return sr[off(y) + x]; }
    __device__ __forceinline__ void stL(int y, int x, u8 v) const { sl[off(y) + x] = v; }
    __device__ __forceinline__ void stR(int y, int x, u8 v) const { sr[off(y) + x] = v; }
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
};

__device__ __forceinline__ int md_lds_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void md_lds_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

__global__ void __launch_bounds__(64 * MD_NW) maxdist_wave_kernel(MdK m, int rad)
{
    __shared__ u32 hist_all[MD_NW][256];
    __shared__ u16 preq_all[MD_NW][MD_PRE_ROWS * 64]; // wait targets of the wave's current row, per row above
    __shared__ int prog[2 * MD_RING]; // hints completed in row r (slot r % 32: a slot is recycled 32 rows later, long after its last reader)
    __shared__ int done_upto;       // rows 0..done_upto are complete (completion token passes in row order)
    __shared__ int loaded_upto;     // rows 0..loaded_upto are (or have been) resident in the ring
    extern __shared__ __attribute__((aligned(16))) u8 md_rows[]; // [2][MD_RING][W]
    const VppK &k = m.k;
    const int f = blockIdx.x / k.C, j = blockIdx.x % k.C;
    const int W = k.W, H = k.H, C = k.C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u8 *lch = k.l + (size_t)f * H * W * C + j;
    u8 *rch = k.r + (size_t)f * H * W * C + j;
    MemRing mem;
    mem.sl = md_rows;
    mem.sr = md_rows + (size_t)MD_RING * W;
    mem.W = W;
    u32 *hist = hist_all[wave];
    u16 *preq = preq_all[wave];
    const int dep = k.n + rad;
    auto load_row = [&](int r) { // by one wave
        const int o = mem.off(r);
        for (int x = lane; x < W; x += 64) {
            mem.sl[o + x] = lch[((size_t)r * W + x) * C];
            mem.sr[o + x] = rch[((size_t)r * W + x) * C];
        }
    };
    auto store_row = [&](int r) {
        const int o = mem.off(r);
        for (int x = lane; x < W; x += 64) {
            lch[((size_t)r * W + x) * C] = mem.sl[o + x];
            rch[((size_t)r * W + x) * C] = mem.sr[o + x];
        }
    };
    // the first MD_RING rows, all waves
    for (int r = wave; r < MD_RING && r < H; r += MD_NW) load_row(r);
    if (threadIdx.x < 2 * MD_RING) prog[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        done_upto = -1;
        loaded_upto = min(MD_RING, H) - 1;
    }
    __syncthreads();

    int pa = 0, pb = 255;
    for (int y = wave; y < H; y += MD_NW) {
        const int row = f * H + y;
        const size_t rowoff = (size_t)row * W;
        const int cnt = k.row_count[row];
        // rows y +- rad must be resident (this also bounds how far a wave may run ahead of the slowest row)
        const int need = min(H - 1, y + rad);
        while (md_lds_load(&loaded_upto) < need) __builtin_amdgcn_s_sleep(2);
        // Rows with at most 64 hints (and neighbours with at most 64): the wait targets of every hint are worked out
        // before the row starts -- lane k holds hint k of a row above, the row's own hints are broadcast one by one
        // and each conflict test is one ballot -- so that the hint loop itself touches no global memory for them.
        const int q_lo = max(0, y - dep);
        bool pre = cnt <= 64 && (y - q_lo) <= MD_PRE_ROWS;
        for (int q = q_lo; q < y; q++) pre = pre && k.row_count[f * H + q] <= 64;
        if (pre && cnt > 0) {
            const HintRec mine = lane < cnt ? k.rec[rowoff + lane] : HintRec{0, 0.f, 0u, 0u};
            for (int q = q_lo; q < y; q++) {
                const int qcnt = k.row_count[f * H + q];
                const HintRec r1 = lane < qcnt ? k.rec[(size_t)(f * H + q) * W + lane] : HintRec{0, 0.f, 0u, 0u};
                const int f1 = (int)floorf(r1.g), c1 = (int)ceilf(r1.g);
                const int lowR1 = r1.x - c1 - k.n - m.n_agg_x, highR1 = r1.x - f1 + k.n + m.n_agg_x;
                const bool wrap1 = (r1.x - f1) >= -k.n && (r1.x - f1) <= k.n;
                for (int i = 0; i < cnt; i++) {
                    const int x2 = __builtin_amdgcn_readlane(mine.x, i);
                    const float g2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.g), i));
                    const int lowR2 = x2 - (int)ceilf(g2) - k.n - m.n_agg_x, highR2 = x2 - (int)floorf(g2) + k.n + m.n_agg_x;
                    const int xd0 = x2 - (int)floorf(g2);
                    int p_req = qcnt;
                    if (!(xd0 >= -k.n && xd0 <= k.n)) {
                        const bool conflict = lane < qcnt && ((abs(r1.x - x2) <= 2 * k.n + m.n_agg_x) ||
                                                              (lowR1 <= highR2 && highR1 >= lowR2) || wrap1);
                        const unsigned long long mk = __builtin_amdgcn_ballot_w64(conflict);
                        p_req = mk ? 64 - __builtin_clzll(mk) : 0;
                    }
                    if (lane == 0) preq[(q - q_lo) * 64 + i] = (u16)p_req;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        for (int i = 0; i < cnt; i++) {
            const HintRec rec = k.rec[rowoff + i];
            if (pre) {
                for (int q = q_lo; q < y; q++) {
                    const int p_req = preq[(q - q_lo) * 64 + i];
                    while (md_lds_load(&prog[q & (2 * MD_RING - 1)]) < p_req) __builtin_amdgcn_s_sleep(1);
                }
                md_hint(m, mem, f, y, rec, hist, pa, pb);
                mem.sync();
                if (lane == 0) md_lds_store(&prog[y & (2 * MD_RING - 1)], i + 1);
                continue;
            }
            const int x2 = rec.x;
            const int lowR2 = x2 - (int)ceilf(rec.g) - k.n - m.n_agg_x;
            const int highR2 = x2 - (int)floorf(rec.g) + k.n + m.n_agg_x;
            // the only accesses that index column -1 (-> W-1, pyx:113,119,121) belong to patch pixels with xd0 + xw == 0:
            // such a hint reaches across the whole row and waits for complete rows
            const int xd0 = x2 - (int)floorf(rec.g);
            const bool wraps = xd0 >= -k.n && xd0 <= k.n;
            const int reachL = 2 * k.n + m.n_agg_x;
            for (int q = max(0, y - dep); q < y; q++) {
                const int qrow = f * H + q;
                const int qcnt = k.row_count[qrow];
                if (qcnt == 0) continue;
                int p_req = qcnt;
                if (!wraps) {
                    const size_t qoff = (size_t)qrow * W;
                    p_req = 0;
                    for (int base = 0; base < qcnt; base += 64) { // 1 + index of the last conflicting hint of row q
                        const int idx = base + lane;
                        bool conflict = false;
                        if (idx < qcnt) {
                            const HintRec r1 = k.rec[qoff + idx];
                            const int f1 = (int)floorf(r1.g), c1 = (int)ceilf(r1.g);
                            const int lowR1 = r1.x - c1 - k.n - m.n_agg_x, highR1 = r1.x - f1 + k.n + m.n_agg_x;
                            const int xd01 = r1.x - f1;
                            conflict = (abs(r1.x - x2) <= reachL) || (lowR1 <= highR2 && highR1 >= lowR2) ||
                                       (xd01 >= -k.n && xd01 <= k.n); // a hint that may index column W-1
                        }
                        const unsigned long long mk = __builtin_amdgcn_ballot_w64(conflict);
                        if (mk) p_req = base + 64 - __builtin_clzll(mk);
                    }
                }
                while (md_lds_load(&prog[q & (2 * MD_RING - 1)]) < p_req) __builtin_amdgcn_s_sleep(1);
            }
            md_hint(m, mem, f, y, rec, hist, pa, pb);
            mem.sync();
            if (lane == 0) md_lds_store(&prog[y & (2 * MD_RING - 1)], i + 1);
        }
        // completion token: rows finish in order; the holder retires the oldest ring row and brings in the next one
        while (md_lds_load(&done_upto) < y - 1) __builtin_amdgcn_s_sleep(2);
        const int old = y - rad; // no row > y touches rows <= y - rad
        if (old >= 0) {
            store_row(old);
            const int nw = old + MD_RING;
            if (nw < H) {
                load_row(nw);
                if (lane == 0) prog[nw & (2 * MD_RING - 1)] = 0;
            }
        }
        if (y == H - 1)
            for (int r = max(0, y - rad + 1); r <= y; r++) store_row(r);
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            if (old >= 0 && old + MD_RING < H) md_lds_store(&loaded_upto, old + MD_RING);
            md_lds_store(&done_upto, y);
        }
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
u64 vpp_draws_upper_bound(const VppxVppParams &p, const VppGeom &g)
{
    const u64 n = (u64)((p.wsize - 1) / 2);
    const u64 per = p.uniform_color ? 1 : (2 * n + 1) * (2 * n + 1);
    u64 ub = (u64)g.H * g.W * g.C * per;
    ub = (ub + LG - 1) / LG * LG;
    return ub;
}

static Poly31 qoff_for(u64 rand_offset)
{
    static std::mutex m;
    static bool have0 = false;
    static Poly31 q0;
    Poly31 q;
    if (rand_offset == 0) {
        std::lock_guard<std::mutex> lk(m);
        if (!have0) { poly_pow_z(310, q0.c); have0 = true; }
        return q0;
    }
    poly_pow_z(rand_offset + 310, q.c);
    return q;
}

int vpp_launch(vppx_ctx *ctx, const VppxVppParams &p, const VppGeom &g, u8 *l, u8 *r, const float *gmap,
               const u8 *occ, const float *filled_g, int64_t *n_hints_dev, const u32 *seeds_dev, const u8 *r_orig)
{
    if (g.C < 1 || g.C > 4) { vppx_set_error("channels must be 1..4 (got %d)", g.C); return VPPX_E_INVALID_ARG; }
    if (p.wsize < 1 || p.wsize > 31) { vppx_set_error("wsize must be in 1..31 (got %d)", p.wsize); return VPPX_E_INVALID_ARG; }
    if (g.W > 32767 || g.H > 65535) { // hint ranges are int16 columns, list ids are (row << 16 | index in row)
        vppx_set_error("frames larger than 65535 x 32767 are not supported (got %d x %d)", g.H, g.W);
        return VPPX_E_UNSUPPORTED;
    }
    VppK k;
    k.B = g.B; k.H = g.H; k.W = g.W; k.C = g.C;
    k.n = (p.wsize - 1) / 2;
    k.direction = p.direction != 0;
    k.uniform = p.uniform_color != 0;
    k.discard = p.discard_occluded != 0;
    k.interp = p.interpolate != 0;
    k.use_dist = p.use_distance_patch != 0;
    k.use_bil = (p.use_bilateral_patch != 0) && filled_g != nullptr;
    k.c = p.c; k.c_occ = p.c_occ; k.dmin = p.dmin; k.dmax = p.dmax;
    k.range = nullptr;
    k.inv_gamma = 1.0 / p.distance_gamma;
    k.l = l; k.r = r; k.r_src = r_orig ? r_orig : r; k.g = gmap; k.occ = occ; k.filled = filled_g;
    const size_t npx = (size_t)g.B * g.H * g.W;
    int rc;
    if ((rc = ws_get(ctx, WS_HINT_REC, npx, &k.rec))) return rc;
    if ((rc = ws_get(ctx, WS_HINT_DENSE, npx, &k.dense))) return rc;
    if ((rc = ws_get(ctx, WS_HINT_X, npx, &k.rng))) return rc;
    if ((rc = ws_get(ctx, WS_HINT_BITS, (size_t)g.B * g.H * vpp_bits_words(g.W), &k.bits))) return rc;
    k.rcnt = nullptr;
    k.lwork = nullptr;
    k.lwork_cnt = nullptr;
    k.rlist = nullptr;
    if ((rc = ws_get(ctx, WS_ROW_COUNT, (size_t)g.B * g.H, &k.row_count))) return rc;
    if ((rc = ws_get(ctx, WS_ROW_DRAWS, (size_t)g.B * g.H, &k.row_draws))) return rc;
    if ((rc = ws_get(ctx, WS_ROW_BASE, (size_t)g.B * g.H, &k.row_base))) return rc;
    if ((rc = ws_get(ctx, WS_FRAME_TOT, (size_t)g.B * 2, &k.frame_tot))) return rc;
    if (k.use_dist && p.per_frame_range) {
        float2 *range;
        if ((rc = ws_get(ctx, WS_HINT_RANGE, (size_t)g.B, &range))) return rc;
        hint_range_kernel<<<dim3(g.B), 1024, 0, ctx->stream>>>(gmap, (size_t)g.H * g.W, range);
        VPPX_CHECK_LAUNCH();
        k.range = range;
    }
    if (p.method == VPPX_METHOD_MAXDIST) {
        if (p.wsize_agg_x < 1 || p.wsize_agg_y < 1) { vppx_set_error("wsize_agg must be >= 1"); return VPPX_E_INVALID_ARG; }
        if ((rc = ws_get(ctx, WS_HINT_X, npx, &k.rng))) return rc;
        k.rnd = nullptr;
        k.rnd_cap = 0;
        k.uniform = 1; // no random draws: keeps the draw bookkeeping of compact_kernel trivial
        if (k.use_dist) compact_kernel<true><<<dim3((g.H + 3) / 4, g.B), 256, 0, ctx->stream>>>(k);
        else compact_kernel<false><<<dim3((g.H + 3) / 4, g.B), 256, 0, ctx->stream>>>(k);
        VPPX_CHECK_LAUNCH();
        rowscan_kernel<<<dim3(g.B), 256, 0, ctx->stream>>>(k, (long long *)n_hints_dev);
        VPPX_CHECK_LAUNCH();
        stage_mark(ctx, ST_VPP_COMPACT);
        MdK m;
        m.k = k;
        m.k.uniform = p.uniform_color != 0;
        m.n_agg_x = (p.wsize_agg_x - 1) / 2;
        m.n_agg_y = (p.wsize_agg_y - 1) / 2;
        // rows one hint row can touch: patch radius + vertical half window of the colour search
        const int rad = k.n + m.n_agg_y;
        const size_t lds = (size_t)2 * (2 * rad + 1) * g.W;
        const size_t lds_wave = (size_t)2 * MD_RING * g.W;
        const int md = ctx->knobs.maxdist; // VPPX_VARIANT maxdist_lds (1) / maxdist_global (2): the one-wave-per-chain kernels
        if (md == 0 && k.direction && 2 * rad + 2 <= MD_RING && lds_wave <= 132 * 1024) {
            // row wavefront: MD_NW waves per chain
            static bool attr_set[VPPX_MAX_DEVICES] = {};
            if (!attr_set[ctx->device & (VPPX_MAX_DEVICES - 1)]) {
                VPPX_HIP(hipFuncSetAttribute((const void *)maxdist_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024));
                attr_set[ctx->device & (VPPX_MAX_DEVICES - 1)] = true;
            }
            maxdist_wave_kernel<<<dim3(g.B * g.C), 64 * MD_NW, lds_wave, ctx->stream>>>(m, rad);
        } else if (lds <= 60 * 1024 && md != 2) {
            maxdist_lds_kernel<<<dim3(g.B * g.C), 64, lds, ctx->stream>>>(m, rad);
        } else {
            maxdist_kernel<<<dim3(g.B * g.C), 64, 0, ctx->stream>>>(m);
        }
        VPPX_CHECK_LAUNCH();
        stage_mark(ctx, ST_VPP_APPLY);
        return 0;
    }
    const u64 cap = vpp_draws_upper_bound(p, g);
    u8 *rnd;
    if ((rc = ws_get(ctx, WS_RAND, (size_t)g.B * cap, &rnd))) return rc;
    k.rnd = rnd;
    k.rnd_cap = cap;
    const int nblk = (int)(cap / LG);
    const u32 *tab;
    if ((rc = ensure_rand_table(ctx, nblk, &tab))) return rc;
    if ((rc = ws_get(ctx, WS_LWORK, npx * 2, &k.lwork))) return rc; // (a pixel is deferred at most once)
    if ((rc = ws_get(ctx, WS_LWORK_CNT, (size_t)4, &k.lwork_cnt))) return rc;

    if (k.use_dist) compact_kernel<true><<<dim3((g.H + 3) / 4, g.B), 256, 0, ctx->stream>>>(k);
        else compact_kernel<false><<<dim3((g.H + 3) / 4, g.B), 256, 0, ctx->stream>>>(k);
    VPPX_CHECK_LAUNCH();
    rowscan_kernel<<<dim3(g.B), 256, 0, ctx->stream>>>(k, (long long *)n_hints_dev);
    VPPX_CHECK_LAUNCH();
    stage_mark(ctx, ST_VPP_COMPACT);
    rand_kernel<<<dim3((nblk + 63) / 64, g.B), 64, 0, ctx->stream>>>(rnd, cap, k.frame_tot, seeds_dev, p.seed,
                                                                      qoff_for(p.rand_offset), tab, nblk);
    VPPX_CHECK_LAUNCH();
    stage_mark(ctx, ST_VPP_RAND);
    const int npairs = ((g.W + 255) / 256) * g.H; // (apply_l_wide_kernel: a bounded grid per frame, strided)
    dim3 grid((unsigned)(npairs < 1024 ? npairs : 1024), g.B);
    if ((rc = ws_get(ctx, WS_RCNT, npx, &k.rcnt))) return rc;
    if ((rc = ws_get(ctx, WS_RLIST, npx * RLCAP, &k.rlist))) return rc;

    // The L side reads R only for occluded hints (pyx:114-122).  Without an occlusion mask the two sides are
    // independent: the (latency-bound) L kernels then run on the side stream next to the R list build + replay.
    // With a mask the L side replays R sub-chains from the pixels' hint lists and the ORIGINAL right image: when the
    // caller kept that image (fused path: the patterned pair is a copy) the fork happens after the list build.
    const bool split_early = (occ == nullptr) && ctx->stream2 != nullptr;
    const bool split_late = !split_early && r_orig != nullptr && r_orig != r && ctx->stream2 != nullptr;
    const bool split = split_early || split_late;
    // pixels with an occluded hint in their window: second pass (at every batch size: for one frame 12 + 25 us against 58
    // in one pass -- the first pass is a tenth of the code and runs with twice the waves per SIMD)
    const bool two_pass = occ != nullptr && !k.discard && k.n <= 3;
    if (!two_pass) k.lwork = nullptr; // (rowscan_kernel has zeroed the work-list counter)
    auto launch_l_hint = [&](hipStream_t st) {
        if (k.n > 3) return; // (apply_l_wide_kernel below)
        // rows per block: four, so that the touched pixels fill whole rounds of the block's replay; one when a call has too few
        // rows to give every CU a block otherwise.  The kernel has no code for occluded hints at all: it defers them
        // (with a mask that does not discard them: two_pass), and otherwise there are none to act on.
        const bool r4 = (long long)g.B * g.H >= 4 * 1024 && ctx->front_lds_budget >= 18 * 1024; // (17 KB of LDS against 4.5)
        const dim3 bg((unsigned)((g.W + 1023) / 1024), (unsigned)(r4 ? (g.H + 3) / 4 : g.H), (unsigned)g.B);
#define LB(NW) do { if (r4) apply_l_bits_kernel<NW, 4><<<bg, 256, 0, st>>>(k); else apply_l_bits_kernel<NW, 1><<<bg, 256, 0, st>>>(k); } while (0)
        switch (k.n) {
        case 0: LB(1); break;
        case 1: LB(3); break;
        case 2: LB(5); break;
        default: LB(7); break;
        }
#undef LB
    };
    auto launch_l_heavy = [&](hipStream_t st) {
        const dim3 hg((unsigned)(g.B * 512 < 32768 ? g.B * 512 : 32768)); // (about one deferred pixel per pair of lanes at 3 % hints: the replays are latency chains)
        switch (k.n) {
        case 0: apply_l_heavy_kernel<1><<<hg, 64, 0, st>>>(k); break;
        case 1: apply_l_heavy_kernel<3><<<hg, 64, 0, st>>>(k); break;
        case 2: apply_l_heavy_kernel<5><<<hg, 64, 0, st>>>(k); break;
        default: apply_l_heavy_kernel<7><<<hg, 64, 0, st>>>(k); break;
        }
    };
    // R side: r_rows_kernel, a block per row of up to 2048 columns (45 KB of LDS with 16-bit list entries), per part of at
    // most 1024 columns (39 KB with 32-bit entries) of a wider one.  (Parts of 512 columns, 11 KB, so that three blocks fit a CU next to the sum / WTA
    // kernel: 135 -> 179 us per 16 frames alone, 312 -> 611 us in the step -- every part walks all the hints of its rows.)
    const bool wide_ids = g.W > 2048;
    int rr_nseg = wide_ids ? (g.W + 1023) / 1024 : 1;
    if (r_rows_lds((g.W + rr_nseg - 1) / rr_nseg, wide_ids) > ctx->front_lds_budget) { // parts that fit next to the sum / WTA kernel
        const int per_col = 4 + RLCAP * (wide_ids ? 4 : 2) + 4;
        const int sw_max = (int)((ctx->front_lds_budget > 2048 ? ctx->front_lds_budget - 512 : 1536) / per_col);
        rr_nseg = (g.W + sw_max - 1) / sw_max;
    }
    const int rr_sw = (g.W + rr_nseg - 1) / rr_nseg;
    const size_t rows_lds = r_rows_lds(rr_sw, wide_ids);
    const bool need_lists = occ != nullptr && !k.discard; // some L kernel replays R sub-chains (r_chain)
    if (!need_lists) k.rcnt = nullptr;
    auto launch_r_rows = [&](int mode) {
        if (wide_ids) r_rows_kernel<u32, 16, RR_NT><<<dim3(g.H * rr_nseg, g.B), RR_NT, rows_lds, ctx->stream>>>(k, mode, rr_sw);
        else r_rows_kernel<unsigned short, 11, RR_NT><<<dim3(g.H * rr_nseg, g.B), RR_NT, rows_lds, ctx->stream>>>(k, mode, rr_sw);
    };
    hipStream_t ls = ctx->stream;
    if (split) { // the L side on the side stream from here on
        VPPX_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        VPPX_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        ls = ctx->stream2;
    }
    if (!need_lists) {
        // no L kernel looks at R: the two sides are independent
        launch_l_hint(ls);
        VPPX_CHECK_LAUNCH();
        if (k.n > 3) apply_l_wide_kernel<<<grid, 256, 0, ls>>>(k);
        VPPX_CHECK_LAUNCH();
        launch_r_rows(1);
        VPPX_CHECK_LAUNCH();
    } else if (split_late) {
        // the caller kept the original right image: the R side runs right away (building the lists as it goes), the first
        // L pass next to it; only the passes that replay R sub-chains wait for the lists
        if (two_pass) {
            launch_l_hint(ls);
            VPPX_CHECK_LAUNCH();
        }
        launch_r_rows(3);
        VPPX_CHECK_LAUNCH();
        VPPX_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        VPPX_HIP(hipStreamWaitEvent(ls, ctx->ev_fork, 0));
        if (two_pass) launch_l_heavy(ls);
        else apply_l_wide_kernel<<<grid, 256, 0, ls>>>(k); // (n > 3)
        VPPX_CHECK_LAUNCH();
    } else {
        // R is patterned in place: every L pass that reads it comes first
        launch_r_rows(2);
        VPPX_CHECK_LAUNCH();
        launch_l_hint(ls);
        VPPX_CHECK_LAUNCH();
        if (two_pass) launch_l_heavy(ls);
        if (k.n > 3) apply_l_wide_kernel<<<grid, 256, 0, ls>>>(k);
        VPPX_CHECK_LAUNCH();
        launch_r_rows(1);
        VPPX_CHECK_LAUNCH();
    }
    if (split) {
        VPPX_HIP(hipEventRecord(ctx->ev_join, ctx->stream2));
        VPPX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    stage_mark(ctx, ST_VPP_APPLY);
    return 0;
}

// ---------------------------------------------------------------------------------------
// _bilateral_filling (vpp_standalone.py:372-394).  The reference rasters over the hints and lets
// each stamp its disparity on the window pixels where its weight beats the best so far
// (`if cmap[n] < weight`, cmap float32, weight float64).  Gather form: one thread per pixel
// replays, in raster order of the hint positions, the hints whose window covers it, with the
// same float32 store / float64 compare.  gray = BGR2GRAY of the (RGB) left image, as the
// wrapper computes it (vpp_standalone.py:415), or the single channel.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gray_ctx_kernel(const u8 *__restrict__ img, u8 *__restrict__ gray, size_t npix, int C)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const u8 *s = img + i * C;
    gray[i] = C == 3 ? (u8)((s[2] * 9798u + s[1] * 19235u + s[0] * 3735u + 16384u) >> 15) : s[0];
}

__global__ void __launch_bounds__(256) bilateral_fill_kernel(const float *__restrict__ g, const u8 *__restrict__ gray,
                                                             float *__restrict__ out, int H, int W, int n, double inv2oxy,
                                                             double inv2oi, double th)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const size_t base = (size_t)f * H * W;
    const int In = gray[base + (size_t)y * W + x];
    float cmap = 0.f;
    float aug = g[base + (size_t)y * W + x];
    for (int hy = max(0, y - n); hy <= min(H - 1, y + n); hy++)
        for (int hx = max(0, x - n); hx <= min(W - 1, x + n); hx++) {
            const float d_ref = g[base + (size_t)hy * W + hx];
            if (!(d_ref > 0)) continue;
            const int yw = y - hy, xw = x - hx;
            const long long di = (long long)In - (long long)gray[base + (size_t)hy * W + hx];
            const double a = __dadd_rn(__ddiv_rn((double)(yw * yw + xw * xw), inv2oxy), __ddiv_rn((double)(di * di), inv2oi));
            const double wgt = exp(-a);
            if ((double)cmap < wgt) {
                cmap = (float)wgt;
                aug = d_ref;
            }
        }
    out[base + (size_t)y * W + x] = ((double)cmap > th) ? aug : 0.f;
}

int vpp_launch_bilateral_fill(vppx_ctx *ctx, const VppxVppParams &p, const VppGeom &g, const u8 *left, const float *gmap,
                              float *filled_out)
{
    int rc;
    u8 *gray;
    const size_t npix = (size_t)g.B * g.H * g.W;
    if ((rc = ws_get(ctx, WS_GRAY_CTX, npix, &gray))) return rc;
    gray_ctx_kernel<<<dim3((unsigned)((npix + 255) / 256)), 256, 0, ctx->stream>>>(left, gray, npix, g.C);
    VPPX_CHECK_LAUNCH();
    // the reference divides by 2*o^2 (vpp_standalone.py:386); a multiplication by the reciprocal
    // would round differently, so the kernel receives the divisors
    dim3 grid((g.W + 255) / 256, g.H, g.B);
    bilateral_fill_kernel<<<grid, 256, 0, ctx->stream>>>(gmap, gray, filled_out, g.H, g.W, (p.wsize - 1) / 2,
                                                          2.0 * (p.bilateral_o_xy * p.bilateral_o_xy),
                                                          2.0 * (p.bilateral_o_i * p.bilateral_o_i), p.bilateral_th);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int vpp_launch_rand_stream(vppx_ctx *ctx, u32 seed, u64 offset, int64_t n, int32_t *out_dev)
{
    if (n <= 0) return 0;
    const int nblk = (int)((n + LG - 1) / LG);
    const u32 *tab;
    int rc;
    if ((rc = ensure_rand_table(ctx, nblk, &tab))) return rc;
    rand_full_kernel<<<dim3((nblk + 63) / 64), 64, 0, ctx->stream>>>(out_dev, (long long)n, seed, qoff_for(offset), tab, nblk);
    VPPX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// occlusion heuristic (filter.py:246-292) -> conf map (the g_occ of test.py:154)
// ---------------------------------------------------------------------------------------
// (four pixels per thread where rows are 16-byte aligned: a quarter of the waves for the same loads)
__global__ void __launch_bounds__(256) occ_warp4_kernel(const float4 *__restrict__ dmap, int *__restrict__ omap_bits, int H, int W)
{
    const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (4 * x4 >= W) return;
    const size_t row = ((size_t)f * H + y) * W;
    const float4 q = dmap[row / 4 + x4];
    const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (v[j] > 0) {
            const int xd = 4 * x4 + j - (int)rintf(v[j]);
            if (0 <= xd && xd <= W - 1) atomicMax(&omap_bits[row + xd], __float_as_int(v[j])); // positive floats order as ints
        }
    }
}
__global__ void __launch_bounds__(256) occ_warp_kernel(const float *__restrict__ dmap, int *__restrict__ omap_bits, int H, int W)
{
    // left_warp (filter.py:8-48): omap[y, x-round(d)] keeps the max on collision
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const size_t row = ((size_t)f * H + y) * W;
    const float v = dmap[row + x];
    if (v > 0) {
        const int xd = x - (int)rintf(v);
        if (0 <= xd && xd <= W - 1) atomicMax(&omap_bits[row + xd], __float_as_int(v)); // positive floats order as ints
    }
}
// weighted_conf (filter.py:115-164) + filter (:168-194) + conf_unwarp (:82-112) in one pass over the HINTS.
// A valid pixel of omap at x_o holds the value v of exactly one hint, the one at x_o + round(v) (left_warp keeps the
// maximum of what lands on x_o, and a hint that lands there with that value sits at that column); conf_unwarp sends the
// pixel's confidence back to x_o + round(v): to the column of the very hint it came from.  So no two pixels of omap
// unwarp to the same place, the result at a hint's position depends on that hint alone, and the unwarp needs neither
// an image of winners nor atomics:  out[y, x] = conf of omap[y, x - round(v)] if the hint at x owns that pixel
// (omap == v) and the filter keeps it, 1 everywhere else (the initial value of conf_unwarp, :101).
// The confidence in gather form: pixel n = (y, xd) of omap is rejected if some valid centre c with n in c's window is
// nearer (larger) by more than the weighted distance.  Owners are sparse (the hints, a few per cent).  A block owns 1024
// pixels of a row: it loads their hints and the pixels of omap they land on (four per thread, in flight together),
// collects the owners in LDS, and its waves then take them four at a time and test all window positions of the four at
// once, one per lane and NJ per owner; the loads are unconditional at clamped coordinates (hipcc does not speculate
// loads: a conditional one costs a branch and a full wait each) so that all 4 NJ are in flight together.
// PT: pixels per thread (4: a block owns 1024 pixels and 11 KB of LDS; 2: 512 pixels, 5.6 KB)
template <int NJ, int PT>
__global__ void __launch_bounds__(256) occ_test_kernel(const float *__restrict__ hints, const float *__restrict__ omap,
                                                       u8 *__restrict__ out, int *__restrict__ tmp, int H, int W, int rx,
                                                       int ry, double l, double g, double th, double th_filter)
{
    __shared__ int s_xd[256 * PT];
    __shared__ float s_v[256 * PT];
    __shared__ unsigned short s_px[256 * PT];
    __shared__ u8 s_cf[256 * PT];
    __shared__ int s_total;
    const int x0 = blockIdx.x * (256 * PT);
    const int y = blockIdx.y, f = blockIdx.z;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *om = omap + (size_t)f * H * W;
    const size_t row = ((size_t)f * H + y) * W;
    if (t == 0) s_total = 0;
    float v[PT], at[PT];
    int xd[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int x = x0 + 256 * u + t;
        v[u] = hints[row + min(x, W - 1)];
        if (x >= W) v[u] = 0.0f;
        s_cf[256 * u + t] = 0;
    }
#pragma unroll
    for (int u = 0; u < PT; u++) {
        xd[u] = x0 + 256 * u + t - (int)rintf(v[u]);
        at[u] = om[(size_t)y * W + min(max(xd[u], 0), W - 1)];
    }
    __syncthreads();
    bool owner[PT];
#pragma unroll
    for (int u = 0; u < PT; u++) {
        owner[u] = v[u] > 0 && 0 <= xd[u] && xd[u] <= W - 1 && at[u] == v[u];
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(owner[u]);
        if (bal) {
            const int leader = __builtin_ctzll(bal);
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_total, (int)__popcll(bal));
            base = __builtin_amdgcn_readlane(base, leader);
            if (owner[u]) {
                const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                s_xd[slot] = xd[u];
                s_v[slot] = v[u];
                s_px[slot] = (unsigned short)(256 * u + t);
            }
        }
    }
    __syncthreads();
    const int total = s_total;
    const int nyw = 2 * ry + 2, nwin = (2 * rx + 1) * nyw; // centre = (y - yw, xd - xw), xw in [-rx,rx], yw in [-ry-1,ry] (:149)
    if (NJ > 0) {
        // a lane's window positions (lane, lane + 64, ...) and their distance weights do not depend on the pixel under
        // test: worked out once (the integer divisions and four of the five float64 operations of every test)
        int xwj[NJ > 0 ? NJ : 1], ycj[NJ > 0 ? NJ : 1];
        double wgj[NJ > 0 ? NJ : 1];
        bool inj[NJ > 0 ? NJ : 1];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int kpos = lane + 64 * j;
            xwj[j] = kpos / nyw - rx;
            const int yw = kpos % nyw - ry - 1;
            wgj[j] = __dmul_rn(l, __dadd_rn(__dmul_rn(g, (double)abs(xwj[j])), __dmul_rn(__dsub_rn(1.0, g), (double)abs(yw))));
            const int yc = y - yw;
            inj[j] = kpos < nwin && yc >= 0 && yc <= H - 1;
            ycj[j] = min(max(yc, 0), H - 1);
        }
        for (int e0 = 4 * wv; e0 < total; e0 += 16) {
            int xs[4];
            float ds[4], c[4][NJ > 0 ? NJ : 1];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = e0 + i < total ? e0 + i : e0;
                xs[i] = s_xd[e];
                ds[i] = s_v[e];
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const float *r = om + (size_t)ycj[j] * W;
#pragma unroll
                for (int i = 0; i < 4; i++) c[i][j] = r[min(max(xs[i] - xwj[j], 0), W - 1)];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                bool h = false;
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const int xc = xs[i] - xwj[j];
                    if (inj[j] && xc >= 0 && xc <= W - 1 && c[i][j] > 0 && ds[i] < c[i][j])
                        h = h || (__dsub_rn((double)__fsub_rn(c[i][j], ds[i]), wgj[j]) > th);
                }
                if (__builtin_amdgcn_ballot_w64(h) != 0 && lane == 0 && e0 + i < total) s_cf[s_px[e0 + i]] = 1;
            }
        }
    } else {
        for (int e = wv; e < total; e += 4) { // windows of more than 256 positions: a wave per owner
            const int xs = s_xd[e];
            const float ds = s_v[e];
            bool hit = false;
            for (int kpos = lane; kpos < nwin; kpos += 64) {
                const int xw = kpos / nyw - rx, yw = kpos % nyw - ry - 1;
                const int yc = y - yw, xc = xs - xw;
                if (yc < 0 || yc > H - 1 || xc < 0 || xc > W - 1) continue;
                const float dc = om[(size_t)yc * W + xc];
                if (dc > 0 && ds < dc) {
                    const double tt = __dsub_rn((double)__fsub_rn(dc, ds),
                                                __dmul_rn(l, __dadd_rn(__dmul_rn(g, (double)abs(xw)),
                                                                       __dmul_rn(__dsub_rn(1.0, g), (double)abs(yw)))));
                    hit = hit || (tt > th);
                }
            }
            if (__builtin_amdgcn_ballot_w64(hit) != 0 && lane == 0) s_cf[s_px[e]] = 1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PT; u++) {
        const int x = x0 + 256 * u + t;
        if (x >= W) continue;
        const int cf = s_cf[256 * u + t];
        const bool kept = owner[u] && !((double)cf > th_filter); // filter (:168-194) drops a pixel of omap whose confidence is above the threshold
        out[row + x] = kept ? (u8)cf : (u8)1;
        if (tmp) tmp[row + x] = kept ? ((xd[u] << 1) | cf) : -1; // where the value of element [0] sits in omap (occ_dmap_kernel)
    }
}

// Element [0] of occlusion_heuristic (filter.py:283-292): left_unwarp of the filtered omap (:51-79, last writer in raster
// order wins: the same winner conf_unwarp has, kept in tmp) followed by interpolate_disparity(dmap, 3) (:197-243).  With
// n = 1 that pass only ever changes a zero pixel whose two flat-memory neighbours are positive and less than 1 apart (the
// neighbours it rewrites get their own values back), and no pixel it reads has been changed before it is read, so the
// sequential pass equals this gather.  The reference indexes dmap[y, x +- 1] without a bounds test (SURVEY C-8): x - 1 = -1
// wraps to column W - 1 of the same row, x + 1 = W is the first pixel of the next row; past the end of the frame is
// undefined there and reads as 0 here (like the oracle).
__global__ void __launch_bounds__(256) occ_dmap_kernel(const int *__restrict__ tmp, const float *__restrict__ omap,
                                                       float *__restrict__ dmap_out, int H, int W)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const size_t fb = (size_t)f * H * W;
    auto unwarped = [&](int yy, int xx) -> float {
        const size_t row = fb + (size_t)yy * W;
        const int t = tmp[row + xx];
        return t < 0 ? 0.0f : omap[row + (t >> 1)];
    };
    float u = unwarped(y, x);
    if (u == 0.0f) {
        const float nl = x > 0 ? unwarped(y, x - 1) : unwarped(y, W - 1);
        const float nr = x < W - 1 ? unwarped(y, x + 1) : (y < H - 1 ? unwarped(y + 1, 0) : 0.0f);
        if (nl > 0 && nr > 0 && fabs(__dsub_rn((double)nl, (double)nr)) < 1.0) {
            const double m = __ddiv_rn(__dsub_rn((double)nr, (double)nl), 2.0);
            const double q = __dsub_rn((double)nl, __dmul_rn(m, -1.0));
            u = (float)__dadd_rn(__dmul_rn(m, 0.0), q);
        }
    }
    dmap_out[fb + (size_t)y * W + x] = u;
}

int occ_launch(vppx_ctx *ctx, int B, int H, int W, const float *hints, int rx, int ry, double l, double g,
               double th_conf, double th_filter, float *omap, u8 *conf_out, float *dmap_out)
{
    const size_t n = (size_t)B * H * W;
    // (next to a sum / WTA kernel that leaves less than 12 KB of LDS -- D = 256 -- the test kernel takes the 5.6 KB variant)
    const bool small = ctx->front_lds_budget < 12 * 1024;
    dim3 grid((W + 255) / 256, H, B), grid_t((W + (small ? 511 : 1023)) / (small ? 512 : 1024), H, B);
    int *tmp = nullptr;
    if (dmap_out) {
        int rc = ws_get(ctx, WS_OCC_TMP, n, &tmp);
        if (rc) return rc;
    }
    VPPX_HIP(hipMemsetAsync(omap, 0, n * sizeof(float), ctx->stream));
    if (W % 4 == 0 && ((size_t)hints & 15) == 0)
        occ_warp4_kernel<<<dim3((W / 4 + 255) / 256, H, B), 256, 0, ctx->stream>>>((const float4 *)hints, (int *)omap, H, W);
    else
        occ_warp_kernel<<<grid, 256, 0, ctx->stream>>>(hints, (int *)omap, H, W);
    VPPX_CHECK_LAUNCH();
    rx /= 2, ry /= 2; // filter.py:142-143
    const long long nwin = (2LL * rx + 1) * (2LL * ry + 2);
#define OCC_TEST(NJ) do { if (small) occ_test_kernel<NJ, 2><<<grid_t, 256, 0, ctx->stream>>>(hints, omap, conf_out, tmp, H, W, rx, ry, l, g, th_conf, th_filter); \
                          else occ_test_kernel<NJ, 4><<<grid_t, 256, 0, ctx->stream>>>(hints, omap, conf_out, tmp, H, W, rx, ry, l, g, th_conf, th_filter); } while (0)
    if (nwin <= 64) OCC_TEST(1);
    else if (nwin <= 128) OCC_TEST(2);
    else if (nwin <= 192) OCC_TEST(3);
    else if (nwin <= 256) OCC_TEST(4);
    else OCC_TEST(0);
#undef OCC_TEST
    VPPX_CHECK_LAUNCH();
    if (dmap_out) {
        occ_dmap_kernel<<<grid, 256, 0, ctx->stream>>>(tmp, omap, dmap_out, H, W);
        VPPX_CHECK_LAUNCH();
    }
    return 0;
}
