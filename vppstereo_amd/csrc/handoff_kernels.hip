// handoff_kernels.hip -- gfx950 kernels for the rows either side of the hot path:
//   * PSMNet concat volume + hint modulation          models/psmnet/psmnet.py:157-197
//   * RAFT-Stereo all-pairs correlation modulation    models/raft_stereo/corr.py:160-178
//   * disparity file payload decoders                 dataloaders/frame_utils.py:34-69
// All of them are HBM streaming kernels (write-bound volume fill; sparse in-place row scaling).
#include "vppx_internal.h"

#include <math.h>

// ---------------------------------------------------------------------------------------
// nearest sub-sampling of the hint pair to feature resolution (F.upsample(size=[H//4, W//4],
// mode='nearest'): src = min((int)floorf(dst * (float)in / out), in - 1)), then
//   psmnet: h = hints * valid / 4                     (psmnet.py:186)
//   raft  : h = x * valid - hints * valid / 4         (corr.py:168-169)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) subsample_hints_kernel(const float *__restrict__ hints, const float *__restrict__ valid,
                                                              int H, int W, int H4, int W4, float sy, float sx,
                                                              float *__restrict__ hs, float *__restrict__ vs, int raft)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x >= W4) return;
    int yy = (int)floorf(__fmul_rn((float)y, sy));
    int xx = (int)floorf(__fmul_rn((float)x, sx));
    yy = yy < H - 1 ? yy : H - 1;
    xx = xx < W - 1 ? xx : W - 1;
    const size_t si = ((size_t)b * H + yy) * W + xx;
    const float v = valid[si];
    float h = __fdiv_rn(__fmul_rn(hints[si], v), 4.0f);
    if (raft) h = __fsub_rn(__fmul_rn((float)x, v), h);
    const size_t di = ((size_t)b * H4 + y) * W4 + x;
    hs[di] = h;
    vs[di] = v;
}

// (1 - v) + (v * height) * exp(-(d - h)^2 / two_w2)     (psmnet.py:197, corr.py:178)
__device__ __forceinline__ float modulation(float v, float h, float d, float two_w2)
{
    const float diff = __fsub_rn(d, h);
    const float e = expf(__fdiv_rn(-__fmul_rn(diff, diff), two_w2));
    return __fadd_rn(__fsub_rn(1.0f, v), __fmul_rn(__fmul_rn(v, 10.0f), e));
}

// cost[b, c, i, y, x]     = x >= i ? fea_l[b, c, y, x]     : 0
// cost[b, C + c, i, y, x] = x >= i ? fea_r[b, c, y, x - i] : 0      (psmnet.py:157-166), times the
// modulation of (b, i, y, x) when hints are given.  The modulation does not depend on the channel:
// one thread owns VEC consecutive x of one (b, i, y), evaluates it once and streams the 2C channels.
template <int VEC>
__global__ void __launch_bounds__(256) psm_volume_kernel(const float *__restrict__ fl, const float *__restrict__ fr,
                                                         const float *__restrict__ hs, const float *__restrict__ vs,
                                                         int C, int D4, int H4, int W4, float *__restrict__ cost)
{
    const int wv = (W4 + VEC - 1) / VEC;                  // vectors per row (VEC divides W4 when VEC > 1)
    const int pv = blockIdx.x * blockDim.x + threadIdx.x; // flat vector index in the (y, x) plane
    if (pv >= wv * H4) return;
    const int y = pv / wv, x0 = (pv % wv) * VEC;
    const int i = blockIdx.y, b = blockIdx.z;
    float mod[VEC];
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; j++) {
        mod[j] = 1.0f;
        if (hs && x0 + j < W4) {
            const size_t pi = ((size_t)b * H4 + y) * W4 + x0 + j;
            const float v = vs[pi];
            if (v != 0.0f) { // v == 0: (1 - 0) + 0 * 10 * e = 1 exactly
                mod[j] = modulation(v, hs[pi], (float)i, 0.125f); // 2 * (1/4)^2  (psmnet.py:187)
                any = true;
            }
        }
    }
    const size_t plane = (size_t)H4 * W4;
    const float *sl = fl + (size_t)b * C * plane + (size_t)y * W4;
    const float *sr = fr + (size_t)b * C * plane + (size_t)y * W4;
    float *dst = cost + (((size_t)b * 2 * C * D4 + i) * H4 + y) * W4 + x0;
    for (int ch = 0; ch < 2 * C; ch++) {
        const bool right = ch >= C;
        const float *src = right ? sr + (size_t)(ch - C) * plane : sl + (size_t)ch * plane;
        float out[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            const int x = x0 + j;
            float val = 0.f;
            if (x < W4 && x >= i) val = src[right ? x - i : x];
            out[j] = any ? __fmul_rn(val, mod[j]) : val;
        }
        float *d = dst + (size_t)ch * D4 * plane;
        if (VEC == 4) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 v4 = {out[0], out[1], out[2], out[3]};
            __builtin_nontemporal_store(v4, (f4 *)d); // written once, read by the next layer much later
        } else {
#pragma unroll
            for (int j = 0; j < VEC; j++)
                if (x0 + j < W4) d[j] = out[j];
        }
    }
}

// corr[b, y, x2, 0, k] *= modulation(b, y, x2; k); rows with valid == 0 are multiplied by exactly 1
// in the reference, i.e. untouched: one 64-lane wave per (b, y, x2) row, skipping those.
__global__ void __launch_bounds__(256) raft_modulate_kernel(float *__restrict__ corr, const float *__restrict__ hs,
                                                            const float *__restrict__ vs, size_t nrows, int W3)
{
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float v = vs[row];
    if (v == 0.0f) return;
    const float h = hs[row];
    float *p = corr + row * (size_t)W3;
    for (int k = threadIdx.x & 63; k < W3; k += 64) p[k] = __fmul_rn(p[k], modulation(v, h, (float)k, 2.0f));
}

__global__ void __launch_bounds__(256) kitti_decode_kernel(const uint16_t *__restrict__ png, size_t n, float *__restrict__ disp,
                                                           u8 *__restrict__ valid)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = __fdiv_rn((float)png[i], 256.0f); // exact for every u16 (frame_utils.py:67)
    disp[i] = d;
    if (valid) valid[i] = d > 0.0f ? 1 : 0;
}

__global__ void __launch_bounds__(256) pfm_decode_kernel(const u32 *__restrict__ raw, int H, int rowwords, int little,
                                                         u32 *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= rowwords) return;
    u32 w = raw[(size_t)(H - 1 - y) * rowwords + x]; // np.flipud (frame_utils.py:63)
    if (!little) w = __builtin_bswap32(w);
    out[(size_t)y * rowwords + x] = w;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static int launch_subsample(vppx_ctx *ctx, const float *hints, const float *valid, int B, int H, int W, int H4, int W4,
                            float **hs, float **vs, int raft)
{
    int rc;
    if (H / 4 != H4 || W / 4 != W4) {
        vppx_set_error("hints are %dx%d: sub-sampled by 4 that is %dx%d, but the features are %dx%d", H, W, H / 4, W / 4, H4, W4);
        return VPPX_E_INVALID_ARG;
    }
    if ((rc = ws_get(ctx, WS_HANDOFF_H, (size_t)B * H4 * W4, hs))) return rc;
    if ((rc = ws_get(ctx, WS_HANDOFF_V, (size_t)B * H4 * W4, vs))) return rc;
    volatile float fy = (float)H / (float)H4, fx = (float)W / (float)W4; // float32 scale like ATen
    subsample_hints_kernel<<<dim3((W4 + 255) / 256, H4, B), 256, 0, ctx->stream>>>(hints, valid, H, W, H4, W4, fy, fx, *hs, *vs,
                                                                                   raft);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int handoff_psmnet_cost_volume(vppx_ctx *ctx, const float *fl, const float *fr, const float *hints, const float *valid, int B,
                               int C, int H4, int W4, int H, int W, int maxdisp, float *cost)
{
    int rc;
    const int D4 = maxdisp / 4;
    float *hs = nullptr, *vs = nullptr;
    if (hints && (rc = launch_subsample(ctx, hints, valid, B, H, W, H4, W4, &hs, &vs, 0))) return rc;
    if (D4 > 65535 || B > 65535) {
        vppx_set_error("psmnet cost volume: maxdisp/4 and B must be <= 65535");
        return VPPX_E_UNSUPPORTED;
    }
    if (W4 % 4 == 0 && ((uintptr_t)cost & 15) == 0) {
        psm_volume_kernel<4><<<dim3((W4 / 4 * H4 + 255) / 256, D4, B), 256, 0, ctx->stream>>>(fl, fr, hs, vs, C, D4, H4, W4, cost);
    } else {
        psm_volume_kernel<1><<<dim3((W4 * H4 + 255) / 256, D4, B), 256, 0, ctx->stream>>>(fl, fr, hs, vs, C, D4, H4, W4, cost);
    }
    VPPX_CHECK_LAUNCH();
    return 0;
}

int handoff_raft_corr_modulate(vppx_ctx *ctx, float *corr, const float *hints, const float *valid, int B, int H4, int W2, int W3,
                               int H, int W)
{
    int rc;
    float *hs, *vs;
    if ((rc = launch_subsample(ctx, hints, valid, B, H, W, H4, W2, &hs, &vs, 1))) return rc;
    const size_t nrows = (size_t)B * H4 * W2;
    raft_modulate_kernel<<<dim3((unsigned)((nrows + 3) / 4)), 256, 0, ctx->stream>>>(corr, hs, vs, nrows, W3);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int handoff_kitti_decode(vppx_ctx *ctx, const uint16_t *png, size_t n, float *disp, u8 *valid)
{
    kitti_decode_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, ctx->stream>>>(png, n, disp, valid);
    VPPX_CHECK_LAUNCH();
    return 0;
}

int handoff_pfm_decode(vppx_ctx *ctx, const u8 *raw, int H, int W, int channels, int little, float *out)
{
    const int rowwords = W * channels;
    pfm_decode_kernel<<<dim3((rowwords + 255) / 256, H), 256, 0, ctx->stream>>>((const u32 *)raw, H, rowwords, little, (u32 *)out);
    VPPX_CHECK_LAUNCH();
    return 0;
}
