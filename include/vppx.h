/*
 * vppx.h -- C-ABI of libvppx.so: MI355X-native (gfx950) virtual-pattern-projection +
 * rSGM hot path.  Drop-in boundary for the reference's native entry points:
 *
 *   reference native (file:line)                                   replaced by
 *   ------------------------------------------------------------   ---------------------------
 *   vpp_core_opt.init_rand              vpp_core_opt.pyx:33-35      vppx_srand
 *   vpp_core_opt.virtual_projection_scan_rnd       .pyx:53-54       vppx_virtual_projection_scan_rnd
 *   vpp_core_opt.virtual_projection_scan_max_dist  .pyx:133-134     vppx_virtual_projection_scan_max_dist
 *   numba twins + wrapper vpp()         vpp_standalone.py:243,14,396 vppx_vpp_host / vppx_vpp_dev
 *   pyrSGM.census5x5_SSE                call site rsgm.py:25        vppx_census5x5
 *   pyrSGM.costMeasureCensus5x5_xyd_SSE call site rsgm.py:44        vppx_cost_census5x5_xyd
 *   pyrSGM.aggregate_SSE                call site rsgm.py:61        vppx_aggregate
 *   pyrSGM.matchWTA_SSE                 call site rsgm.py:141       vppx_match_wta
 *   pyrSGM.subPixelRefine               call site rsgm.py:142       vppx_subpixel_refine
 *   pyrSGM.median3x3_SSE                call site rsgm.py:145,173   vppx_median3x3
 *   pyrSGM.matchWTARight_SSE            call site rsgm.py:170       vppx_match_wta_right
 *   compute_rsgm()                      rsgm.py:250-294             vppx_rsgm_host / vppx_rsgm_dev
 *   occlusion_heuristic()               filter.py:246-292           vppx_occlusion_heuristic_host/_dev
 *   test.py:158-225 (VPP -> rSGM per frame)                         vppx_vpp_rsgm_dev (fused, batched)
 *
 * Conventions
 *   - plain C, no torch types.  "_host" entry points take host pointers (what a ctypes /
 *     numpy caller has) and are synchronous; "_dev" entry points take device pointers,
 *     enqueue on the context's HIP stream and return without synchronising.
 *   - images are uint8, HWC, C-contiguous, batch outermost: [B,H,W,C]; hint / disparity
 *     maps are float32 [B,H,W]; masks uint8 [B,H,W].  Exactly the reference's layouts
 *     (vpp_core_opt.pyx:53, rsgm.py:42,60) with a leading batch dimension.
 *   - every function returns 0 on success or a negative VPPX_E_* code; the message of
 *     the last error on the calling thread is available from vppx_last_error().  Codes map
 *     one-to-one onto the reference's Python exceptions (rsgm.py:19-20,31-40,157-167).
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     VPPX_E_NO_DEVICE.
 */
#ifndef VPPX_H
#define VPPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPPX_VERSION 100

enum {
    VPPX_OK = 0,
    VPPX_E_INVALID_ARG = -1,     /* null pointer, bad shape, bad enum                       */
    VPPX_E_DMAX_MOD8 = -2,       /* "Invalid dmax (..): dmax % 8 != 0"      rsgm.py:31-32   */
    VPPX_E_DMAX_GT256 = -3,      /* "Invalid dmax (..): dmax > 256"         rsgm.py:34-35   */
    VPPX_E_UNIQUENESS = -4,      /* "Invalid uniqueness (..) in ]0,1]"      rsgm.py:166-167 */
    VPPX_E_WIDTH_MOD16 = -5,     /* "Invalid width (..): width % 16 != 0"   rsgm.py:19-20   */
    VPPX_E_METHOD = -6,          /* assert method in ["rnd","maxDistance"]  vpp_standalone.py:400 */
    VPPX_E_NO_DEVICE = -7,       /* no HIP device / HIP runtime error at init               */
    VPPX_E_HIP = -8,             /* HIP runtime error (message has the hipError string)     */
    VPPX_E_OOM = -9,             /* workspace allocation failed                             */
    VPPX_E_UNSUPPORTED = -10     /* valid in the reference but outside this build's limits  */
};

enum { VPPX_METHOD_RND = 0, VPPX_METHOD_MAXDIST = 1 };

/* Parameters of the VPP scans.  Field meaning = the same-named argument of
 * vpp_core_opt.pyx:53-54 / :133-134 and vpp_standalone.py:396 (defaults in brackets). */
typedef struct VppxVppParams {
    int32_t method;             /* VPPX_METHOD_*                 ["rnd"]                    */
    int32_t wsize;              /* patch size 2n+1               [3]                        */
    int32_t wsize_agg_x;        /* maxDistance window            [64]                       */
    int32_t wsize_agg_y;        /*                               [3]                        */
    int32_t direction;          /* 1 = left2right scan, 0 = r2l  [1]                        */
    int32_t uniform_color;      /*                               [0]                        */
    int32_t discard_occluded;   /*                               [0]                        */
    int32_t interpolate;        /*                               [1]                        */
    float c;                    /* blending                      [0.4]                      */
    float c_occ;                /*                               [0.0]                      */
    /* numba-twin extras (vpp_standalone.py:93-96,154,315-318,335) */
    int32_t use_distance_patch; /*                               [0]                        */
    int32_t use_bilateral_patch;/* gate on |g - filled_g| < 0.1  [0] (needs filled_g)       */
    double distance_gamma;      /*                               [0.3]                      */
    float dmin, dmax;           /* min / max positive hint of the frame (wrapper :410-411); ignored with per_frame_range */
    /* random stream (libc rand() in the reference, vpp_core_opt.pyx:33-35,93,102):
     * frame f of a batch draws from srand(seed + f) starting rand_offset draws in.        */
    uint32_t seed;              /*                               [1]                        */
    /* use_distance_patch in a batch: 1 = every frame's dmin / dmax are taken from ITS OWN hints on the device (min / max of the
     * positive values), which is what vpp() does per call (vpp_standalone.py:410-411); 0 = dmin / dmax above, one pair for the
     * whole batch.  A frame whose positive hints all have one value -- the reference's _get_patch_size_based_on_distance divides
     * by zero there -- gets the full patch size.                                    [0]     */
    uint32_t per_frame_range;
    uint64_t rand_offset;       /*                               [0]                        */
    /* _bilateral_filling (vpp_standalone.py:372-394, wrapper :419-420): when
     * use_bilateral_patch is set and no filled_g is passed, the library densifies the hints
     * itself from the un-patterned left image (BGR2GRAY of it, as the wrapper does at :415) */
    double bilateral_o_xy;      /*                               [2]                        */
    double bilateral_o_i;       /*                               [1]                        */
    double bilateral_th;        /*                               [0.001]                    */
} VppxVppParams;

/* Parameters of compute_rsgm (rsgm.py:250). */
typedef struct VppxRsgmParams {
    int32_t dmax;        /* [192]  multiple of 8, <= 256 */
    int32_t p1;          /* [11]   */
    int32_t p2min;       /* [17]   */
    float alpha;         /* [0.5]  */
    int32_t gamma;       /* [35]   */
    float uniqueness;    /* [0.95] */
    int32_t subpixel;    /* [1]  (test.py:71 passes False) */
    int32_t reserved0;
} VppxRsgmParams;

/* Parameters of occlusion_heuristic (filter.py:246; defaults in brackets). */
typedef struct VppxOccParams {
    int32_t rx, ry;      /* [9] [7]   */
    double l, g;         /* [2] [0.4375] */
    double th_conf;      /* [1]   */
    double th_filter;    /* [0.1] */
} VppxOccParams;

typedef struct vppx_ctx vppx_ctx; /* opaque: device, stream, workspace arena */

/* ---- library / context ------------------------------------------------------------ */
int vppx_version(void);
const char *vppx_last_error(void);
void vppx_vpp_params_default(VppxVppParams *p);
void vppx_rsgm_params_default(VppxRsgmParams *p);
void vppx_occ_params_default(VppxOccParams *p);

/* Create a context on HIP device `device` (-1 = current device).  One context per
 * process/GPU; not thread-safe (the reference is single-threaded: rsgm.py:44). */
int vppx_create(vppx_ctx **out, int device);
void vppx_destroy(vppx_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current side stream); NULL = the context's own (non-blocking) stream.
 * Binding the stream that is already bound costs nothing (no synchronisation). */
int vppx_set_stream(vppx_ctx *ctx, void *hip_stream);
/* Launch on the legacy default ("null") stream itself -- what torch's default stream is (its handle is 0, which
 * vppx_set_stream reads as "own stream").  All work is then ordered with the caller's default-stream work. */
int vppx_set_stream_legacy(vppx_ctx *ctx);

/* Cross-call pipelining for streams of batches through vppx_occ_vpp_rsgm_dev / vppx_vpp_rsgm_dev: with `on`, the front
 * stage of a call (occlusion heuristic, VPP, pad + gray, census) runs on a second stream as soon as the PREVIOUS call's
 * aggregation is done, i.e. next to that call's sum / WTA and post kernels (calls that take the 8-path layout -- fewer than
 * 3 frames -- leave most of the GPU idle during their aggregation: there the next front stage starts as soon as the previous
 * one has delivered its results, next to that aggregation; the gray / census images exist twice for this).  A call of more
 * frames than one batch quantum (vppx_batch_quantum) runs as consecutive parts of one quantum each, pipelined like separate
 * calls; results do not depend on the split.  Safe by construction:
 *   - outputs keep the launch stream's order: the launch stream waits for the front stage, and everything the front stage
 *     produces for the caller (conf_out, l_vpp, r_vpp) is computed into library-owned buffers and copied to the caller's
 *     memory ON THE LAUNCH STREAM -- the front stream never writes caller memory, so buffers allocated per call by a
 *     stream-ordered allocator (torch's) are fine;
 *   - inputs: the front stage waits for an event.  By default the library records it on the launch stream when the call
 *     is made, i.e. the front stage waits for everything queued before the call (correct for any caller; no overlap then).
 *     A caller whose inputs are ready earlier -- resident frames, a loader running on its own stream -- says so with
 *     vppx_inputs_ready_event: the front stage of the NEXT call then waits for that event instead and overlaps the
 *     previous call's tail.
 * Any other entry point called in between uses the same workspace on the launch stream and makes the next front stage
 * wait for the whole launch stream again.  Off by default; ignored during graph capture, stage timing and for sub-stream
 * parts.  (No reference counterpart: test.py handles one pair at a time.) */
int vppx_set_pipeline(vppx_ctx *ctx, int on);
/* One-shot: `hip_event` (a hipEvent_t the caller recorded on the stream that produced them) completes when ALL inputs of
 * the next vppx_occ_vpp_rsgm_dev / vppx_vpp_rsgm_dev call (left, right, g, g_occ) are ready.  NULL withdraws it.  Without
 * pipelining the call runs on the launch stream and the event is simply waited for there. */
int vppx_inputs_ready_event(vppx_ctx *ctx, void *hip_event);
/* Wait for everything queued on the context's stream.  Returns VPPX_E_HIP (message in vppx_last_error) when a fused
 * aggregation launch that has finished by then lost its lock step (see vppx_status). */
int vppx_synchronize(vppx_ctx *ctx);
/* Health of the asynchronous hot path, non-blocking.  The fused aggregation kernel (vppx_uses_vert() == 3) hands
 * diagonal path state between neighbouring waves in lock step; every wait is bounded (VPPX_V3_TIMEOUT_MS, default 250 ms
 * of wall clock per wait), and a wave that gives up marks its launch: the disparities of that call are void -- and are
 * overwritten with NaN by a kernel queued behind the call, so that nothing later on the stream can take them for a
 * disparity map.  The mark is reported -- VPPX_E_HIP, once -- by whichever of these looks first: vppx_status,
 * vppx_synchronize, the next vppx_occlusion_heuristic_dev / vppx_vpp_rsgm_dev / vppx_rsgm_dev call, or vppx_rsgm_host, which
 * repeats its OWN aggregation on the line-parallel kernel and returns correct results (a mark left by an earlier
 * asynchronous call is reported, not consumed).  The context then aggregates with the line-parallel kernel for its next 64
 * aggregation launches (doubling with every further loss, at most 4096) and tries the fused layout again.  After the caller
 * has synchronised the stream (by any means) a 0 from vppx_status covers every call made so far.  (No reference
 * counterpart: aggregate_SSE, rsgm.py:61, is synchronous.) */
int vppx_status(vppx_ctx *ctx);
/* Number of fused launches that reported a lost lock step on this context so far. */
long vppx_lockstep_failures(vppx_ctx *ctx);
/* Bytes of device workspace currently held by the context. */
size_t vppx_workspace_bytes(const vppx_ctx *ctx);
/* Name of the HIP device the context runs on (e.g. "AMD Instinct MI355X"). */
const char *vppx_device_name(const vppx_ctx *ctx);

/* ---- libc-style random stream of the single-frame scans ----------------------------- */
/* init_rand(seed) (vpp_core_opt.pyx:33): restart the context's stream; subsequent
 * vppx_virtual_projection_scan_* calls continue it exactly like libc's global state. */
int vppx_srand(vppx_ctx *ctx, uint32_t seed);
/* Position of that stream: the seed of the last vppx_srand and the number of draws consumed since.  A batched
 * caller (vpp_standalone.vpp through vppx_vpp_host) continues the stream by passing these as
 * VppxVppParams.seed / .rand_offset and reporting what it consumed with vppx_rand_advance. */
int vppx_rand_state(vppx_ctx *ctx, uint32_t *seed, uint64_t *consumed);
int vppx_rand_advance(vppx_ctx *ctx, uint64_t draws);
/* Device-generated glibc rand() stream: out[i] = i-th rand() after srand(seed), for
 * i in [offset, offset+n).  Host pointer. */
int vppx_rand_stream(vppx_ctx *ctx, uint32_t seed, uint64_t offset, int64_t n, int32_t *out);

/* ---- VPP: reference native signatures (host pointers, in place, single frame) -------- */
/* vpp_core_opt.pyx:53-54.  Returns the number of hints (>= 0) or a negative error. */
int vppx_virtual_projection_scan_rnd(vppx_ctx *ctx, uint8_t *l, uint8_t *r, const float *g, int width, int height,
                                     int channels, int uniform_color, int wsize, int direction, float c,
                                     float c_occ, const uint8_t *g_occ, int discard_occluded, int interpolate);
/* vpp_core_opt.pyx:133-134. */
int vppx_virtual_projection_scan_max_dist(vppx_ctx *ctx, uint8_t *l, uint8_t *r, const float *g, int width,
                                          int height, int channels, int uniform_color, int wsize, int wsize_agg_x,
                                          int wsize_agg_y, int direction, float c, float c_occ,
                                          const uint8_t *g_occ, int discard_occluded, int interpolate);

/* ---- VPP: batched form ---------------------------------------------------------------- */
/* l, r are updated in place; g_occ / filled_g may be NULL; n_hints (host, [B]) may be NULL
 * for the _dev form (then nothing is copied back). */
int vppx_vpp_host(vppx_ctx *ctx, const VppxVppParams *p, int B, int H, int W, int C, uint8_t *l, uint8_t *r,
                  const float *g, const uint8_t *g_occ, const float *filled_g, int64_t *n_hints);
int vppx_vpp_dev(vppx_ctx *ctx, const VppxVppParams *p, int B, int H, int W, int C, uint8_t *l, uint8_t *r,
                 const float *g, const uint8_t *g_occ, const float *filled_g, int64_t *n_hints_dev);

/* Number of rand() draws each frame of the last vppx_vpp_host / vppx_vpp_dev call consumed (host array [B]);
 * synchronises the stream.  Lets a caller continue the libc-like stream across calls.  Not defined after the fused
 * entry points (vppx_vpp_rsgm_dev / vppx_occ_vpp_rsgm_dev): those may run a batch in parts (vppx_last_call_parts) and keep
 * the counts of the last part only; their frames draw from srand(seed + frame) and never continue a stream. */
int vppx_vpp_last_draws(vppx_ctx *ctx, int B, uint64_t *draws);

/* ---- rSGM: pyrSGM-compatible stage entry points (host pointers, single frame) ---------- */
int vppx_census5x5(vppx_ctx *ctx, const uint8_t *img, uint32_t *out, int w, int h);
int vppx_cost_census5x5_xyd(vppx_ctx *ctx, const uint32_t *cl, const uint32_t *cr, uint16_t *dsi, int w, int h,
                            int dmax, int n_threads_ignored);
int vppx_aggregate(vppx_ctx *ctx, const uint8_t *img, const uint16_t *dsi, uint16_t *dsi_agg, int w, int h,
                   int dmax, int p1, int p2min, float alpha, int gamma);
/* The same with the image as the reference's glue passes it (rsgm.py:258,270: the padded H x W x 3 colour array):
 * channels = 3 converts to gray on the device first, so that this route and vppx_rsgm_* use the same P2 image. */
int vppx_aggregate_img(vppx_ctx *ctx, const uint8_t *img, int channels, const uint16_t *dsi, uint16_t *dsi_agg, int w,
                       int h, int dmax, int p1, int p2min, float alpha, int gamma);
int vppx_match_wta(vppx_ctx *ctx, const uint16_t *dsi, float *disp, int w, int h, int dmax, float uniqueness);
int vppx_match_wta_right(vppx_ctx *ctx, const uint16_t *dsi, float *disp, int w, int h, int dmax,
                         float uniqueness);
int vppx_subpixel_refine(vppx_ctx *ctx, const uint16_t *dsi, float *disp, int w, int h, int dmax, int method);
int vppx_median3x3(vppx_ctx *ctx, const float *src, float *dst, int w, int h);

/* ---- rSGM: whole compute_rsgm (rsgm.py:250-294), batched -------------------------------- */
/* Stage API: the post-processing of compute_rsgm (models/rsgm/rsgm.py:275-292) on device buffers: crop of the padded
 * left / right disparity maps [B][Hp][Wp] (Hp, Wp = H, W rounded up to multiples of 16, the frame centred as rsgm.py:254-260
 * pads it), left/right check (:230-248, threshold 1), astype(uint8), cv2.filterSpeckles(0, 200, 10), astype(float32),
 * sub-pixel values restored where the integer map kept a pixel (subpixel != 0), _interpolate_background (:185-227).
 * disp_out: [B][H][W] float32. */
int vppx_rsgm_post_dev(vppx_ctx *ctx, int B, int H, int W, const float *disp_l_pad, const float *disp_r_pad, int subpixel,
                       float *disp_out);

/* left / left_vpp / right_vpp: uint8 [B,H,W,C] (C = 1 or 3); disp_out: float32 [B,H,W].
 * hints / validhints: float32 [B,H,W] or both NULL; non-NULL = --guided (_guided_dsi, rsgm.py:116-127). */
int vppx_rsgm_host(vppx_ctx *ctx, const VppxRsgmParams *p, int B, int H, int W, int C, const uint8_t *left,
                   const uint8_t *left_vpp, const uint8_t *right_vpp, const float *hints, const float *validhints,
                   float *disp_out);
int vppx_rsgm_dev(vppx_ctx *ctx, const VppxRsgmParams *p, int B, int H, int W, int C, const uint8_t *left,
                  const uint8_t *left_vpp, const uint8_t *right_vpp, const float *hints, const float *validhints,
                  float *disp_out);

/* ---- fused hot path: VPP (in place on copies) + rSGM, batched, device pointers ----------- */
/* left/right: original pair uint8 [B,H,W,C] (not modified); l_vpp/r_vpp: outputs, patterned
 * pair (may be NULL -> internal scratch); disp_out float32 [B,H,W]. */
int vppx_vpp_rsgm_dev(vppx_ctx *ctx, const VppxVppParams *vp, const VppxRsgmParams *rp, int B, int H, int W, int C,
                      const uint8_t *left, const uint8_t *right, const float *g, const uint8_t *g_occ,
                      uint8_t *l_vpp, uint8_t *r_vpp, float *disp_out);

/* The same with the occlusion mask computed on the way (test.py:154 --maskocc, then :158-225): g_occ =
 * occlusion_heuristic(g, op)[1] never leaves the library between the two stages.  conf_out (uint8 [B,H,W], may be NULL)
 * receives the mask.  This is the entry point bench.py times. */
int vppx_occ_vpp_rsgm_dev(vppx_ctx *ctx, const VppxOccParams *op, const VppxVppParams *vp, const VppxRsgmParams *rp, int B,
                          int H, int W, int C, const uint8_t *left, const uint8_t *right, const float *g, uint8_t *conf_out,
                          uint8_t *l_vpp, uint8_t *r_vpp, float *disp_out);
/* The same for HOST arrays, one frame or a batch: test.py:154-225 in one call.  The pair, the hints and a caller's mask go up
 * once, the disparities -- and, where the pointers are not NULL, the mask
 * (op != NULL: computed on the way, test.py:154), and the patterned pair -- come down once; synchronous.  op == NULL: g_occ
 * (may be NULL) is the caller's mask.  draws_out (may be NULL, [B]): rand() draws each frame consumed (0 for maxDistance or
 * when the call ran in parts), for callers that continue a libc-like stream (vppx_rand_advance). */
int vppx_occ_vpp_rsgm_host(vppx_ctx *ctx, const VppxOccParams *op, const VppxVppParams *vp, const VppxRsgmParams *rp, int B, int H,
                           int W, int C, const uint8_t *left, const uint8_t *right, const float *g, const uint8_t *g_occ,
                           uint8_t *conf_out, uint8_t *l_vpp_out, uint8_t *r_vpp_out, float *disp_out, uint64_t *draws_out);

/* ---- frame stream: the hot path for callers that hold one host frame at a time ---------------------------------------
 * test.py:291-311 iterates a DataLoader with batch size 1 and runs :154-225 per frame on numpy arrays; the kernels want
 * batches of one lock-step round, resident.  A frame stream takes (left, right, hints[, g_occ]) host arrays one frame at a
 * time and returns the disparities (and, by flags, mask and patterned pair) one frame at a time IN INPUT ORDER.  Inside,
 * every frame is copied once, by `copy_threads` threads, into a page-locked ring of `depth` batches; a full batch (`batch`
 * frames, vppx_batch_quantum is a good value) is uploaded on a copy stream under the previous batch's kernels, runs through
 * vppx_occ_vpp_rsgm_dev (op != NULL: the mask of test.py:154 on the way) or vppx_vpp_rsgm_dev with cross-call pipelining,
 * and comes down on a third stream: a copy-out kernel of a few workgroups that stores into the ring (no runtime copy engine:
 * which one a device -> host hipMemcpyAsync takes differs between runtime versions, and the blit-kernel one disturbs the hot
 * path).  With depth >= 3 (recommended: 3) a batch's copy-out waits for the NEXT batch's aggregation to finish, so that it
 * runs beside the sum / WTA kernel and never beside the lock-step launch; the pop, flush or re-run that needs the results
 * first releases it at once.  With depth 2 it runs right behind the batch's own kernels.  Frame f (counted from the stream's creation) draws from srand(vp->seed + f): results
 * equal one-frame calls with that seed whatever the batch size and wherever a flush falls.  The context must launch on
 * its own stream (the default of vppx_create) and must not be used for other calls while the stream exists.
 * A lost lock step (vppx_status) is handled inside: pop verifies each batch after its download and re-runs what was in
 * flight from the device inputs it still holds (vppx_fstream_counts reports how often).  Not thread-safe: one producer /
 * consumer thread (the calls release nothing to other threads but do not hold Python's GIL).
 * flags: VPPX_FS_PATTERNS pop can return the patterned pair; VPPX_FS_MASK (with op) the mask; VPPX_FS_GOCC every push brings
 * a caller's mask (op must be NULL).  copy_threads < 0: chosen from the host's core count.
 * use_distance_patch: dmin / dmax are each frame's own, computed on the device (per_frame_range is set for the stream). */
#define VPPX_FS_PATTERNS 1
#define VPPX_FS_MASK 2
#define VPPX_FS_GOCC 4
typedef struct vppx_fstream vppx_fstream;
int vppx_fstream_create(vppx_ctx *ctx, const VppxOccParams *op, const VppxVppParams *vp, const VppxRsgmParams *rp, int batch,
                        int depth, int H, int W, int C, int flags, int copy_threads, vppx_fstream **out);
void vppx_fstream_destroy(vppx_fstream *fs);
/* One frame in (host arrays of the stream's H x W x C).  Returns when the frame has been copied: the arrays may be
 * reused at once.  The frame that completes a batch also queues the batch's upload, kernels and download (no waiting).
 * Fails with VPPX_E_INVALID_ARG when `depth` whole batches are waiting to be popped. */
int vppx_fstream_push(vppx_fstream *fs, const uint8_t *left, const uint8_t *right, const float *hints, const uint8_t *g_occ);
/* Submit the frames pushed so far as a (smaller) batch: end of the sequence, or a latency bound. */
int vppx_fstream_flush(vppx_fstream *fs);
/* The next frame's results, in input order: *got = 1 and the arrays filled (NULL = not wanted; draws_out = rand() draws the
 * frame consumed, 0 for maxDistance), waiting for its batch when that is still running; *got = 0 when nothing submitted is
 * outstanding (frames of a batch still being filled need vppx_fstream_flush). */
int vppx_fstream_pop(vppx_fstream *fs, float *disp_out, uint8_t *l_vpp_out, uint8_t *r_vpp_out, uint8_t *conf_out,
                     uint64_t *draws_out, int *got);
/* Frames pushed since creation, frames of the batch being filled, frames submitted and not yet popped, batches re-run
 * after a lost lock step (any pointer may be NULL). */
int vppx_fstream_counts(vppx_fstream *fs, int64_t *pushed, int64_t *filling, int64_t *unpopped, int64_t *reruns);

/* ---- hand-off to the deep front-ends (test.py:179-200) -------------------------------------- */
/* uint8 [B,H,W,C] (device) -> [B,C,Hq,Wq] float32 (dst_is_bf16 = 0) or bfloat16 (1) in [0,1]
 * (value/255. as the reference computes it), replicate-padded up to multiples of
 * pad_multiple (32 for PSMNet / RAFT-Stereo, test.py:189-198) with lo = pad//2. */
int vppx_u8_to_nchw_dev(vppx_ctx *ctx, int B, int H, int W, int C, int pad_multiple, const uint8_t *src, void *dst,
                        int dst_is_bf16);

/* PSMNet matching volume (models/psmnet/psmnet.py:157-197): fea_l/fea_r float32 [B,C,H4,W4] ->
 * cost float32 [B,2C,maxdisp/4,H4,W4] (caller-allocated, every element written):
 *   cost[b,c,i,y,x] = x>=i ? fea_l[b,c,y,x] : 0,  cost[b,C+c,i,y,x] = x>=i ? fea_r[b,c,y,x-i] : 0,
 * multiplied, when hints != NULL, by (1-v) + v*10*exp(-(i-h)^2 / (2*(1/4)^2)) with h, v the hints
 * (float32 [B,1,H,W], H/4 == H4, W/4 == W4) and validhints sub-sampled 'nearest' like F.upsample and
 * h = hints*valid/4 (psmnet.py:172-197).  float32 throughout; exp() is the only inexact step. */
int vppx_psmnet_cost_volume_dev(vppx_ctx *ctx, const float *fea_l, const float *fea_r, const float *hints,
                                const float *validhints, int B, int C, int H4, int W4, int H, int W, int maxdisp,
                                float *cost);
/* RAFT-Stereo hint modulation of the all-pairs correlation, in place (models/raft_stereo/corr.py:160-178):
 * corr float32 [B,H4,W2,1,W3] (= einsum(fmap2,fmap3)/sqrt(D), a library GEMM left to the caller);
 * corr[b,y,x,0,k] *= (1-v) + v*10*exp(-(k-(x*v-h))^2/2).  Rows without a hint are untouched. */
int vppx_raft_corr_modulate_dev(vppx_ctx *ctx, float *corr, const float *hints, const float *validhints, int B, int H4,
                                int W2, int W3, int H, int W);
/* Disparity payload decoders (dataloaders/frame_utils.py): KITTI uint16 PNG samples -> disp = v/256
 * float32 + valid = disp>0 uint8 (readDispKITTI :66-69, valid may be NULL); PFM payload (after the
 * text header) -> float32 [H,W(,3)] flipped upside-down, byte-swapped if big-endian (readPFM :34-64).
 * Inflating the PNG stays with the host's image library. */
int vppx_kitti_disp_decode_dev(vppx_ctx *ctx, const uint16_t *png_u16, int64_t n, float *disp, uint8_t *valid);
/* Whole PNG FILES decoded on the device (chunk walk, zlib/DEFLATE inflate, scanline unfiltering), one wave per
 * file of a batch: `blob` (device) holds n_files files back to back -- no alignment or padding is required of it --,
 * `offsets` (HOST, n_files + 1 entries) their byte ranges.  Non-interlaced gray 8/16 bit (C = 1) or RGB 8 bit (C = 3), all files H x W.  disp/valid (C = 1):
 * disp = sample * scale as float32 (scale 1/256 = readDispKITTI frame_utils.py:66-69, 1 = readDispMidd :71-74),
 * valid = disp > 0; out_u8 [n,H,W,C]: the 8-bit samples.  status (device int32 [n_files], may be NULL): 0 = ok,
 * 1 signature, 2 IHDR, 3 unsupported format, 4 size mismatch, 5 bad stream, 6 bad Huffman code, 7 bad filter,
 * 8 truncated, 9 Adler-32 of the inflated stream does not match the one in the file (chunk CRCs are not verified); a file that
 * fails before its first scanline leaves its outputs untouched, one that fails later has written the scanlines decoded so far.
 * Scanlines up to 16 KB (1 + W * bytes per pixel). */
int vppx_png_decode_dev(vppx_ctx *ctx, int n_files, const uint8_t *blob, const int64_t *offsets, int H, int W, int C,
                        float scale, float *disp, uint8_t *valid, uint8_t *out_u8, int32_t *status);
int vppx_pfm_decode_dev(vppx_ctx *ctx, const uint8_t *raw, int H, int W, int channels, int little_endian, float *out);

/* ---- occlusion heuristic (filter.py:246-292): hints -> g_occ mask ------------------------- */
int vppx_occlusion_heuristic_host(vppx_ctx *ctx, int B, int H, int W, const float *hints, int rx, int ry, double l,
                                  double g, double th_conf, double th_filter, uint8_t *conf_out);
int vppx_occlusion_heuristic_dev(vppx_ctx *ctx, int B, int H, int W, const float *hints, int rx, int ry, double l,
                                 double g, double th_conf, double th_filter, uint8_t *conf_out);
/* Both elements of the reference's return value (filter.py:283-292): dmap_out (float32 [B,H,W], may be NULL) = the
 * filtered hints un-warped and passed through interpolate_disparity(dmap, 3), conf_out = the mask.  The reference's
 * interpolate_disparity indexes dmap[y, x +- 1] without a bounds test (filter.py:223,229): column -1 wraps to W-1, column W
 * is the first pixel of the next row; the read past the end of the last row is undefined there and taken as 0 here. */
int vppx_occlusion_heuristic_full_host(vppx_ctx *ctx, int B, int H, int W, const float *hints, int rx, int ry, double l,
                                       double g, double th_conf, double th_filter, float *dmap_out, uint8_t *conf_out);
int vppx_occlusion_heuristic_full_dev(vppx_ctx *ctx, int B, int H, int W, const float *hints, int rx, int ry, double l,
                                      double g, double th_conf, double th_filter, float *dmap_out, uint8_t *conf_out);

/* ---- measurement helpers (bench.py) ------------------------------------------------------- */
/* Time `iters` back-to-back launches of the dominant kernel (8-path aggregation) on the
 * context's stream with hipEvents around the kernel only; returns average ms per launch
 * in *ms_out.  Operates on the workspace state left by the last vppx_rsgm_dev call. */
/* hipGraph replay of vppx_vpp_rsgm_dev: with the mode on, the second call with identical
 * shapes, parameters, pointers and stream is stream-captured and later identical calls launch the instantiated
 * graph (one launch instead of ~40: matters for small batches).  Needs a non-default stream (vppx_set_stream);
 * where capture is not possible the call silently stays on the eager path.  vppx_graph_replays counts launches. */
int vppx_set_graph_mode(vppx_ctx *ctx, int enable);
long vppx_graph_replays(vppx_ctx *ctx);
int vppx_time_aggregate(vppx_ctx *ctx, int iters, float *ms_out);
/* Average duration (ms) of the aggregation kernel over the last `last_n` launches the pipeline made
 * on this context, from hipEvent pairs recorded around every launch on its launch stream (ring of
 * 64).  Call after synchronising.  last_n <= 0 resets the launch counter.  n_used may be NULL. */
int vppx_agg_kernel_ms(vppx_ctx *ctx, int last_n, float *avg_ms, int *n_used);
/* The same for the W/E launch of the fused layout (line-parallel kernel; the other launch of its aggregation stage). */
int vppx_we_kernel_ms(vppx_ctx *ctx, int last_n, float *avg_ms, int *n_used);
/* Frames per launch that vppx_time_aggregate re-runs: vppx_vpp_rsgm_dev splits a batch over
 * internal sub-streams, and the helper times the launches of one part. */
int vppx_time_aggregate_frames(vppx_ctx *ctx);
/* Same for one part of the stage: 1 = horizontal-path line kernel, 2 = the band launches of the
 * vertical/diagonal paths (fast path only). */
int vppx_time_aggregate_part(vppx_ctx *ctx, int iters, int part, float *ms_out);
/* Aggregation layout of the last fused / rsgm call: 0 = all 8 paths in the line-parallel kernel, 3 = N/NW/NE and
 * S/SW/SE fused three at a time in the lock-step kernel + W and E line-parallel (register-window kernel for D = 128 / 192 /
 * 256) (default for D = 128 / 192 from 3 frames per call on; environment VPPX_VERT = 0 / 3 forces a layout), 1 = the round-1
 * band-marching experiment
 * (VPPX_VERT = 1).  All layouts give identical results. */
int vppx_uses_vert(vppx_ctx *ctx);
/* Which fused kernel the last call used: 16 = sgm_vert4_kernel (16 pixels per wave, 4 lanes per pixel: the default once a
 * batch fills the chip with its groups), 8 = sgm_vert3_kernel (8 lanes per pixel; VPPX_V3_PPW = 8 / 16 forces one),
 * 0 = the last call did not use the fused layout. */
int vppx_fused_pixels_per_wave(vppx_ctx *ctx);
/* Batch quantum of the fused layout for H x W frames and this disparity range: the number of frames that fills the GPU
 * with one whole round of (frame, pass) groups of the lock-step kernel -- batches that are a multiple of it leave no
 * part-filled last round (540 x 960 x 192 on an MI355X: 16; 375 x 1242 x 192: 12).  0 when the fused layout does not apply
 * to this device or shape.  A scheduling hint only: every batch size gives the same results.  (Runs a one-time device
 * probe on first use; not inside a graph capture.) */
int vppx_batch_quantum(vppx_ctx *ctx, int H, int W, int dmax);
/* Number of consecutive parts the last fused call (vppx_vpp_rsgm_dev / vppx_occ_vpp_rsgm_dev) ran as: a batch larger than
 * one round of the lock-step kernel runs as parts of one round each (VPPX_CHUNK); 1 = the whole batch at once. */
int vppx_last_call_parts(vppx_ctx *ctx);
/* Per-stage hipEvent timing of the last vppx_vpp_rsgm_dev/vppx_rsgm_dev call when stage
 * timing is enabled: fills ms[0..n) and returns the number of stages; names via
 * vppx_stage_name(i). */
int vppx_enable_stage_timing(vppx_ctx *ctx, int enable);
int vppx_get_stage_ms(vppx_ctx *ctx, float *ms, int max_n);
const char *vppx_stage_name(int i);

/* ---- environment ------------------------------------------------------------------------------
 * Every variable the shipped library reads, all of them ONCE, in vppx_create (nothing else calls getenv; `tools/build_exp.sh`
 * measurement builds add VPPX_V3_IGNORE_LOST, VPPX_EXP_WE_TRACE and the other VPPX_EXP_*, tools/ only):
 *   VPPX_VERT           aggregation layout: -1 / unset = by shape, 0 = eight line-parallel paths, 3 = fused layout whenever the
 *                       shape allows it, 1 = the round-1 band-marching kernel (vppx_uses_vert)
 *   VPPX_CHUNK          parts of a fused call: unset = one round of the lock-step kernel each, 0 = whole batches, n = n frames
 *   VPPX_SUBSTREAMS     1..4 child contexts that share a batch (default 1)
 *   VPPX_V3_PPW         8 / 16: pixels per wave of the fused vertical kernel (default: by batch size)
 *   VPPX_V3_TIMEOUT_MS  bound of one neighbour wait of the lock-step kernel (default 250)
 *   VPPX_V3_SPIN_LIMIT  the same bound in polls (tests force the give-up path with 1; default unbounded)
 *   VPPX_VARIANT        comma-separated tokens that select alternative kernels with identical results, each run by
 *                       tests/test_gpu_variants.py: sum_general, sum_gl8, sum_trap0, sum_trap1 (fused sum / WTA kernel), gw4, gw8,
 *                       gw16 (lanes per pixel of the line-parallel kernel), we_line (W / E on the line-parallel kernel), we_after (W / E never next to
 *                       an under-filled lock-step launch), we_lq0, we_lq1 (W / E kernel: left-view operands per step / per quad of steps), we_whole (W / E kernel: no line of the last layer of waves is cut into pieces; tests: we_layer=N takes a layer to be N waves, we_mute makes every later piece give up waiting and compute its line from the start),
 *                       maxdist_lds, maxdist_global (one-wave maxDistance kernels) */

#ifdef __cplusplus
}
#endif
#endif /* VPPX_H */
