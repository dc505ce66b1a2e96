// vppx_internal.h -- shared declarations of libvppx.so (gfx950 only; not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/vppx.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
void vppx_set_error(const char *fmt, ...);

#define VPPX_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            vppx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return VPPX_E_HIP;                                                                      \
        }                                                                                           \
    } while (0)

#define VPPX_CHECK_LAUNCH() VPPX_HIP(hipGetLastError())

// Every entry point runs with the context's device current and puts the caller's device back on return
// (a torch process may hold contexts on several GPUs; the library must not change its current device).
struct DevGuard {
    int prev = -1;
    bool changed = false;
    explicit DevGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
    }
    ~DevGuard()
    {
        if (changed && prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------------------
// workspace arena: named, grow-only device buffers owned by the context.  Everything the
// pipeline needs between kernels lives in HBM for the lifetime of the context (288 GB:
// batches are sized so that whole cost-path volumes stay resident).
// ---------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

enum WsSlot {
    WS_HINT_X = 0,   // compacted hint rows
    WS_HINT_REC,
    WS_HINT_DENSE,
    WS_HINT_BITS,    // per row: which scan positions hold a hint (u64 words)
    WS_RCNT,
    WS_FILLED_G,
    WS_GRAY_CTX,
    WS_RLIST,
    WS_LWORK,        // L pixels whose window holds an occluded hint (second pass of the sparse L mapping)
    WS_LWORK_CNT,
    WS_ROW_COUNT,
    WS_ROW_DRAWS,
    WS_ROW_BASE,
    WS_FRAME_TOT,
    WS_HINT_RANGE,   // [B] {dmin, dmax} of every frame's own hints (use_distance_patch with per_frame_range)
    WS_RAND,
    WS_SEEDS,
    WS_VPP_L,        // scratch copies of the pair (fused path)
    WS_VPP_R,
    WS_VPP_LSRC,
    WS_OCC_ZERO,
    WS_GRAY_L,       // padded gray images
    WS_GRAY_LV,
    WS_GRAY_RV,
    WS_CENSUS_L,
    WS_CENSUS_R,
    WS_GRAY_L2,      // second set of the images the aggregation reads (pipelined calls alternate: front_end)
    WS_CENSUS_L2,
    WS_CENSUS_R2,
    WS_PATHS,        // 8 per-path L volumes
    WS_S,            // aggregated volume (u16)
    WS_SV,           // per-pass sums of the vertical/diagonal paths (band-marching kernel)
    WS_VSTATE,
    WS_VMIN,
    WS_V3ERR,        // device-side copy of the fused kernel's error word (void_if_lost_kernel)
    WS_WE_TRACE,     // experiment builds: per-block wall-clock trace of the W/E kernel
    WS_WE_HAND,      // W/E kernel: path states handed from piece to piece of the split tail lines + a flag per line (We12Args)
    WS_DSI,          // materialised cost volume (stage API only)
    WS_DISP_L0,
    WS_DISP_L1,
    WS_DISP_R0,
    WS_DISP_R1,
    WS_FD8,
    WS_SPK_ROOTS,    // per tile of the speckle filter: its tile-local roots
    WS_SPK_NROOTS,
    WS_ROW_ANY,      // per row of the result: does it hold a valid disparity (column pass of _interpolate_background)
    WS_LABEL,
    WS_LCOUNT,
    WS_P2LUT,
    WS_STAGE_A,      // host-API staging (device copies of host arrays)
    WS_STAGE_B,
    WS_STAGE_C,
    WS_STAGE_D,
    WS_STAGE_E,
    WS_STAGE_F,
    WS_OCC_OMAP,
    WS_OCC_OUT,       // the mask of the fused occlusion + VPP + rSGM call (library-owned: the front stage never writes caller memory)
    WS_OCC_TMP,       // unwarp scratch of the occlusion heuristic (its own slot: with vppx_set_pipeline it runs next to the previous call's speckle filter, which owns WS_LABEL)
    WS_NHINTS,
    WS_HANDOFF_H,    // hints / valid at feature resolution (hand-off kernels)
    WS_HANDOFF_V,
    WS_PNG_RAW,      // concatenated IDAT payloads of a batch of PNG files (the zlib streams)
    WS_PNG_OFFS,
    WS_PNG_STATUS,
    WS_NUM
};

#define VPPX_MAX_STAGES 96
#define VPPX_MAX_DEVICES 64 // power of two: per-device one-time initialisations are indexed by device id

struct vppx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool legacy_stream = false;    // launches go to the legacy default (null) stream: vppx_set_stream_legacy
    hipStream_t stream2 = nullptr; // side stream: horizontal paths overlap the vertical band launches
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // sub-contexts: the fused batched call splits its frames over `nsub` child contexts (own stream
    // and arena) so that latency-bound stages of one part overlap bandwidth-bound stages of another
    int nsub = 1;   // VPPX_SUBSTREAMS=2 overlaps two half batches (+3 %; off by default so that profiles and
                    // the in-process kernel timing see the same, un-overlapped launches)
    vppx_ctx *sub[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t sub_done[4] = {nullptr, nullptr, nullptr, nullptr};
    bool is_child = false;
    // Environment, read ONCE in vppx_create (include/vppx.h lists every variable the shipped library reads): VPPX_CHUNK and
    // the VPPX_VARIANT tokens, which select kernels that give identical results (tests/test_gpu_variants.py runs each)
    struct Knobs {
        int chunk = -1;       // VPPX_CHUNK: -1 parts of one lock-step round (default), 0 whole batches, n parts of n frames
        int sum_general = 0;  // "sum_general": the general WTA decision code of the fused sum / WTA kernel (round 1)
        int gw = 0;           // "gw4" / "gw8" / "gw16": lanes per pixel of the line-parallel aggregation kernel
        int maxdist = 0;      // "maxdist_lds" (1): one wave per chain, rows in LDS; "maxdist_global" (2): in place in global memory
        int we_line = 0;      // "we_line": W / E of the fused layout on the line-parallel kernel instead of sgm_we12_kernel
        int sum_gl8 = 0;      // "sum_gl8": 8 lanes per pixel in the fused sum / WTA kernel (D = 128 / 192)
        int sum_trap = 2;     // D = 256, fused layout: trapezoid ring with spare slots; "sum_trap1": without them; "sum_trap0": uniform ring, 32-pixel rounds
        int we_next = 1;      // W/E next to an under-filled lock-step launch (rsgm_vert3_plan); "we_after": always behind it
        int we_split = 1;     // W/E kernel: the lines of the last, part-filled layer of waves are cut into pieces; "we_whole": every line whole
        int we_layer = 0;     // "we_layer=N" (tests): the W/E launcher takes a layer of waves to be N instead of the device's SIMD count
        int we_mute = 0;      // "we_mute" (tests): pieces do not publish their state: every later piece takes the give-up path
        int we_lq = -1;       // "we_lq0" / "we_lq1": the W/E kernel's left-view operand loads (per step / per quad of steps); default by launch
        int sum_blocks = 0;   // (experiment builds) "sum_blocks=N": forced number of blocks of the sum / WTA kernel
    } knobs;
    int use_vert = -1;             // VPPX_VERT: -1 pick by shape (default), 0 eight line-parallel paths, 1 band marching, 3 fused vertical kernel
    int last_vert = 0;             // what the last aggregation used (vppx_uses_vert)
    int last_parts = 1;            // parts the last fused call ran as (vppx_last_call_parts)
#ifdef VPPX_EXPERIMENT
    std::string exp_we_trace;      // VPPX_EXP_WE_TRACE: file the W/E kernel's per-block trace of the last launch goes to (tools/we_trace.py)
#endif
    unsigned we_serial = 0;        // launch serial of the W/E kernel's hand-off flags
    void *we_hand_seen = nullptr;  // the hand-off buffer as it was last cleared (launch_we12)
    size_t we_hand_cap = 0;
    bool we_beside_vert = false;   // the W/E launch being queued runs next to a lock-step launch (run_aggregation)
    unsigned long long *draws_dst = nullptr; // device [B]: where the fused call leaves every frame's draw count (set per call by a frame stream)
    int last_sum_nvol = 0;         // volumes the last fused sum / WTA launch added (4: fused layout, 8: eight paths) and its
    int last_sum_D = 0, last_sum_B = 0; // shape: what a pipelined front stage has to fit next to (vpp_rsgm_one)
    // cross-call pipelining (vppx_set_pipeline): the front stage of a fused call (occlusion heuristic, VPP, pad + gray,
    // census) runs on stream_front, after the PREVIOUS call's aggregation, i.e. under that call's sum / WTA and post kernels
    bool pipeline = false;
    bool front_active = false;     // ctx->stream currently is stream_front
    bool pipe_call = false;        // the running vpp_rsgm call is pipelined (record ev_agg_done after its aggregation)
    bool have_agg_done = false;
    bool pipe_early = false;       // this pipelined call lets the next front stage start next to its aggregation (few frames per call)
    int pipe_parity = 0;           // which set of gray / census images the next pipelined call writes
    bool pipe_mid = false;         // the next front stage starts behind the fused vertical kernel, next to W/E (VPPX_VARIANT=pipe_mid of an experiment build)
    bool agg_done_recorded = false; // ... and this call's aggregation has recorded ev_agg_done already
    const u8 *last_gl = nullptr;   // the images the last call aggregated from (timing helpers)
    const u32 *last_cl = nullptr, *last_cr = nullptr;
    hipStream_t stream_front = nullptr, main_saved = nullptr;
    hipEvent_t ev_agg_done = nullptr, ev_front_done = nullptr, ev_inputs_auto = nullptr;
    void *inputs_ev = nullptr;     // caller's "inputs of the next call are ready" event (vppx_inputs_ready_event), one-shot
    // front-stage results the caller asked for (mask, patterned pair): computed into library-owned buffers on the front
    // stream, copied to the caller's memory on the LAUNCH stream once it has waited for the front stage
    struct PipeCopy { void *dst; const void *src; size_t bytes; bool early; } pipe_copy[3]; // early: ready at ev_occ_done already
    int n_pipe_copy = 0;
    hipEvent_t ev_occ_done = nullptr;   // front stream: the occlusion mask of the call is complete
    bool occ_done_recorded = false;
    // LDS a front-stage kernel can count on per CU next to the sum / WTA kernel of the previous part or call, which holds
    // one block per CU: 160 KB minus that kernel's tile ring (36 KB at D <= 192, 7 KB at D = 256).  Kernels pick block
    // shapes that fit; set by the pipelined fused entry, "everything" otherwise.
    size_t front_lds_budget = 64 * 1024;
    void *xbuf_last = nullptr;          // exchange records of the last lock-step launch ...
    size_t xbuf_last_bytes = 0;
    bool xbuf_cleared = false;          // ... and whether a clear of them is already queued on the launch stream behind it
    unsigned *vert3_err = nullptr; // pinned host word the fused vertical kernel sets (to its launch serial) when a wave gave up waiting
    bool vert3_broken = false;     // the XCD / residency probe failed on this device: the context never uses the fused layout
    int vert3_rest = 0;            // aggregations left on the line-parallel layout after a lost lock step, before the fused layout is tried again
    int vert3_backoff = 64;        // ... doubling with every loss (64, 128, ... 4096)
    bool vert3_probed = false;     // rsgm_vert3_probe has run on this context's device
    long lockstep_failures = 0;    // fused launches that reported a lost lock step (vppx_lockstep_failures)
    unsigned lockstep_last_serial = 0; // serial of the launch the last report was about
    struct V3Caps {                // what rsgm_vert3_probe found out about the device and this build of the kernels
        bool ok = false;
        int nxcd = 0, cus_per_xcd = 0;
        int blocks_per_cu[4] = {0, 0, 0, 0}; // D = 64, 128, 192, 256 (8 pixels per wave)
        int blocks_per_cu16[4] = {0, 0, 0, 0}; // the same for 16 pixels per wave (sgm_vert4_kernel)
        int ppw = 0;               // VPPX_V3_PPW: pixels per wave of the fused kernel at D = 192 (0 = by batch size, 8, 16)
        int wall_khz = 100000;     // rate of s_memrealtime
        int timeout_ms = 250;      // bound of one wait for a neighbour (VPPX_V3_TIMEOUT_MS)
        unsigned spin_limit = 0;   // polls per wait, 0 = unbounded (VPPX_V3_SPIN_LIMIT: tests force the give-up path with 1)
        unsigned serial = 0;       // serial of the last fused launch
        int last_ppw = 0;          // pixels per wave of the last fused launch (8: sgm_vert3_kernel, 16: sgm_vert4_kernel)
        bool ignore_lost = false;  // VPPX_V3_IGNORE_LOST (experiment builds only): measurement runs with forced give-ups (results void) carry on
    } v3;
    DevBuf ws[WS_NUM];
    std::string devname;
    // libc-like stream state of the single-frame scans
    uint32_t rnd_seed = 1;
    uint64_t rnd_consumed = 0;
    // last rsgm geometry (for vppx_time_aggregate)
    int last_B = 0, last_Hp = 0, last_Wp = 0, last_D = 0;
    VppxRsgmParams last_rp;
    bool have_last = false;
    // hipEvent pairs around the aggregation kernel of the most recent pipeline calls (ring), recorded
    // on the launch stream: vppx_agg_kernel_ms averages them after the caller's synchronize
    static const int AGG_RING = 64;
    hipEvent_t agg_ev[2][AGG_RING];
    bool agg_ev_created = false;
    long agg_calls = 0;
    // the same around the W/E launch of the fused layout (vppx_we_kernel_ms): its in-step duration differs from what a
    // back-to-back re-launch measures, so both are on record
    hipEvent_t we_ev[2][AGG_RING];
    long we_calls = 0;
    // P2 look-up table: host copy kept alive for the asynchronous upload, re-sent only when the penalties change
    u16 lut_host[256];
    bool lut_valid = false;
    int lut_p2min = 0, lut_gamma = 0, lut_maxp2 = 0;
    float lut_alpha = 0.f;
    // hipGraph replay of the fused call (vppx_set_graph_mode): the second identical call
    // (same shapes, parameters and pointers) is stream-captured, later ones launch the instantiated graph
    struct GraphKey {
        int B, H, W, C;
        VppxVppParams vp;
        VppxRsgmParams rp;
        const void *ptr[8];
        VppxOccParams op;
        int has_op;
        void *stream;
        unsigned long ws_gen;
    };
    bool graph_mode = false, capturing = false;
    bool have_gkey = false, have_lastkey = false;
    GraphKey gkey, lastkey;
    hipGraphExec_t gexec = nullptr;
    unsigned long ws_gen = 0;      // bumped whenever a workspace buffer is (re)allocated
    long graph_replays = 0, graph_captures = 0;
    // stage timing
    bool stage_timing = false;
    bool stage_append = false;     // the running call is a later part of a split batch: its marks follow the earlier parts'
    int n_stages = 0;
    hipEvent_t ev[VPPX_MAX_STAGES + 1];
    bool ev_created = false;
    int stage_id[VPPX_MAX_STAGES];
    size_t total_bytes = 0;
    std::vector<long long> png_offs_host; // kept alive for the asynchronous upload
};

int ws_reserve(vppx_ctx *ctx, WsSlot s, size_t bytes, void **out);
template <typename T>
static inline int ws_get(vppx_ctx *ctx, WsSlot s, size_t count, T **out)
{
    void *p = nullptr;
    int rc = ws_reserve(ctx, s, count * sizeof(T), &p);
    *out = (T *)p;
    return rc;
}

// stage ids (vppx_stage_name)
enum {
    ST_VPP_COMPACT = 0,
    ST_VPP_RAND,
    ST_VPP_APPLY,
    ST_PAD_GRAY,
    ST_CENSUS,
    ST_AGGREGATE,
    ST_SUM_WTA,
    ST_WTA_RIGHT,
    ST_MEDIAN_INTERP,
    ST_POST,
    ST_OCC,
    ST_COUNT
};
void stage_begin(vppx_ctx *ctx);
void stage_mark(vppx_ctx *ctx, int stage);

// ---------------------------------------------------------------------------------------
// kernel launchers (defined in the .hip files)
// ---------------------------------------------------------------------------------------
struct RsgmGeom {
    int B, H, W, C;      // input frames
    int Hp, Wp;          // padded to multiples of 16 (rsgm.py:254-256)
    int pad_l, pad_r, pad_t, pad_b;
    int D;
};

// rsgm_kernels.hip
int rsgm_launch_pad_gray(vppx_ctx *ctx, const RsgmGeom &g, const u8 *img, u8 *gray);
int rsgm_launch_pad_gray_n(vppx_ctx *ctx, const RsgmGeom &g, int n, const u8 *const *img, u8 *const *gray); // n <= 3, one launch
int rsgm_launch_to_nchw(vppx_ctx *ctx, int B, int H, int W, int C, int mult, const u8 *src, void *dst, int bf16);
int rsgm_launch_census(vppx_ctx *ctx, int B, int Hp, int Wp, const u8 *gray, u32 *census);
int rsgm_launch_pad_gray_census(vppx_ctx *ctx, const RsgmGeom &g, const u8 *left, const u8 *left_vpp, const u8 *right_vpp, u8 *gray_left,
                                u32 *census_l, u32 *census_r); // the three of the fused pipeline in one launch
int rsgm_launch_census_n(vppx_ctx *ctx, int B, int Hp, int Wp, int n, const u8 *const *gray, u32 *const *census); // n <= 2
int rsgm_launch_cost(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u32 *cl, const u32 *cr, u16 *dsi);
int rsgm_launch_guided_dsi(vppx_ctx *ctx, const RsgmGeom &g, u16 *dsi, const float *hints, const float *valid);
// 8-path aggregation.  Cost source: census pair (dsi == nullptr) or a materialised u16 DSI.
// Writes the 8 per-path volumes into `paths` (element size elem_bytes = 1 or 2).
int rsgm_launch_paths(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u8 *gray, const u32 *cl, const u32 *cr,
                      const u16 *dsi, const u16 *p2lut, int p1, void *paths, int elem_bytes, int dir_mask);
// band-marching vertical/diagonal paths (3 summed paths per pass, one byte per cell)
int rsgm_vert3_probe(vppx_ctx *ctx, u32 *scratch_dev, bool *ok);
bool rsgm_vert3_supported(int B, int Hp, int Wp, int D, int maxp2);
bool rsgm_vert3_fits(const vppx_ctx *ctx, int B, int Wp, int D);
int rsgm_vert3_frames_per_round(const vppx_ctx *ctx, int Wp, int D);
bool rsgm_vert3_wide(const vppx_ctx *ctx, int B, int Wp, int D);
size_t rsgm_vert3_xbuf_bytes(int B, int Wp, int D);
void rsgm_vert3_plan(const vppx_ctx *ctx, int B, int Wp, int D, int *whole_frames, bool *rest_underfilled);
int rsgm_launch_vert3_range(vppx_ctx *ctx, hipStream_t stream, int B_total, int f0, int nB, int Hp, int Wp, int D, const u8 *gray,
                            const u32 *cl, const u32 *cr, const u16 *p2lut, int p1, u8 *sv, u32 *xbuf, unsigned *err, unsigned *err_dev,
                            bool next_to_we);
int rsgm_launch_vert3(vppx_ctx *ctx, hipStream_t stream, int B, int Hp, int Wp, int D, const u8 *gray, const u32 *cl,
                      const u32 *cr, const u16 *p2lut, int p1, u8 *sv, u32 *xbuf, unsigned *err, unsigned *err_dev);
int rsgm_launch_void_if_lost(vppx_ctx *ctx, float *out, size_t n, const unsigned *err_dev, unsigned serial);
bool rsgm_vert_supported(int D, int maxp2);
size_t rsgm_vert_state_bytes(int B, int Wp, int D);
size_t rsgm_vert_min_elems(int B, int Wp);
int rsgm_launch_vert(vppx_ctx *ctx, hipStream_t stream, int B, int Hp, int Wp, int D, const u8 *gray, const u32 *cl,
                     const u32 *cr, const u16 *p2lut, int p1, u8 *sv, u8 *gst, u16 *gmin);
// S = sum of the 8 path volumes (+ left WTA, sub-pixel) ; disp_l may be null (stage API: S only)
int rsgm_launch_sum_wta(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const void *paths, int elem_bytes, u16 *S,
                        u16 *ST, float *disp_l, u32 factor_uniq, int do_subpixel);
int rsgm_launch_wta_right_t(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *ST, float *disp, u32 factor_uniq);
int rsgm_paths_elem_bytes(int D, int maxp2);
// fused sum + left/right WTA; returns 1 (not an error) when the shape is not covered
// max_path_value: upper bound of one path value (Cmax + P2max); 0 = unknown
size_t rsgm_sum_lds_bytes(const vppx_ctx *ctx, int D, int nvol);
int rsgm_launch_sum_wta_lr(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const void *const *vols, int nvol, int elem_bytes,
                           float *disp_l, float *disp_r, u32 factor_uniq, int do_subpixel, int max_path_value);
int rsgm_launch_wta_left(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *S, float *disp, u32 factor_uniq);
int rsgm_launch_subpixel(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *S, float *disp);
int rsgm_launch_wta_right(vppx_ctx *ctx, int B, int Hp, int Wp, int D, const u16 *S, float *disp, u32 factor_uniq);
int rsgm_launch_median(vppx_ctx *ctx, int B, int Hp, int Wp, const float *src, float *dst);
int rsgm_launch_linear_interp_clip(vppx_ctx *ctx, int B, int Hp, int Wp, const float *src, float *dst);
// median3x3 + _linear_interpolate + clip of both views in one launch (fused pipeline)
int rsgm_launch_median_interp_clip(vppx_ctx *ctx, int B, int Hp, int Wp, const float *src_l, float *dst_l, const float *src_r,
                                   float *dst_r);
int rsgm_launch_post(vppx_ctx *ctx, const RsgmGeom &g, const float *dl_pad, const float *dr_pad, int subpixel, float *out);

// vpp_kernels.hip
struct VppGeom {
    int B, H, W, C;
};
int vpp_launch(vppx_ctx *ctx, const VppxVppParams &p, const VppGeom &g, u8 *l, u8 *r, const float *gmap,
               const u8 *occ, const float *filled_g, int64_t *n_hints_dev, const u32 *seeds_dev, const u8 *r_orig = nullptr);
// r_orig: the un-patterned right image when r is a copy of it (lets the L and R sides run side by side with a mask)
// _bilateral_filling (vpp_standalone.py:372-394) of the hints g guided by BGR2GRAY(left)
int vpp_launch_bilateral_fill(vppx_ctx *ctx, const VppxVppParams &p, const VppGeom &g, const u8 *left, const float *gmap,
                              float *filled_out);
int vpp_launch_rand_stream(vppx_ctx *ctx, u32 seed, u64 offset, int64_t n, int32_t *out_dev);
u64 vpp_draws_upper_bound(const VppxVppParams &p, const VppGeom &g);

// handoff_kernels.hip
int handoff_psmnet_cost_volume(vppx_ctx *ctx, const float *fl, const float *fr, const float *hints, const float *valid, int B,
                               int C, int H4, int W4, int H, int W, int maxdisp, float *cost);
int handoff_raft_corr_modulate(vppx_ctx *ctx, float *corr, const float *hints, const float *valid, int B, int H4, int W2, int W3,
                               int H, int W);
int handoff_kitti_decode(vppx_ctx *ctx, const uint16_t *png, size_t n, float *disp, u8 *valid);
int handoff_pfm_decode(vppx_ctx *ctx, const u8 *raw, int H, int W, int channels, int little, float *out);
// png_kernels.hip
int handoff_png_decode(vppx_ctx *ctx, int n_files, const u8 *blob_dev, const long long *offs_dev, int H, int W, int C, u8 *zcat,
                       const long long *zoffs_dev, float scale, float *disp, u8 *valid, u8 *out_u8, int *status_dev);

// occ_kernels (in vpp_kernels.hip)
int occ_launch(vppx_ctx *ctx, int B, int H, int W, const float *hints, int rx, int ry, double l, double g,
               double th_conf, double th_filter, float *omap, u8 *conf_out, float *dmap_out = nullptr);
