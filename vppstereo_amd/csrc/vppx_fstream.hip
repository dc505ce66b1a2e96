// vppx_fstream.hip -- host-side streaming adapter of the hot path (include/vppx.h, "frame stream").
//
// The reference's harness holds numpy frames one at a time (test.py:291-311: batch size 1, a DataLoader; :154-225 per frame);
// the kernels want batches of one lock-step round (16 frames at 540x960x192) that are resident in HBM.  A frame stream
// sits between the two: frames are pushed one at a time from pageable host arrays, results are popped one at a time in
// input order, and in between
//   * a frame is copied ONCE on the host, by a small pool of threads, into a page-locked ring the stream owns (the slot of
//     the batch being filled) -- the copy a pageable hipMemcpy would make into the runtime's own staging buffer anyway;
//   * a full batch goes up on a copy stream (asynchronous DMA from the ring) under the kernels of the batch before it: the
//     hot path is called with the upload's event as its inputs-ready event and with cross-call pipelining on, so the front
//     stage of batch k+1 starts next to the sum / WTA kernel of batch k as soon as its inputs are there;
//   * the disparities (and, on request, mask and patterned pair, and the draw counts) come down on a third stream into the
//     ring, under the kernels of the next batch: by a copy-out kernel of a few workgroups, held back (ring of >= 3 batches)
//     until the next batch's lock-step aggregation is over (fs_copy_out_kernel, fs_submit).
// Frame f of the stream (counted from its creation) draws from srand(seed + f), as frame f of any batched call does: results
// do not depend on the batch size, on flushes or on where a batch boundary falls, and equal one-frame calls with that seed.
// A lost lock step (vppx_status) never surfaces as wrong disparities: pop verifies every batch after its download and
// re-runs what is in flight from the still-intact device inputs (the context rests on the line-parallel layout meanwhile).
#include "vppx_internal.h"
#include "copy_pool.h"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string.h>
#include <thread>

int vppx_lockstep_check_internal(vppx_ctx *ctx); // vppx_api.hip
int vppx_check_hot_path_args_internal(const VppxVppParams *vp, const VppxRsgmParams *rp, int B, int H, int W, int C);

namespace {

using vppx_host::CopyJob;
using vppx_host::CopyPool;

struct Slot {
    // page-locked host ring: inputs of the batch being filled / uploaded, outputs of the batch downloaded
    u8 *h_in = nullptr, *h_out = nullptr;
    // device buffers of the batch
    u8 *d_in = nullptr, *d_out = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
    int nb = 0;            // frames of the batch the slot holds
    uint32_t seed = 0;     // seed of its first frame
    bool submitted = false, verified = false;
    bool out_pending = false; // the batch's kernels are enqueued, its copy-out is not yet (see fs_submit)
    bool out_method_rnd = false;
};

} // namespace

struct vppx_fstream {
    vppx_ctx *ctx = nullptr;
    bool prev_pipeline = false;
    VppxOccParams op;
    bool have_op = false;
    VppxVppParams vp;
    VppxRsgmParams rp;
    int batch = 0, depth = 0, H = 0, W = 0, C = 0, flags = 0;
    size_t img = 0, px = 0;                 // bytes of one image, pixels of one frame
    size_t in_frame = 0, out_frame = 0;     // ring bytes per frame
    size_t o_left = 0, o_right = 0, o_hint = 0, o_gocc = 0;          // offsets inside a slot's input area (planes of `batch` frames)
    size_t o_disp = 0, o_lv = 0, o_rv = 0, o_conf = 0, o_draws = 0;  // offsets inside a slot's output area
    size_t in_bytes = 0, out_bytes = 0;
    std::vector<Slot> slots;
    hipStream_t s_in = nullptr, s_out = nullptr;
    CopyPool *pool = nullptr;
    // positions, in frames / batches since creation
    uint64_t pushed = 0;       // frames pushed
    uint64_t fill_batch = 0;   // index of the batch being filled
    int fill_count = 0;        // frames in it
    uint64_t sub_batches = 0;  // batches submitted
    uint64_t pop_batch = 0;    // batch the next pop reads
    int pop_idx = 0;           // next frame inside it
    uint64_t reruns = 0;       // batches re-run after a lost lock step
    bool lost_pending = false; // a lost lock step was noticed while submitting: what is in flight is re-run at the next pop
};

static int fs_free(vppx_fstream *fs)
{
    if (!fs) return 0;
    DevGuard g(fs->ctx->device);
    (void)hipStreamSynchronize(fs->ctx->stream);
    if (fs->s_in) (void)hipStreamSynchronize(fs->s_in);
    if (fs->s_out) (void)hipStreamSynchronize(fs->s_out);
    for (auto &s : fs->slots) {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    }
    if (fs->s_in) (void)hipStreamDestroy(fs->s_in);
    if (fs->s_out) (void)hipStreamDestroy(fs->s_out);
    delete fs->pool;
    fs->ctx->pipeline = fs->prev_pipeline;
    fs->ctx->have_agg_done = false;
    fs->ctx->draws_dst = nullptr;
    delete fs;
    return 0;
}

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int vppx_fstream_create(vppx_ctx *ctx, const VppxOccParams *op, const VppxVppParams *vp, const VppxRsgmParams *rp,
                                   int batch, int depth, int H, int W, int C, int flags, int copy_threads, vppx_fstream **out)
{
    if (!ctx) { vppx_set_error("context is NULL (vppx_create failed? there is no CPU fallback)"); return VPPX_E_NO_DEVICE; }
    if (!out || !vp || !rp) { vppx_set_error("vppx_fstream_create: NULL argument"); return VPPX_E_INVALID_ARG; }
    *out = nullptr;
    if (batch < 1 || batch > 4096 || depth < 2 || depth > 8 || H < 1 || W < 1 || (C != 1 && C != 3)) {
        vppx_set_error("vppx_fstream_create: batch in 1..4096, depth in 2..8, C in {1, 3} (got batch %d, depth %d, %d x %d x %d)", batch, depth, H, W, C);
        return VPPX_E_INVALID_ARG;
    }
    if (op && (flags & VPPX_FS_GOCC)) { vppx_set_error("vppx_fstream_create: either the occlusion heuristic (op) or pushed masks (VPPX_FS_GOCC)"); return VPPX_E_INVALID_ARG; }
    if ((flags & VPPX_FS_MASK) && !op) { vppx_set_error("vppx_fstream_create: VPPX_FS_MASK returns the mask of the occlusion heuristic (op is NULL)"); return VPPX_E_INVALID_ARG; }
    if (ctx->is_child || ctx->nsub > 1 || ctx->graph_mode || ctx->stage_timing || ctx->legacy_stream) {
        vppx_set_error("vppx_fstream_create: the context must launch on a stream of its own, without sub-streams, graph mode or stage timing");
        return VPPX_E_INVALID_ARG;
    }
    {
        const int rc = vppx_check_hot_path_args_internal(vp, rp, batch, H, W, C); // (same codes and messages as the hot path itself)
        if (rc) return rc;
    }
    DevGuard g(ctx->device);
    vppx_fstream *fs = new vppx_fstream();
    fs->ctx = ctx;
    fs->prev_pipeline = ctx->pipeline;
    fs->have_op = op != nullptr;
    if (op) fs->op = *op;
    fs->vp = *vp;
    fs->vp.rand_offset = 0;
    fs->vp.per_frame_range = 1; // vpp() takes dmin / dmax from each frame's own hints (vpp_standalone.py:410-411)
    fs->rp = *rp;
    fs->batch = batch; fs->depth = depth; fs->H = H; fs->W = W; fs->C = C; fs->flags = flags;
    fs->px = (size_t)H * W;
    fs->img = fs->px * C;
    // planes of `batch` frames each, so that a batch goes up / comes down in one copy per plane and the hot path sees [B,H,W,C]
    size_t o = 0;
    fs->o_left = o;  o += up256(fs->img * batch);
    fs->o_right = o; o += up256(fs->img * batch);
    fs->o_hint = o;  o += up256(fs->px * 4 * batch);
    fs->o_gocc = o;  if (flags & VPPX_FS_GOCC) o += up256(fs->px * batch);
    fs->in_bytes = o;
    o = 0;
    fs->o_disp = o;  o += up256(fs->px * 4 * batch);
    fs->o_lv = o;    if (flags & VPPX_FS_PATTERNS) o += up256(fs->img * batch);
    fs->o_rv = o;    if (flags & VPPX_FS_PATTERNS) o += up256(fs->img * batch);
    fs->o_conf = o;  if (flags & VPPX_FS_MASK) o += up256(fs->px * batch);
    fs->o_draws = o; o += up256((size_t)8 * batch);
    fs->out_bytes = o;
    fs->slots.resize(depth);
    auto fail = [&](const char *what, hipError_t e) {
        vppx_set_error("vppx_fstream_create: %s failed: %s", what, hipGetErrorString(e));
        fs_free(fs);
        return e == hipErrorOutOfMemory ? VPPX_E_OOM : VPPX_E_HIP;
    };
    hipError_t e;
    // The two copy streams are created at the LOWEST priority: the runtime deals streams of one priority onto a pool of four hardware
    // queues round-robin, and with the context's three streams (and the caller's own) the copy-out stream landed in the queue of the
    // context's main stream -- every batch's lock-step launch then queued behind the previous batch's copy-out (0.7-1.0 ms per batch,
    // seen in the kernel trace's Queue_Id column).  Streams of another priority get queues of their own pool, and the copy-out
    // kernel's waves yield to the hot path's.
    int prio_least = 0, prio_greatest = 0;
    if ((e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest)) != hipSuccess) return fail("hipDeviceGetStreamPriorityRange", e);
    if ((e = hipStreamCreateWithPriority(&fs->s_in, hipStreamNonBlocking, prio_least)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithPriority(&fs->s_out, hipStreamNonBlocking, prio_least)) != hipSuccess) return fail("hipStreamCreate", e);
    for (auto &s : fs->slots) {
        if ((e = hipHostMalloc((void **)&s.h_in, fs->in_bytes, hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc", e);
        if ((e = hipHostMalloc((void **)&s.h_out, fs->out_bytes, hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc", e);
        if ((e = hipMalloc((void **)&s.d_in, fs->in_bytes)) != hipSuccess) return fail("hipMalloc", e);
        if ((e = hipMalloc((void **)&s.d_out, fs->out_bytes)) != hipSuccess) return fail("hipMalloc", e);
        if ((e = hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
        memset(s.h_out + fs->o_draws, 0, (size_t)8 * batch);
    }
    if (copy_threads < 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        copy_threads = hc >= 16 ? 4 : (hc >= 4 ? 2 : 1);
    }
    if (copy_threads > 16) copy_threads = 16;
    fs->pool = new CopyPool(copy_threads > 1 ? copy_threads - 1 : 0); // the pushing thread copies too
    ctx->pipeline = true;
    ctx->have_agg_done = false;
    *out = fs;
    return 0;
}

extern "C" void vppx_fstream_destroy(vppx_fstream *fs) { fs_free(fs); }

// Results leave the device through a kernel of a few workgroups that stores into the pinned ring (FS_COPY_WGS x 256 lanes, 16 bytes
// each per trip: a 41 MB batch in 0.7-1.0 ms alone, ~1.4 ms beside the sum / WTA kernel, of the ~4 ms the next batch computes).
// Why not hipMemcpyAsync: which engine a device -> pinned-host copy takes is the runtime's choice -- SDMA with ROCm 7.2's
// libamdhip64, a 256-workgroup blit kernel with the 7.0 one a process that imported torch has loaded -- and the blit kernel,
// saturating PCIe from every CU, stretched whatever front-stage kernel of the next batch ran beside it (pad+gray+census 66 -> 620 us,
// the batch period 4.1 -> 4.7 ms; tools/d2h_probe.hip, docs/NOTEBOOK.md round 6).  Few waves keep few PCIe writes in flight and
// leave the CUs to the hot path; 4, 8 and 16 workgroups, or the same lanes as one-wave workgroups on more CUs, measure alike.
#ifndef FS_COPY_WGS
#define FS_COPY_WGS 8
#endif
struct FsCopySegs {
    const uint4 *src[5];
    uint4 *dst[5];
    unsigned long long n16[5];
    int n;
};
__global__ void __launch_bounds__(256) fs_copy_out_kernel(FsCopySegs a)
{
    for (int s = 0; s < a.n; s++) {
        const uint4 *__restrict__ src = a.src[s];
        uint4 *__restrict__ dst = a.dst[s];
        for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < a.n16[s]; i += (unsigned long long)gridDim.x * 256ull) dst[i] = src[i];
    }
}

// Enqueue a submitted batch's copy-out on the copy-out stream: behind the batch's own kernels and, when asked, behind the
// aggregation of the call enqueued last on the context.
static int fs_copy_out(vppx_fstream *fs, Slot &s, bool behind_last_aggregation)
{
    vppx_ctx *ctx = fs->ctx;
    const int nb = s.nb;
    VPPX_HIP(hipStreamWaitEvent(fs->s_out, s.ev_done, 0));
    if (behind_last_aggregation) VPPX_HIP(hipStreamWaitEvent(fs->s_out, ctx->ev_agg_done, 0));
    // segments start on 256-byte boundaries of both buffers (same layout) and are padded to them: whole 16-byte pieces
    FsCopySegs a;
    a.n = 0;
    auto seg = [&](size_t off, size_t bytes) {
        a.src[a.n] = (const uint4 *)(s.d_out + off);
        a.dst[a.n] = (uint4 *)(s.h_out + off);
        a.n16[a.n] = (bytes + 15) / 16;
        a.n++;
    };
    seg(fs->o_disp, fs->px * 4 * nb);
    if (fs->flags & VPPX_FS_PATTERNS) {
        seg(fs->o_lv, fs->img * nb);
        seg(fs->o_rv, fs->img * nb);
    }
    if (fs->flags & VPPX_FS_MASK) seg(fs->o_conf, fs->px * nb);
    if (s.out_method_rnd) seg(fs->o_draws, (size_t)8 * nb);
    fs_copy_out_kernel<<<FS_COPY_WGS, 256, 0, fs->s_out>>>(a);
    VPPX_HIP(hipGetLastError());
    VPPX_HIP(hipEventRecord(s.ev_out, fs->s_out));
    s.out_pending = false;
    return 0;
}

// One batch through the hot path: upload (unless the inputs are still on the device: a re-run), kernels, download.
static int fs_submit(vppx_fstream *fs, Slot &s, bool upload)
{
    vppx_ctx *ctx = fs->ctx;
    const int nb = s.nb;
    if (upload) {
        // the slot's device buffers are free once its last batch's results have left them (pop has waited for that event on the
        // host already; the stream-side wait costs nothing and keeps the order explicit)
        if (s.submitted) VPPX_HIP(hipStreamWaitEvent(fs->s_in, s.ev_out, 0));
        VPPX_HIP(hipMemcpyAsync(s.d_in + fs->o_left, s.h_in + fs->o_left, fs->img * nb, hipMemcpyHostToDevice, fs->s_in));
        VPPX_HIP(hipMemcpyAsync(s.d_in + fs->o_right, s.h_in + fs->o_right, fs->img * nb, hipMemcpyHostToDevice, fs->s_in));
        VPPX_HIP(hipMemcpyAsync(s.d_in + fs->o_hint, s.h_in + fs->o_hint, fs->px * 4 * nb, hipMemcpyHostToDevice, fs->s_in));
        if (fs->flags & VPPX_FS_GOCC)
            VPPX_HIP(hipMemcpyAsync(s.d_in + fs->o_gocc, s.h_in + fs->o_gocc, fs->px * nb, hipMemcpyHostToDevice, fs->s_in));
        VPPX_HIP(hipEventRecord(s.ev_in, fs->s_in));
    }
    VppxVppParams vp = fs->vp;
    vp.seed = s.seed;
    // the hot path reports an earlier call's lost lock step instead of running (its caller could not tell otherwise): a stream
    // looks first, keeps going -- the context now rests on the line-parallel layout -- and re-runs what was in flight at the next pop
    if (vppx_lockstep_check_internal(ctx) != 0) fs->lost_pending = true;
    u8 *lv = (fs->flags & VPPX_FS_PATTERNS) ? s.d_out + fs->o_lv : nullptr;
    u8 *rv = (fs->flags & VPPX_FS_PATTERNS) ? s.d_out + fs->o_rv : nullptr;
    int rc = 0;
    for (int attempt = 0;; attempt++) {
        const long lost_before = ctx->lockstep_failures;
        rc = vppx_inputs_ready_event(ctx, (void *)s.ev_in);
        if (rc) return rc;
        ctx->draws_dst = (vp.method == VPPX_METHOD_RND) ? (unsigned long long *)(s.d_out + fs->o_draws) : nullptr;
        if (fs->have_op)
            rc = vppx_occ_vpp_rsgm_dev(ctx, &fs->op, &vp, &fs->rp, nb, fs->H, fs->W, fs->C, s.d_in + fs->o_left, s.d_in + fs->o_right,
                                       (const float *)(s.d_in + fs->o_hint), (fs->flags & VPPX_FS_MASK) ? s.d_out + fs->o_conf : nullptr, lv, rv,
                                       (float *)(s.d_out + fs->o_disp));
        else
            rc = vppx_vpp_rsgm_dev(ctx, &vp, &fs->rp, nb, fs->H, fs->W, fs->C, s.d_in + fs->o_left, s.d_in + fs->o_right,
                                   (const float *)(s.d_in + fs->o_hint), (fs->flags & VPPX_FS_GOCC) ? s.d_in + fs->o_gocc : nullptr, lv, rv,
                                   (float *)(s.d_out + fs->o_disp));
        ctx->draws_dst = nullptr;
        // A launch still in flight can lose its lock step between the look above and the hot path's own look, just before it queues
        // the aggregation (the deeper the ring, the more launches are in flight): the call then reports that instead of running.
        // Same treatment: note it, run the call again (each look consumes one report; at most `depth` launches are in flight).
        if (rc != 0 && ctx->lockstep_failures != lost_before && attempt < fs->depth) {
            fs->lost_pending = true;
            continue;
        }
        break;
    }
    if (rc) return rc;
    VPPX_HIP(hipEventRecord(s.ev_done, ctx->stream));
    s.out_method_rnd = vp.method == VPPX_METHOD_RND;
    s.out_pending = true;
    // Copy-outs ride in the NEXT batch, behind its aggregation: the lock-step launch goes at the pace of its slowest CU, and a CU
    // that also hosts copy-out waves (stores over PCIe back up its memory pipeline) is slow -- next to the copy-out the launch took
    // 2.2 ms instead of 1.65; the sum / WTA kernel beside it pays less (1.22 -> 1.5 ms).  So: what earlier batches still owe is
    // enqueued now, behind this call's aggregation event; this batch's own copy-out waits for the next submit -- or for the pop or
    // re-run that needs it first (fs_verify).  With two slots the host would sit out half a batch waiting for results that are held
    // back: a ring of two copies out at once.
    for (auto &o : fs->slots)
        if (o.out_pending && (&o != &s || fs->depth < 3)) {
            const int rc2 = fs_copy_out(fs, o, &o != &s && ctx->have_agg_done);
            if (rc2) return rc2;
        }
    s.submitted = true;
    s.verified = false;
    return 0;
}

static int fs_close_batch(vppx_fstream *fs)
{
    if (fs->fill_count == 0) return 0;
    Slot &s = fs->slots[fs->fill_batch % fs->depth];
    s.nb = fs->fill_count;
    s.seed = fs->vp.seed + (uint32_t)(fs->pushed - (uint64_t)fs->fill_count);
    const int rc = fs_submit(fs, s, true);
    if (rc) return rc;
    fs->sub_batches++;
    fs->fill_batch++;
    fs->fill_count = 0;
    return 0;
}

extern "C" int vppx_fstream_push(vppx_fstream *fs, const uint8_t *left, const uint8_t *right, const float *hints, const uint8_t *g_occ)
{
    if (!fs) { vppx_set_error("vppx_fstream_push: stream is NULL"); return VPPX_E_INVALID_ARG; }
    if (!left || !right || !hints || (((fs->flags & VPPX_FS_GOCC) != 0) != (g_occ != nullptr))) {
        vppx_set_error("vppx_fstream_push: left, right, hints%s", (fs->flags & VPPX_FS_GOCC) ? " and g_occ (VPPX_FS_GOCC)" : "; g_occ only with VPPX_FS_GOCC");
        return VPPX_E_INVALID_ARG;
    }
    DevGuard g(fs->ctx->device);
    Slot &s = fs->slots[fs->fill_batch % fs->depth];
    if (fs->fill_count == 0 && s.submitted && fs->fill_batch - fs->pop_batch >= (uint64_t)fs->depth) {
        // every slot holds a batch whose results have not been popped: its ring areas (and device inputs, kept for a re-run) are in use
        vppx_set_error("vppx_fstream_push: %d batches are waiting to be popped (depth %d); pop results first", fs->depth, fs->depth);
        return VPPX_E_INVALID_ARG;
    }
    const int i = fs->fill_count;
    CopyJob jobs[4] = {{s.h_in + fs->o_left + fs->img * i, left, fs->img},
                       {s.h_in + fs->o_right + fs->img * i, right, fs->img},
                       {s.h_in + fs->o_hint + fs->px * 4 * i, hints, fs->px * 4},
                       {g_occ ? s.h_in + fs->o_gocc + fs->px * i : nullptr, g_occ, fs->px}};
    fs->pool->copy(jobs, 4);
    fs->fill_count++;
    fs->pushed++;
    if (fs->fill_count == fs->batch) return fs_close_batch(fs);
    return 0;
}

extern "C" int vppx_fstream_flush(vppx_fstream *fs)
{
    if (!fs) { vppx_set_error("vppx_fstream_flush: stream is NULL"); return VPPX_E_INVALID_ARG; }
    DevGuard g(fs->ctx->device);
    return fs_close_batch(fs);
}

extern "C" int vppx_fstream_counts(vppx_fstream *fs, int64_t *pushed, int64_t *filling, int64_t *unpopped, int64_t *reruns)
{
    if (!fs) { vppx_set_error("vppx_fstream_counts: stream is NULL"); return VPPX_E_INVALID_ARG; }
    if (pushed) *pushed = (int64_t)fs->pushed;
    if (filling) *filling = fs->fill_count;
    if (unpopped) {
        int64_t n = 0;
        for (uint64_t b = fs->pop_batch; b < fs->sub_batches; b++) n += fs->slots[b % fs->depth].nb;
        *unpopped = n - fs->pop_idx;
    }
    if (reruns) *reruns = (int64_t)fs->reruns;
    return 0;
}

// The batch at the head of the queue has come down: has it (or anything in flight) lost a lock step?  Then everything
// submitted and not popped is run again, in order, from the device inputs the slots still hold.
static int fs_verify(vppx_fstream *fs, Slot &head)
{
    vppx_ctx *ctx = fs->ctx;
    if (head.out_pending) { // no later batch has been submitted: nothing to hide behind
        const int rc = fs_copy_out(fs, head, false);
        if (rc) return rc;
    }
    VPPX_HIP(hipEventSynchronize(head.ev_out));
    if (vppx_lockstep_check_internal(ctx) == 0 && !fs->lost_pending) {
        head.verified = true;
        return 0;
    }
    fs->lost_pending = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        fs->lost_pending = false;
        VPPX_HIP(hipStreamSynchronize(ctx->stream));
        VPPX_HIP(hipStreamSynchronize(fs->s_out));
        (void)vppx_lockstep_check_internal(ctx); // whatever else was lost meanwhile is re-run as well
        ctx->have_agg_done = false;              // the next front stage waits for the whole launch stream again
        for (uint64_t b = fs->pop_batch; b < fs->sub_batches; b++) {
            Slot &s = fs->slots[b % fs->depth];
            int rc = fs_submit(fs, s, false);
            if (rc) return rc;
            fs->reruns++;
        }
        for (auto &o : fs->slots)
            if (o.out_pending) {
                const int rc = fs_copy_out(fs, o, false);
                if (rc) return rc;
            }
        VPPX_HIP(hipStreamSynchronize(fs->s_out));
        if (vppx_lockstep_check_internal(ctx) == 0 && !fs->lost_pending) {
            for (uint64_t b = fs->pop_batch; b < fs->sub_batches; b++) fs->slots[b % fs->depth].verified = true;
            return 0;
        }
    }
    return VPPX_E_HIP; // (the message of the last check stands)
}

extern "C" int vppx_fstream_pop(vppx_fstream *fs, float *disp_out, uint8_t *l_vpp_out, uint8_t *r_vpp_out, uint8_t *conf_out,
                                uint64_t *draws_out, int *got)
{
    if (!fs || !got) { vppx_set_error("vppx_fstream_pop: NULL argument"); return VPPX_E_INVALID_ARG; }
    *got = 0;
    if (fs->pop_batch == fs->sub_batches) return 0; // nothing submitted is outstanding (frames of an unfinished batch: vppx_fstream_flush)
    if (!disp_out) { vppx_set_error("vppx_fstream_pop: disp_out is NULL"); return VPPX_E_INVALID_ARG; }
    if ((l_vpp_out || r_vpp_out) && !(fs->flags & VPPX_FS_PATTERNS)) { vppx_set_error("vppx_fstream_pop: the stream was created without VPPX_FS_PATTERNS"); return VPPX_E_INVALID_ARG; }
    if (conf_out && !(fs->flags & VPPX_FS_MASK)) { vppx_set_error("vppx_fstream_pop: the stream was created without VPPX_FS_MASK"); return VPPX_E_INVALID_ARG; }
    DevGuard g(fs->ctx->device);
    Slot &s = fs->slots[fs->pop_batch % fs->depth];
    if (!s.verified) {
        const int rc = fs_verify(fs, s);
        if (rc) return rc;
    }
    const int i = fs->pop_idx;
    CopyJob jobs[4] = {{disp_out, s.h_out + fs->o_disp + fs->px * 4 * i, fs->px * 4},
                       {l_vpp_out, s.h_out + fs->o_lv + fs->img * i, fs->img},
                       {r_vpp_out, s.h_out + fs->o_rv + fs->img * i, fs->img},
                       {conf_out, s.h_out + fs->o_conf + fs->px * i, fs->px}};
    fs->pool->copy(jobs, 4);
    if (draws_out) *draws_out = fs->vp.method == VPPX_METHOD_RND ? ((const uint64_t *)(s.h_out + fs->o_draws))[i] : 0;
    *got = 1;
    if (++fs->pop_idx == s.nb) {
        fs->pop_idx = 0;
        fs->pop_batch++;
    }
    return 0;
}
