/* fstream_demo.c -- the C-ABI of the frame stream (include/vppx.h) used from plain C, the way a C / C++ / cgo / JNI host of the
 * reference's loop (test.py:291-311) would: no Python, no torch.  Makes N small frames from a 32-bit LCG, pushes them one at a time,
 * pops the disparity maps in input order and prints one FNV-1a checksum per frame; tests/test_gpu_cabi.py compiles this with gcc, runs
 * it on the GPU box and compares the checksums with vppstereo_amd.pipeline.FrameStream on the same frames.
 * Build: gcc -std=c99 -O1 -I include tests/c/fstream_demo.c -o demo -L vppstereo_amd -lvppx -Wl,-rpath,$PWD/vppstereo_amd
 * Usage: demo <frames> <H> <W> <dmax> <batch> <seed> */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vppx.h"

static uint32_t lcg_state;
static uint32_t lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state; }

static uint64_t fnv1a(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ < 0) { fprintf(stderr, "%s: %d: %s\n", #call, rc_, vppx_last_error()); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s frames H W dmax batch seed\n", argv[0]); return 2; }
    const int n = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), D = atoi(argv[4]), batch = atoi(argv[5]);
    const uint32_t seed = (uint32_t)strtoul(argv[6], NULL, 10);
    const size_t px = (size_t)H * W;
    uint8_t *left = malloc(px * 3), *right = malloc(px * 3);
    float *hints = malloc(px * sizeof(float)), *disp = malloc(px * sizeof(float));
    vppx_ctx *ctx = NULL;
    vppx_fstream *fs = NULL;
    VppxVppParams vp;
    VppxRsgmParams rp;
    VppxOccParams op;
    CHECK(vppx_create(&ctx, -1));
    vppx_vpp_params_default(&vp);
    vppx_rsgm_params_default(&rp);
    vppx_occ_params_default(&op);
    vp.seed = seed;
    rp.dmax = D;
    CHECK(vppx_fstream_create(ctx, &op, &vp, &rp, batch, 3, H, W, 3, 0, 2, &fs)); /* a ring of three batches (include/vppx.h) */
    int popped = 0, got = 0;
    for (int f = 0; f < n; f++) {
        /* the right view is the left one shifted by 5 columns (a plane at disparity 5) plus noise in the low bits; 4 % hints near 5 */
        lcg_state = 12345u + 977u * (uint32_t)f;
        for (size_t i = 0; i < px * 3; i++) left[i] = (uint8_t)(lcg() >> 24);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                for (int c = 0; c < 3; c++) {
                    const int xs = x + 5 < W ? x + 5 : W - 1;
                    right[((size_t)y * W + x) * 3 + c] = (uint8_t)(left[((size_t)y * W + xs) * 3 + c] ^ (lcg() >> 31));
                }
        for (size_t i = 0; i < px; i++) {
            const uint32_t r = lcg();
            hints[i] = (r >> 24) < 10 ? 4.0f + (float)((r >> 16) & 3) * 0.5f : 0.0f;
        }
        CHECK(vppx_fstream_push(fs, left, right, hints, NULL));
        /* keep two batches in flight, take the rest: the oldest batch's copy-out was released when the newest was submitted */
        int64_t unpopped = 0, filling = 0;
        CHECK(vppx_fstream_counts(fs, NULL, &filling, &unpopped, NULL));
        while (unpopped > 2 * batch) {
            CHECK(vppx_fstream_pop(fs, disp, NULL, NULL, NULL, NULL, &got));
            if (!got) break;
            printf("%d %016llx\n", popped++, (unsigned long long)fnv1a(disp, px * sizeof(float)));
            unpopped--;
        }
    }
    CHECK(vppx_fstream_flush(fs));
    for (;;) {
        CHECK(vppx_fstream_pop(fs, disp, NULL, NULL, NULL, NULL, &got));
        if (!got) break;
        printf("%d %016llx\n", popped++, (unsigned long long)fnv1a(disp, px * sizeof(float)));
    }
    vppx_fstream_destroy(fs);
    vppx_destroy(ctx);
    free(left); free(right); free(hints); free(disp);
    return popped == n ? 0 : 3;
}
