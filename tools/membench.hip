// tools/membench.hip -- achievable HBM bandwidth on the box (calibration for the roofline):
// linear write, linear read, copy, and a "strided streams" write that mimics the aggregation
// kernel's store pattern (many waves, each storing 768 B..3 KB pieces one image row apart).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_write(u32x4 *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    u32x4 v = {1, 2, 3, (unsigned)i};
    for (; i < n; i += st) p[i] = v;
}
__global__ void k_read(const u32x4 *p, size_t n, unsigned *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += st) { u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) *out = acc;
}
__global__ void k_copy(const u32x4 *s, u32x4 *d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) d[i] = s[i];
}
// nvol volumes of rows x rowbytes; a wave owns `piece` bytes of every row and walks the rows
__global__ void k_strided(unsigned char *base, size_t volbytes, int nvol, int rows, int rowbytes, int piece) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int pieces_per_row = rowbytes / piece;
    const size_t vol = wave / pieces_per_row;
    if (vol >= (size_t)nvol) return;
    const int pc = wave % pieces_per_row;
    unsigned char *p = base + vol * volbytes + (size_t)pc * piece + (size_t)lane * (piece / 64);
    u32x4 v = {1, 2, 3, 4};
    for (int r = 0; r < rows; r++) {
        for (int b = 0; b < piece / 64; b += 16) *(u32x4 *)(p + b) = v;
        p += rowbytes;
    }
}
int main() {
    const size_t bytes = (size_t)6 << 30;
    void *a, *b; unsigned *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    const size_t n = bytes / 16;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0)); k_write<<<256 * 8, 256>>>((u32x4 *)a, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("write  %.1f GB/s\n", bytes / ms / 1e6);
        CK(hipEventRecord(e0)); k_read<<<256 * 8, 256>>>((const u32x4 *)a, n, o); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("read   %.1f GB/s\n", bytes / ms / 1e6);
        CK(hipEventRecord(e0)); k_copy<<<256 * 8, 256>>>((const u32x4 *)a, (u32x4 *)b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("copy   %.1f GB/s (read+write bytes)\n", 2.0 * bytes / ms / 1e6);
        CK(hipEventRecord(e0)); CK(hipMemsetAsync(a, 0, bytes, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) printf("memset %.1f GB/s\n", bytes / ms / 1e6);
    }
    // strided: 60 volumes (8 dirs x ~8 frames) of 544 rows x 184320 B (960 px x 192 B)
    const int rows = 544, rowbytes = 960 * 192;
    const size_t volbytes = (size_t)rows * rowbytes;
    const int nvol = (int)(bytes / volbytes);
    for (int piece : {768, 3072, 12288}) {
        const size_t waves = (size_t)nvol * (rowbytes / piece);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            k_strided<<<(unsigned)((waves * 64 + 255) / 256), 256>>>((unsigned char *)a, volbytes, nvol, rows, rowbytes, piece);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("strided piece=%5d B  waves=%zu  %.1f GB/s\n", piece, waves, (double)nvol * volbytes / ms / 1e6);
    }
    return 0;
}
