// tools/d2h_probe.hip -- which engine does the runtime take for a device -> pinned-host hipMemcpyAsync, and what does a
// concurrent write-heavy kernel pay for it?
//   hipcc --offload-arch=gfx950 -O2 tools/d2h_probe.hip -o tools/bin/d2h_probe
//   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/d2h -o d2h -- tools/bin/d2h_probe
// Cases (each prints the wall time of a 256 MB fill kernel alone and next to a 40 MB D2H, and the D2H's own time):
//   A  copy on its own stream, nothing before it
//   B  copy stream waits on an event of the compute stream (what the frame stream does)
//   C  as B, the host buffer allocated hipHostMallocNonCoherent
//   D  as B, the copy made by a small kernel of this file (32 workgroups) that stores to the mapped host buffer
// The trace tells blit kernel (__amd_rocclr_copyBuffer in the kernel trace) from SDMA (a MEMORY_COPY record).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(uint4 *p, size_t n, unsigned v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v, v, v);
}
__global__ void tiny_kernel(unsigned *p) { if (threadIdx.x == 0) p[0] = 1; }
__global__ void copy_out_kernel(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t NB = 40u << 20, FB = 256u << 20;
    hipStream_t sc, sx;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sx, hipStreamNonBlocking));
    uint4 *d_fill, *d_src;
    void *h_def, *h_nc;
    unsigned *d_flag;
    CK(hipMalloc(&d_fill, FB)); CK(hipMalloc(&d_src, NB)); CK(hipMalloc(&d_flag, 4));
    CK(hipHostMalloc(&h_def, NB, hipHostMallocDefault));
    CK(hipHostMalloc(&h_nc, NB, hipHostMallocNonCoherent));
    hipEvent_t ev, e0, e1, c0, c1;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    auto fill = [&](float *ms) {
        CK(hipEventRecord(e0, sc));
        fill_kernel<<<2048, 256, 0, sc>>>(d_fill, FB / 16, 7u);
        CK(hipEventRecord(e1, sc));
        CK(hipStreamSynchronize(sc));
        CK(hipEventElapsedTime(ms, e0, e1));
    };
    float alone = 0;
    for (int i = 0; i < 3; i++) fill(&alone);
    printf("fill alone %.3f ms (%.0f GB/s)\n", alone, FB / alone / 1e6);
    for (int c = 0; c < 4; c++) {
        for (int rep = 0; rep < 3; rep++) {
            void *h = c == 2 ? h_nc : h_def;
            if (c >= 1) {
                tiny_kernel<<<1, 64, 0, sc>>>(d_flag);
                CK(hipEventRecord(ev, sc));
                CK(hipStreamWaitEvent(sx, ev, 0));
            }
            CK(hipEventRecord(c0, sx));
            if (c == 3) copy_out_kernel<<<32, 256, 0, sx>>>(d_src, (uint4 *)h, NB / 16);
            else CK(hipMemcpyAsync(h, d_src, NB, hipMemcpyDeviceToHost, sx));
            CK(hipEventRecord(c1, sx));
            float f = 0, cp = 0;
            fill(&f);
            CK(hipStreamSynchronize(sx));
            CK(hipEventElapsedTime(&cp, c0, c1));
            printf("case %c rep %d: fill beside the copy %.3f ms (alone %.3f), copy %.3f ms (%.1f GB/s)\n", 'A' + c, rep, f, alone, cp, NB / cp / 1e6);
        }
    }
    // G: the copy-out kernel at 1 .. 256 workgroups beside 40 fills (2 ms of a write-bound kernel): what the writer pays and what the
    //    copy gets.  H: the same beside the runtime's own D2H (SDMA in this process).
    {
        auto fills = [&](float *ms) {
            CK(hipEventRecord(e0, sc));
            for (int i = 0; i < 40; i++) fill_kernel<<<2048, 256, 0, sc>>>(d_fill, FB / 16, 7u);
            CK(hipEventRecord(e1, sc));
            CK(hipStreamSynchronize(sc));
            CK(hipEventElapsedTime(ms, e0, e1));
        };
        float base = 0;
        fills(&base); fills(&base);
        printf("40 fills alone %.3f ms\n", base);
        const int wgs[] = {1, 2, 4, 8, 16, 32, 64, 256, 0};
        for (int wi = 0; wi < 9; wi++)
            for (int thr = 64; thr <= 256; thr *= 4) {
                if (wgs[wi] == 0 && thr != 64) continue;
                float f = 0, cp = 0;
                CK(hipEventRecord(c0, sx));
                if (wgs[wi]) copy_out_kernel<<<wgs[wi], thr, 0, sx>>>(d_src, (uint4 *)h_def, NB / 16);
                else CK(hipMemcpyAsync(h_def, d_src, NB, hipMemcpyDeviceToHost, sx));
                CK(hipEventRecord(c1, sx));
                fills(&f);
                CK(hipStreamSynchronize(sx));
                CK(hipEventElapsedTime(&cp, c0, c1));
                printf("copy-out %3d workgroups x %3d threads: 40 fills %.3f ms (alone %.3f, +%.0f %%), copy %.3f ms (%.1f GB/s)\n", wgs[wi], thr, f, base,
                       (f / base - 1) * 100, cp, NB / cp / 1e6);
            }
    }
    // E: three 25 MB H2D copies are enqueued on a third stream just before the D2H is enqueued (the frame stream's submit order);
    //    the D2H itself runs 5 ms later, behind a long kernel, when the H2D copies are long done.  F: the D2H enqueued first.
    hipStream_t sh;
    CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    void *h_in, *d_in;
    CK(hipHostMalloc(&h_in, 75u << 20, hipHostMallocDefault));
    CK(hipMalloc(&d_in, 75u << 20));
    for (int c = 4; c < 6; c++)
        for (int rep = 0; rep < 3; rep++) {
            auto h2d = [&]() {
                for (int i = 0; i < 3; i++)
                    CK(hipMemcpyAsync((char *)d_in + (size_t)i * (25u << 20), (char *)h_in + (size_t)i * (25u << 20), 25u << 20, hipMemcpyHostToDevice, sh));
            };
            if (c == 4) h2d();
            for (int i = 0; i < 100; i++) fill_kernel<<<2048, 256, 0, sc>>>(d_fill, FB / 16, 7u);
            CK(hipEventRecord(ev, sc));
            CK(hipStreamWaitEvent(sx, ev, 0));
            CK(hipEventRecord(c0, sx));
            CK(hipMemcpyAsync(h_def, d_src, NB, hipMemcpyDeviceToHost, sx));
            CK(hipEventRecord(c1, sx));
            if (c == 5) h2d();
            float f = 0, cp = 0;
            fill(&f);
            CK(hipStreamSynchronize(sx));
            CK(hipStreamSynchronize(sh));
            CK(hipEventElapsedTime(&cp, c0, c1));
            printf("case %c rep %d: fill beside the copy %.3f ms (alone %.3f), copy %.3f ms (%.1f GB/s)\n", 'A' + c, rep, f, alone, cp, NB / cp / 1e6);
        }
    return 0;
}
