// tear_bench.hip -- is a lane's aligned 16-byte store seen whole or not at all by a 16-byte sc1 load from another CU of the same XCD?
// (the lock-step kernel tags BOTH ends of every 16-byte piece of an edge record; one tag would save ~30 of its 698
// instructions per row if pieces cannot tear.)  Writer blocks b store {k, k, k, k} (k counting up) into 64 x NSLOT slots with the store
// the kernel uses (a plain global_store_dwordx4); reader blocks b + 8 (same XCD: block ids go round-robin over the 8 XCDs) load the same
// slots with buffer_load_dwordx4 ... sc1 and count pieces whose four dwords differ.  Control: the same with four dword stores, which MUST
// tear -- the test has to be able to see a tear.  Prints pieces read, tears, and distinct values seen (the readers do see new data).
// Build: hipcc --offload-arch=gfx950 -O2 tools/tear_bench.hip -o tools/bin/tear_bench        Usage: tear_bench [iterations per reader lane]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define NSLOT 8 // 16-byte slots per lane and pair (neighbouring pieces of one record, like the kernel's)

template <bool WHOLE>
__global__ void __launch_bounds__(64) tear_kernel(uint32_t *buf, unsigned long long *stats, int npairs, int iters, volatile int *stop)
{
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3; // j even: writer, j odd: its reader (8 block ids apart: same XCD)
    const int pair = (j >> 1) * 8 + xcd;
    if (pair >= npairs) return;
    uint32_t *base = buf + ((size_t)pair * 64 + threadIdx.x) * NSLOT * 4;
    if ((j & 1) == 0) { // writer: until the readers are done
        for (uint32_t k = 1; !*stop; k++) {
#pragma unroll
            for (int s = 0; s < NSLOT; s++) {
                const uint32_t v = k * NSLOT + s;
                if (WHOLE) {
                    *(u32x4 *)(base + 4 * s) = u32x4{v, v, v, v};
                } else {
                    asm volatile("global_store_dword %0, %1, off\n global_store_dword %0, %1, off offset:4\n"
                                 "global_store_dword %0, %1, off offset:8\n global_store_dword %0, %1, off offset:12" ::"v"(base + 4 * s), "v"(v) : "memory");
                }
            }
            if ((k & 1023u) == 0) __builtin_amdgcn_s_sleep(1);
        }
    } else {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, NSLOT * 16, 0x00020000);
        unsigned long long tears = 0, changes = 0;
        uint32_t last = 0;
        for (int it = 0; it < iters; it++) {
            asm volatile("" ::: "memory"); // (every iteration loads anew)
#pragma unroll
            for (int s = 0; s < NSLOT; s++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * s, 0, 16 /* sc1 */);
                tears += (v.x != v.y || v.x != v.z || v.x != v.w) ? 1 : 0;
                if (s == 0) { changes += v.x != last; last = v.x; }
            }
        }
        atomicAdd(&stats[0], (unsigned long long)iters * NSLOT);
        atomicAdd(&stats[1], tears);
        atomicAdd(&stats[2], changes);
        if (threadIdx.x == 0 && atomicAdd(&stats[3], 1ull) + 1 == (unsigned long long)npairs) *stop = 1; // the last reader block lets the writers go
    }
}

template <bool WHOLE>
static void run(const char *name, int npairs, int iters)
{
    uint32_t *buf;
    unsigned long long *stats;
    int *stop;
    hipMalloc(&buf, (size_t)npairs * 64 * NSLOT * 16);
    hipMemset(buf, 0, (size_t)npairs * 64 * NSLOT * 16);
    hipMalloc(&stats, 32);
    hipMemset(stats, 0, 32);
    hipHostMalloc(&stop, 4, hipHostMallocMapped);
    *stop = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    tear_kernel<WHOLE><<<((npairs + 7) / 8) * 16, 64>>>(buf, stats, npairs, iters, stop);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4];
    hipMemcpy(h, stats, 32, hipMemcpyDeviceToHost);
    printf("%-34s %3d pairs: %.3e pieces read, %llu torn, %.3e new values seen at slot 0, %.1f ms\n", name, npairs, (double)h[0], h[1], (double)h[2], ms);
    hipFree(buf); hipFree(stats); hipHostFree(stop);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    run<false>("control: four dword stores", 64, iters / 8);
    run<true>("one global_store_dwordx4", 64, iters);
    run<true>("one global_store_dwordx4", 256, iters);
    return 0;
}
