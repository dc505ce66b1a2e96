// pingpong_bench.hip -- latency of a flag hand-off between two workgroups of the SAME XCD (blocks 0 and 8 of the grid)
// for the cache-scope bits a store / a polling load can carry on gfx950.  Prints round trips per microsecond pair and
// whether the protocol completed (bounded spins: a combination that never sees the other side's store reports FAIL).
// Build: hipcc --offload-arch=gfx950 -O2 tools/pingpong_bench.hip -o tools/bin/pingpong_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ROUNDS 2000
#define LIMIT (1 << 18)

template <int SMODE, int LMODE>
__device__ __forceinline__ void st(uint32_t *p, uint32_t v)
{
    if (SMODE == 0) asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
    if (SMODE == 1) asm volatile("global_store_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
    if (SMODE == 2) asm volatile("global_store_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
    if (SMODE == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
template <int LMODE>
__device__ __forceinline__ uint32_t ld(const uint32_t *p)
{
    uint32_t v;
    if (LMODE == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LMODE == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LMODE == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LMODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LMODE == 4) asm volatile("buffer_inv sc1\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LMODE == 5) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int SMODE, int LMODE>
__global__ void pingpong(uint32_t *flags, uint32_t *result, int other)
{
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == other ? 1 : -1);
    if (me < 0) return;
    uint32_t *mine = flags + me * 64, *theirs = flags + (1 - me) * 64; // separate cache lines
    bool ok = true;
    for (uint32_t r = 1; r <= ROUNDS && ok; r++) {
        if (me == 0) st<SMODE, LMODE>(mine, r);
        int spins = 0;
        while (ld<LMODE>(theirs) < r && ++spins < LIMIT) __builtin_amdgcn_s_sleep(1);
        ok = spins < LIMIT;
        if (me == 1) st<SMODE, LMODE>(mine, r);
    }
    result[me] = ok ? 1u : 0u;
}

template <int SMODE, int LMODE>
static void run(uint32_t *flags, uint32_t *result, int other, const char *sname, const char *lname)
{
    hipMemset(flags, 0, 1024);
    hipMemset(result, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    pingpong<SMODE, LMODE><<<other + 1, 64>>>(flags, result, other);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[2];
    hipMemcpy(h, result, 8, hipMemcpyDeviceToHost);
    printf("store %-8s load %-14s other block %2d: %s  %.2f us per round trip\n", sname, lname, other, (h[0] && h[1]) ? "ok  " : "FAIL", ms * 1e3 / ROUNDS);
}

int main()
{
    uint32_t *flags, *result;
    hipMalloc(&flags, 1024); hipMalloc(&result, 8);
    for (int other : {8, 1}) { // block 8: same XCD as block 0; block 1: the next XCD
#define R(S, L, SN, LN) run<S, L>(flags, result, other, SN, LN)
        R(0, 1, "plain", "sc0"); R(0, 2, "plain", "sc1"); R(0, 3, "plain", "sc0 sc1"); R(0, 4, "plain", "inv+plain"); R(0, 5, "plain", "nt");
        R(1, 1, "sc0", "sc0"); R(2, 2, "sc1", "sc1"); R(3, 3, "sc0 sc1", "sc0 sc1"); R(2, 1, "sc1", "sc0"); R(0, 0, "plain", "plain");
    }
    return 0;
}
