// bank_bench.hip -- does the VGPR bank (register number mod 4) of the three source operands of a VOP3 / VOP3P instruction
// change its issue cost on gfx950?  Same-bank vs spread operands, 8 independent chains, N waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/bank_bench.hip -o tools/bin/bank_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define ITER 256
#define K(NAME, BODY)                                                                                     \
    __global__ void __launch_bounds__(256) NAME(unsigned *out, unsigned long long *cyc)                   \
    {                                                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int it = 0; it < ITER; it++) {                                                               \
            asm volatile(BODY BODY BODY BODY ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", \
                         "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40"); \
        }                                                                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                    \
        out[blockIdx.x * 256 + threadIdx.x] = 0;                                                          \
    }
// 8 instructions per body; dst distinct; sources: same bank (8,12,16 ...) or spread (9,14,19)
K(add3_same, "v_add3_u32 v0, v8, v12, v16\n v_add3_u32 v1, v20, v24, v28\n v_add3_u32 v2, v8, v12, v16\n v_add3_u32 v3, v20, v24, v28\n v_add3_u32 v4, v8, v12, v16\n v_add3_u32 v5, v20, v24, v28\n v_add3_u32 v6, v8, v12, v16\n v_add3_u32 v7, v20, v24, v28\n")
K(add3_spread, "v_add3_u32 v0, v8, v13, v18\n v_add3_u32 v1, v20, v25, v30\n v_add3_u32 v2, v8, v13, v18\n v_add3_u32 v3, v20, v25, v30\n v_add3_u32 v4, v8, v13, v18\n v_add3_u32 v5, v20, v25, v30\n v_add3_u32 v6, v8, v13, v18\n v_add3_u32 v7, v20, v25, v30\n")
K(min3_same, "v_pk_minimum3_f16 v0, v8, v12, v16\n v_pk_minimum3_f16 v1, v20, v24, v28\n v_pk_minimum3_f16 v2, v8, v12, v16\n v_pk_minimum3_f16 v3, v20, v24, v28\n v_pk_minimum3_f16 v4, v8, v12, v16\n v_pk_minimum3_f16 v5, v20, v24, v28\n v_pk_minimum3_f16 v6, v8, v12, v16\n v_pk_minimum3_f16 v7, v20, v24, v28\n")
K(min3_spread, "v_pk_minimum3_f16 v0, v8, v13, v18\n v_pk_minimum3_f16 v1, v20, v25, v30\n v_pk_minimum3_f16 v2, v8, v13, v18\n v_pk_minimum3_f16 v3, v20, v25, v30\n v_pk_minimum3_f16 v4, v8, v13, v18\n v_pk_minimum3_f16 v5, v20, v25, v30\n v_pk_minimum3_f16 v6, v8, v13, v18\n v_pk_minimum3_f16 v7, v20, v25, v30\n")
K(pkmin_same, "v_pk_min_u16 v0, v8, v12\n v_pk_min_u16 v1, v20, v24\n v_pk_min_u16 v2, v8, v12\n v_pk_min_u16 v3, v20, v24\n v_pk_min_u16 v4, v8, v12\n v_pk_min_u16 v5, v20, v24\n v_pk_min_u16 v6, v8, v12\n v_pk_min_u16 v7, v20, v24\n")
K(pkmin_spread, "v_pk_min_u16 v0, v8, v13\n v_pk_min_u16 v1, v20, v25\n v_pk_min_u16 v2, v8, v13\n v_pk_min_u16 v3, v20, v25\n v_pk_min_u16 v4, v8, v13\n v_pk_min_u16 v5, v20, v25\n v_pk_min_u16 v6, v8, v13\n v_pk_min_u16 v7, v20, v25\n")
K(add_same, "v_add_u32 v0, v8, v12\n v_add_u32 v1, v20, v24\n v_add_u32 v2, v8, v12\n v_add_u32 v3, v20, v24\n v_add_u32 v4, v8, v12\n v_add_u32 v5, v20, v24\n v_add_u32 v6, v8, v12\n v_add_u32 v7, v20, v24\n")
K(add_spread, "v_add_u32 v0, v8, v13\n v_add_u32 v1, v20, v25\n v_add_u32 v2, v8, v13\n v_add_u32 v3, v20, v25\n v_add_u32 v4, v8, v13\n v_add_u32 v5, v20, v25\n v_add_u32 v6, v8, v13\n v_add_u32 v7, v20, v25\n")
// a dependent chain: each instruction reads the previous result (latency per instruction at low occupancy)
K(add3_chain, "v_add3_u32 v0, v0, v13, v18\n v_add3_u32 v0, v0, v13, v18\n v_add3_u32 v0, v0, v13, v18\n v_add3_u32 v0, v0, v13, v18\n v_add3_u32 v0, v0, v13, v18\n v_add3_u32 v0, v0, v13, v18\n v_add3_u32 v0, v0, v13, v18\n v_add3_u32 v0, v0, v13, v18\n")
K(min3_chain, "v_pk_minimum3_f16 v0, v0, v13, v18\n v_pk_minimum3_f16 v0, v0, v13, v18\n v_pk_minimum3_f16 v0, v0, v13, v18\n v_pk_minimum3_f16 v0, v0, v13, v18\n v_pk_minimum3_f16 v0, v0, v13, v18\n v_pk_minimum3_f16 v0, v0, v13, v18\n v_pk_minimum3_f16 v0, v0, v13, v18\n v_pk_minimum3_f16 v0, v0, v13, v18\n")
template <typename F>
static void run(const char *name, F fn, int blocks)
{
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    fn<<<blocks, 256>>>(out, cyc); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); fn<<<blocks, 256>>>(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninst = (double)ITER * 32;                 // per wave
    const double waves_per_simd = blocks * 4 / 1024.0;
    printf("%-14s blocks %5d (%.0f waves/SIMD): %.2f ns per wave-instruction and SIMD\n", name, blocks, waves_per_simd,
           ms * 1e6 / (ninst * waves_per_simd));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int blocks : {256, 512, 2048}) {
        run("add3_same", add3_same, blocks); run("add3_spread", add3_spread, blocks);
        run("min3_same", min3_same, blocks); run("min3_spread", min3_spread, blocks);
        run("pkmin_same", pkmin_same, blocks); run("pkmin_spread", pkmin_spread, blocks);
        run("add_same", add_same, blocks); run("add_spread", add_spread, blocks);
        run("add3_chain", add3_chain, blocks); run("min3_chain", min3_chain, blocks);
    }
    return 0;
}
