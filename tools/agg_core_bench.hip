// agg_core_bench.hip -- the aggregation kernel's per-step arithmetic (cost XORs + fused min-plus update + min
// reduction + byte packing) on registers only: no operand loads, no volume stores.  Tells the issue-bound time of
// the update itself apart from everything memory does to the real kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vppstereo_amd/csrc tools/agg_core_bench.hip -o tools/bin/agg_core_bench
#include "../vppstereo_amd/csrc/rsgm_kernels.hip"

template <int GW, int DPL>
__global__ void __launch_bounds__(256) core_kernel(u32 *out, int nsteps, u32 seed)
{
    constexpr int NP = DPL / 2;
    __shared__ u32 s_lut[256];
    s_lut[threadIdx.x] = pk_splat(17u + (threadIdx.x & 15));
    __syncthreads();
    const int lg = threadIdx.x % GW;
    u32 L[NP], inact[NP], w[DPL];
    for (int i = 0; i < NP; i++) { L[i] = 0; inact[i] = 0; }
    u32 clv = threadIdx.x * 2654435761u + seed;
    for (int j = 0; j < DPL; j++) w[j] = clv * (2 * j + 3);
    u32 minpk = 0, acc = 0;
    int prevI = threadIdx.x & 255;
    const u32 P1pk = pk_splat(11);
    for (int t = 0; t < nsteps; t++) {
        u32 X0[NP], X1[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            X0[i] = (clv ^ w[DPL - 1 - 2 * i]) & 0xFFFFFFu;
            X1[i] = (clv ^ w[DPL - 2 - 2 * i]) & 0xFFFFFFu;
        }
        const int I = (prevI + (int)(minpk & 7u)) & 255;
        const u32 di = __builtin_amdgcn_sad_u8((u32)I, (u32)prevI, 0u);
        const u32 P2pk = s_lut[di];
        sgm_update<NP, true, GW, true>(L, X0, X0, X1, P1pk, P2pk, minpk, inact, lg == 0, lg == GW - 1);
        u32 bw[NP / 2];
#pragma unroll
        for (int i = 0; i + 1 < NP; i += 2) bw[i / 2] = __builtin_amdgcn_perm(L[i + 1], L[i], 0x06040200u);
#pragma unroll
        for (int i = 0; i < NP / 2; i++) acc ^= bw[i];
        // next step's "operands": cheap full-rate mixing so that nothing is loop invariant
        clv = clv + 0x9E3779B9u;
#pragma unroll
        for (int j = 0; j < DPL; j++) w[j] ^= clv;
        prevI = I;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ minpk;
}

int main(int argc, char **argv)
{
    const int nsteps = 4096;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    u32 *out;
    hipMalloc(&out, (size_t)ncu * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps++) { // waves per SIMD = 256-thread blocks per CU
        if (wps > 4 && wps != 5 && wps != 8) continue;
        const int grid = ncu * wps;
        core_kernel<8, 24><<<grid, 256>>>(out, 64, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        core_kernel<8, 24><<<grid, 256>>>(out, nsteps, 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // the real launch at B=32 walks 16.3 k wave-steps per SIMD
        const double ns_per_step = ms * 1e6 / ((double)nsteps * wps);
        printf("waves/SIMD %d: %.1f ns per wave-step per SIMD -> %.2f ms for 16.3 k wave-steps per SIMD (B=32 launch)\n", wps,
               ns_per_step, ns_per_step * 16.32e3 / 1e6);
    }
    return 0;
}
void vppx_set_error(const char *, ...) {}
