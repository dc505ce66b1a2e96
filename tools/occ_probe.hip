// occ_probe.hip -- what the runtime says about the residency of the big kernels: blocks per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor)
// for the W/E kernel's one-wave workgroups, the lock-step kernels and the sum / WTA kernel, next to their register counts.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vppstereo_amd/csrc tools/occ_probe.hip -o tools/bin/occ_probe
#include "../vppstereo_amd/csrc/rsgm_kernels.hip"
#include <stdarg.h>
void vppx_set_error(const char *, ...) {}
int ws_reserve(vppx_ctx *, WsSlot, size_t, void **) { return -1; }

template <typename F>
static void show(const char *name, F fn, int block)
{
    int n = -1;
    hipFuncAttributes fa;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, block, 0);
    hipFuncGetAttributes(&fa, (const void *)fn);
    printf("%-34s block %4d: %2d blocks per CU (%s), %3d VGPRs, %6zu B LDS\n", name, block, n, hipGetErrorString(e), fa.numRegs, fa.sharedSizeBytes);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%s: %d CUs, maxThreadsPerMultiProcessor %d, maxBlocksPerMultiProcessor %d, regsPerMultiprocessor %d\n", prop.gcnArchName,
           prop.multiProcessorCount, prop.maxThreadsPerMultiProcessor, prop.maxBlocksPerMultiProcessor, prop.regsPerMultiprocessor);
    show("sgm_we12_kernel<8, 0>", sgm_we12_kernel<8, 0>, 64);
    show("sgm_we12_kernel<12, 0>", sgm_we12_kernel<12, 0>, 64);
    show("sgm_we12_kernel<12, 1>", sgm_we12_kernel<12, 1>, 64);
    show("sgm_we12_kernel<16, 0>", sgm_we12_kernel<16, 0>, 64);
    show("sgm_vert4_kernel<48>", sgm_vert4_kernel<48>, 256);
    show("sgm_vert3_kernel<24>", sgm_vert3_kernel<24>, 256);
    return 0;
}
