// valu_bench.hip -- issue cost of the VALU / DS instructions the SGM kernels are built from (gfx950).
// Every kernel runs ITER x 64 instances of one instruction (8 independent chains) per wave with the chip
// full (8 waves per SIMD), and prints ns and shader cycles (s_memtime) per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_bench.hip -o /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define ITER 512

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define DEF_KERNEL(NAME, ASM_LINE)                                                                                    \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, unsigned long long *cyc, uint32_t seed)            \
    {                                                                                                                 \
        __shared__ uint32_t lds_[4608];                                                                               \
        lds_[threadIdx.x] = seed;                                                                                     \
        __syncthreads();                                                                                              \
        uint32_t a0 = threadIdx.x * 2654435761u + seed + lds_[(threadIdx.x + 1) & 255], a1 = a0 ^ 0x9e3779b9u, a2 = a0 * 3u, a3 = a0 + 77u;          \
        uint32_t a4 = a1 * 5u, a5 = a2 ^ 0xabcdu, a6 = a3 * 7u, a7 = a4 + 1234567u;                                   \
        uint32_t b = (a0 >> 3) & 0x3FFCu, c = 0x01010101u * (threadIdx.x & 15);                                                   \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                   \
        for (int it = 0; it < ITER; it++) {                                                                           \
            asm volatile(ASM_LINE ASM_LINE ASM_LINE ASM_LINE ASM_LINE ASM_LINE ASM_LINE ASM_LINE                      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c)                                                                             \
                         : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                                                                                    \
        }                                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                  \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                                \
    }

// one line = 8 independent instructions, one per chain; %8 = b, %9 = c
#define L3(OP) \
    OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
    OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define L3S(OP, SUF) \
    OP " %0, %0, %8 " SUF "\n" OP " %1, %1, %8 " SUF "\n" OP " %2, %2, %8 " SUF "\n" OP " %3, %3, %8 " SUF "\n" \
    OP " %4, %4, %8 " SUF "\n" OP " %5, %5, %8 " SUF "\n" OP " %6, %6, %8 " SUF "\n" OP " %7, %7, %8 " SUF "\n"
#define L4(OP) \
    OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" \
    OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
#define L4I(OP, IMM) \
    OP " %0, %0, " IMM ", %9\n" OP " %1, %1, " IMM ", %9\n" OP " %2, %2, " IMM ", %9\n" OP " %3, %3, " IMM ", %9\n" \
    OP " %4, %4, " IMM ", %9\n" OP " %5, %5, " IMM ", %9\n" OP " %6, %6, " IMM ", %9\n" OP " %7, %7, " IMM ", %9\n"
#define L2S(OP, SUF) \
    OP " %0, %8 " SUF "\n" OP " %1, %8 " SUF "\n" OP " %2, %8 " SUF "\n" OP " %3, %8 " SUF "\n" \
    OP " %4, %8 " SUF "\n" OP " %5, %8 " SUF "\n" OP " %6, %8 " SUF "\n" OP " %7, %8 " SUF "\n"
// dpp on the chain itself (src0 = own value of a neighbour lane)
#define L3DPP(OP, SUF) \
    OP " %0, %0, %8 " SUF "\n" OP " %1, %1, %8 " SUF "\n" OP " %2, %2, %8 " SUF "\n" OP " %3, %3, %8 " SUF "\n" \
    OP " %4, %4, %8 " SUF "\n" OP " %5, %5, %8 " SUF "\n" OP " %6, %6, %8 " SUF "\n" OP " %7, %7, %8 " SUF "\n"
#define LCMP(OP) \
    OP " vcc, %0, %8\n" OP " vcc, %1, %8\n" OP " vcc, %2, %8\n" OP " vcc, %3, %8\n" \
    OP " vcc, %4, %8\n" OP " vcc, %5, %8\n" OP " vcc, %6, %8\n" OP " vcc, %7, %8\n"
#define LCND \
    "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n" \
    "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
#define LDS1(OP) \
    OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n" \
    OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"

DEF_KERNEL(xor, L3("v_xor_b32"))
DEF_KERNEL(add_u32, L3("v_add_u32"))
DEF_KERNEL(min_u32, L3("v_min_u32"))
DEF_KERNEL(bcnt, L3("v_bcnt_u32_b32"))
DEF_KERNEL(lshl_add, L4I("v_lshl_add_u32", "16"))
DEF_KERNEL(and_or, L4("v_and_or_b32"))
DEF_KERNEL(alignbit, L4I("v_alignbit_b32", "%8"))
DEF_KERNEL(perm, L4("v_perm_b32"))
DEF_KERNEL(sad_u8, L4("v_sad_u8"))
DEF_KERNEL(min3_u32, L4("v_min3_u32"))
DEF_KERNEL(mad_u32_u24, L4("v_mad_u32_u24"))
DEF_KERNEL(bfe_u32, L4I("v_bfe_u32", "3"))
DEF_KERNEL(pk_min_u16, L3("v_pk_min_u16"))
DEF_KERNEL(pk_max_u16, L3("v_pk_max_u16"))
DEF_KERNEL(pk_add_u16, L3("v_pk_add_u16"))
DEF_KERNEL(pk_add_u16_clamp, L3S("v_pk_add_u16", "clamp"))
DEF_KERNEL(pk_sub_u16, L3("v_pk_sub_u16"))
DEF_KERNEL(pk_mad_u16, L4("v_pk_mad_u16"))
DEF_KERNEL(pk_lshl_b16, L3("v_pk_lshlrev_b16"))
DEF_KERNEL(min_u16, L3("v_min_u16"))
DEF_KERNEL(min_u16_sdwa, L3S("v_min_u16_sdwa", "dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"))
DEF_KERNEL(add_u32_sdwa, L3S("v_add_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"))
DEF_KERNEL(mov_dpp_row_shr1, L2S("v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_KERNEL(min_dpp_quad, L3DPP("v_min_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
DEF_KERNEL(min_dpp_row_mirror, L3DPP("v_min_u32_dpp", "row_mirror row_mask:0xf bank_mask:0xf"))
DEF_KERNEL(cmp_lt_u32, LCMP("v_cmp_lt_u32"))
DEF_KERNEL(cndmask, LCND)
DEF_KERNEL(fma_f32, L4("v_fma_f32"))
DEF_KERNEL(bpermute, LDS1("ds_bpermute_b32"))
DEF_KERNEL(lshlrev, L3("v_lshlrev_b32"))
DEF_KERNEL(max3_u32, L4("v_max3_u32"))
DEF_KERNEL(med3_u32, L4("v_med3_u32"))
DEF_KERNEL(sub_u16, L3("v_sub_u16"))
DEF_KERNEL(msad_u8, L4("v_msad_u8"))
DEF_KERNEL(mul_u32_u24, L3("v_mul_u32_u24"))
DEF_KERNEL(dot4_u32_u8, L4("v_dot4_u32_u8"))

DEF_KERNEL(and_b32, L3("v_and_b32"))
DEF_KERNEL(or_b32, L3("v_or_b32"))
DEF_KERNEL(sub_u32, L3("v_sub_u32"))
DEF_KERNEL(subrev_u32, L3("v_subrev_u32"))
DEF_KERNEL(max_u32, L3("v_max_u32"))
DEF_KERNEL(min_i32, L3("v_min_i32"))
DEF_KERNEL(lshrrev, L3("v_lshrrev_b32"))
DEF_KERNEL(mov_b32, L2S("v_mov_b32", ""))
DEF_KERNEL(add_u16, L3("v_add_u16"))
DEF_KERNEL(max_u16, L3("v_max_u16"))
DEF_KERNEL(min_i16, L3("v_min_i16"))
DEF_KERNEL(mul_lo_u16, L3("v_mul_lo_u16"))
DEF_KERNEL(lshlrev_b16, L3("v_lshlrev_b16"))
DEF_KERNEL(mad_u16, L4("v_mad_u16"))
DEF_KERNEL(add3_u32, L4("v_add3_u32"))
DEF_KERNEL(or3_b32, L4("v_or3_b32"))
DEF_KERNEL(xad_u32, L4("v_xad_u32"))
DEF_KERNEL(add_lshl_u32, L4I("v_add_lshl_u32", "%8"))
DEF_KERNEL(bfi_b32, L4("v_bfi_b32"))
DEF_KERNEL(min_f32, L3("v_min_f32"))
DEF_KERNEL(max_f32, L3("v_max_f32"))
DEF_KERNEL(add_f32, L3("v_add_f32"))
DEF_KERNEL(mul_f32, L3("v_mul_f32"))
DEF_KERNEL(min3_f32, L4("v_min3_f32"))
DEF_KERNEL(min_f16, L3("v_min_f16"))
DEF_KERNEL(add_f16, L3("v_add_f16"))
DEF_KERNEL(pk_min_f16, L3("v_pk_min_f16"))
DEF_KERNEL(pk_max_f16, L3("v_pk_max_f16"))
DEF_KERNEL(pk_add_f16, L3("v_pk_add_f16"))
DEF_KERNEL(pk_mul_f16, L3("v_pk_mul_f16"))
DEF_KERNEL(pk_fma_f16, L4("v_pk_fma_f16"))
DEF_KERNEL(pk_min_i16, L3("v_pk_min_i16"))
DEF_KERNEL(pk_add_i16, L3("v_pk_add_i16"))
DEF_KERNEL(min3_f16, L4("v_min3_f16"))
DEF_KERNEL(min3_u16, L4("v_min3_u16"))
DEF_KERNEL(xor_e64, L3("v_xor_b32_e64"))
DEF_KERNEL(add_u32_e64, L3("v_add_u32_e64"))
DEF_KERNEL(min_u16_e64, L3("v_min_u16_e64"))
DEF_KERNEL(xor_dpp, L3DPP("v_xor_b32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
DEF_KERNEL(min_u16_dpp, L3DPP("v_min_u16_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
DEF_KERNEL(add_u32_dpp_shr, L3DPP("v_add_u32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_KERNEL(cndmask_sgpr, "s_mov_b64 s[20:21], 0x5555\n"
    "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
    "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
DEF_KERNEL(cmp_cnd, "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
    "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
DEF_KERNEL(readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
    "v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3\n")
DEF_KERNEL(ds_write_b16, "ds_write_b16 %8, %0\n ds_write_b16 %8, %1 offset:2\n ds_write_b16 %8, %2 offset:4\n ds_write_b16 %8, %3 offset:6\n"
    "ds_write_b16 %8, %4 offset:8\n ds_write_b16 %8, %5 offset:10\n ds_write_b16 %8, %6 offset:12\n ds_write_b16 %8, %7 offset:14\n s_waitcnt lgkmcnt(0)\n")
DEF_KERNEL(ds_write_b32, "ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:256\n ds_write_b32 %8, %2 offset:512\n ds_write_b32 %8, %3 offset:768\n"
    "ds_write_b32 %8, %4 offset:1024\n ds_write_b32 %8, %5 offset:1280\n ds_write_b32 %8, %6 offset:1536\n ds_write_b32 %8, %7 offset:1792\n s_waitcnt lgkmcnt(0)\n")
DEF_KERNEL(ds_read_u16, "ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:2\n ds_read_u16 %2, %8 offset:4\n ds_read_u16 %3, %8 offset:6\n"
    "ds_read_u16 %4, %8 offset:8\n ds_read_u16 %5, %8 offset:10\n ds_read_u16 %6, %8 offset:12\n ds_read_u16 %7, %8 offset:14\n s_waitcnt lgkmcnt(0)\n")
DEF_KERNEL(ds_read_b32, "ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
    "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n")
// gfx950 additions
DEF_KERNEL(pk_minimum3_f16, L4("v_pk_minimum3_f16"))
DEF_KERNEL(pk_maximum3_f16, L4("v_pk_maximum3_f16"))
DEF_KERNEL(minimum3_f32, L4("v_minimum3_f32"))
DEF_KERNEL(bitop3_b32, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n"
    "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\n")
DEF_KERNEL(pk_min_u16_opsel, L3S("v_pk_min_u16", "op_sel:[1,0] op_sel_hi:[0,1]"))
// mixed: the min-plus inner sequence of one pair (alignbit, pk_min, pk_add, pk_min, pk_min, pk_sub) on 8 chains
#define MIXLINE(R) \
    "v_alignbit_b32 " R ", " R ", %8, 16\n v_pk_min_u16 " R ", " R ", %9\n v_pk_add_u16 " R ", " R ", %8 clamp\n" \
    "v_pk_min_u16 " R ", " R ", %9\n v_xor_b32 " R ", " R ", %8\n v_bcnt_u32_b32 " R ", " R ", %9\n v_pk_sub_u16 " R ", " R ", %8\n v_lshl_add_u32 " R ", " R ", 16, %9\n"
DEF_KERNEL(mix_dep, MIXLINE("%0") )
DEF_KERNEL(mix_ilv,
    "v_alignbit_b32 %0, %0, %8, 16\n v_alignbit_b32 %1, %1, %8, 16\n v_pk_min_u16 %2, %2, %9\n v_pk_min_u16 %3, %3, %9\n"
    "v_pk_add_u16 %4, %4, %8 clamp\n v_pk_add_u16 %5, %5, %8 clamp\n v_xor_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %9\n")

struct Entry { const char *name; void (*fn)(uint32_t *, unsigned long long *, uint32_t); };
#define E(N) {#N, k_##N}

int main(int argc, char **argv)
{
    Entry list[] = {E(xor), E(add_u32), E(min_u32), E(bcnt), E(lshl_add), E(and_or), E(alignbit), E(perm), E(sad_u8), E(min3_u32),
                    E(max3_u32), E(med3_u32), E(mad_u32_u24), E(mul_u32_u24), E(bfe_u32), E(lshlrev), E(pk_min_u16), E(pk_max_u16),
                    E(pk_add_u16), E(pk_add_u16_clamp), E(pk_sub_u16), E(pk_mad_u16), E(pk_lshl_b16), E(min_u16), E(sub_u16),
                    E(min_u16_sdwa), E(add_u32_sdwa), E(mov_dpp_row_shr1), E(min_dpp_quad), E(min_dpp_row_mirror), E(cmp_lt_u32),
                    E(cndmask), E(fma_f32), E(msad_u8), E(dot4_u32_u8), E(bpermute), E(mix_dep), E(mix_ilv),
                    E(and_b32), E(or_b32), E(sub_u32), E(subrev_u32), E(max_u32), E(min_i32), E(lshrrev), E(mov_b32), E(add_u16),
                    E(max_u16), E(min_i16), E(mul_lo_u16), E(lshlrev_b16), E(mad_u16), E(add3_u32), E(or3_b32), E(xad_u32),
                    E(add_lshl_u32), E(bfi_b32), E(min_f32), E(max_f32), E(add_f32), E(mul_f32), E(min3_f32), E(min_f16), E(add_f16),
                    E(pk_min_f16), E(pk_max_f16), E(pk_add_f16), E(pk_mul_f16), E(pk_fma_f16), E(pk_min_i16), E(pk_add_i16),
                    E(min3_f16), E(min3_u16), E(xor_e64), E(add_u32_e64), E(min_u16_e64), E(xor_dpp), E(min_u16_dpp),
                    E(add_u32_dpp_shr), E(cndmask_sgpr), E(cmp_cnd), E(readlane), E(ds_write_b16), E(ds_write_b32), E(ds_read_u16),
                    E(ds_read_b32), E(pk_minimum3_f16), E(pk_maximum3_f16), E(minimum3_f32), E(bitop3_b32), E(pk_min_u16_opsel)};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const int wps = argc > 1 ? atoi(argv[1]) : 8; // waves per SIMD
    const int grid = ncu * wps;                   // 256-thread blocks: 4 waves each, one per SIMD
    uint32_t *out;
    unsigned long long *cyc;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&cyc, (size_t)grid * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("device %s, %d CUs, clock %d kHz, %d waves/SIMD, %d x 64 instr per wave\n", prop.gcnArchName, ncu, prop.clockRate, wps, ITER);
    printf("%-22s %10s %12s %14s\n", "instr", "ms", "ns/instr/SIMD", "memtime/instr");
    std::vector<unsigned long long> h((size_t)grid * 4);
    for (auto &en : list) {
        en.fn<<<grid, 256>>>(out, cyc, 1); // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        en.fn<<<grid, 256>>>(out, cyc, 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= (double)h.size();
        const double n_per_simd = (double)wps * ITER * 64.0;
        // a wave's own span covers the issue slots of all wps waves on its SIMD
        printf("%-22s %10.4f %12.3f %14.3f\n", en.name, ms, ms * 1e6 / n_per_simd, avg / n_per_simd);
    }
    return 0;
}
