// Issue rate of the fp32 / integer VALU instructions the fused forward and inverse kernels are made of, on one
// MI355X: wave-level instructions per second per SIMD, 8 independent chains per lane, at 1 / 2 / 4 / 8 waves per
// SIMD.  The question behind it: does a hand-packed v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 butterfly halve the VALU time of the DCT, i.e. does a packed instruction issue at the rate of a
// plain one (2 results per slot) or at half of it (no gain)?  "flops-equivalent" below counts a packed
// instruction as two.
//   hipcc --offload-arch=gfx950 -O3 -o microbench/valu_rate microbench/valu_rate.hip && microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters, float seed)
{
    float f[8], f2[8];
    f32x2 p[8];
    unsigned u[8], u2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f[k] = seed + k + threadIdx.x * 1e-3f;
        f2[k] = f[k] * 0.5f;
        p[k] = f32x2{f[k], f[k] + 0.5f};
        u[k] = (unsigned)threadIdx.x * 2654435761u + k;
        u2[k] = u[k] ^ 0x55u;
    }
    const float m = 1.0000001f, c = 1e-9f;
    const float sm = __builtin_amdgcn_readfirstlane(__float_as_int(seed)) ? 1.0000001f : 1.0f;   // lives in an SGPR
    const f32x2 pm = {m, m}, pc = {c, c};
    double dd[4] = {1.0 + seed, 2.0, 3.0, 4.0};
    const double dm = 1.0000001, dc = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pc));
                if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pm));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c));
                if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(m));
                if (OP == 6) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[k]));
                if (OP == 7) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[k]) : "v"(f[k]));
                if (OP == 8) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[k]) : "v"(u[k]));
                if (OP == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(0x05040100u));
                if (OP == 10) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(f[k]) : "v"(m), "v"(c));
                if (OP == 11) asm volatile("v_fma_f32 %0, %1, |%2|, |%0|" : "+v"(f[k]) : "v"(m), "v"(c));
                if (OP == 12) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c), "v"(m));
                if (OP == 13) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(0xFFFFu), "v"(u[(k + 1) & 7]));
                if (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 15) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 16) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[k]) : "v"(f[k]));
                if (OP == 17) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c));
                if (OP == 18) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p[k]) : "v"(pc));
                if (OP == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(p[k]) : "v"(pm), "v"(pc));
                if (OP == 20) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 22) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "s"(sm), "v"(c));            // one SGPR operand
                if (OP == 23) asm volatile("v_fmamk_f32 %0, %0, 0x3f7b14be, %1" : "+v"(f[k]) : "v"(c));          // VOP2 + 32-bit literal
                if (OP == 24) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[k]) : "s"(sm));                        // VOP2 with an SGPR
                if (OP == 25) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[k]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]), "v"(f[(k + 3) & 7]));
                if (OP == 26) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[0]) : "v"(m), "v"(c));            // ONE dependent chain
                if (OP == 27) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k & 1]) : "v"(m), "v"(c));        // two chains
                if (OP == 28) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k & 3]) : "v"(m), "v"(c));        // four chains
                if (OP == 29) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[k]) : "v"(u[k]));
                if (OP == 30) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[k]) : "v"(u[k]));
                if (OP == 31) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[k]) : "v"(f[k]));
                if (OP == 32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c));
                if (OP == 33) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(f[k]) : "v"(c));
                if (OP == 34) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 35) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 36) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[k]));
                if (OP == 37) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 38) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 39) asm volatile("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 40) asm volatile("v_fract_f32 %0, %0" : "+v"(f[k]));
                if (OP == 41) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[k]) : "v"(u[k]));
                if (OP == 42) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 43) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 44) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[k]) : "v"(0xFFFFu), "v"(u[(k + 1) & 7]));
                if (OP == 45) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[k]) : "v"(0x10101u));
                if (OP == 46) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 47) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 7]) : "vcc");
                if (OP == 48) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[k]), "v"(c) : "vcc");
                if (OP == 49) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[k]) : "v"(m), "v"(c));
                if (OP == 50) asm volatile("v_cvt_f32_ubyte0_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f[k]) : "v"(u[k]));
                if (OP == 51) asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(f[k]) : "v"(c));
                if (OP == 52) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[k & 3]) : "v"(dm), "v"(dc));
                // integer / bit opcodes of the entropy stage's kernels (round 3, second batch)
                if (OP == 53) asm volatile("v_ffbh_u32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 54) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7]));
                if (OP == 55) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 56) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7]));
                if (OP == 57) asm volatile("v_pk_sub_i16 %0, 0, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 58) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 59) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u[k]) : "s"(0x00010001u));
                if (OP == 60) asm volatile("v_lshlrev_b64 %0, 5, %0" : "+v"(dd[k & 3]));
                if (OP == 61) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 62) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[k]), "v"(u[(k + 1) & 7]) : "vcc");
                if (OP == 63) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 64) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 65) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 66) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 67) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 68) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7]));
                if (OP == 69) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "s"(0x5555555555555555ull));
                if (OP == 70) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 71) asm volatile("v_ashrrev_i32 %0, 16, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 72) asm volatile("v_ffbh_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 73) asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 74) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[k])); }   // one fast + one slow per count: do the classes overlap?
                if (OP == 75) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u2[k]) : "v"(u[(k + 1) & 7])); asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(u[k])); }   // two fast + one slow
                if (OP == 76) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u2[k]) : "v"(u[(k + 1) & 7]), "v"(0x05040100u)); asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(u[k])); }   // one fast + two slow
                if (OP == 77) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
                if (OP == 78) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[k]));
                if (OP == 79) { asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(u[k])); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u2[k]) : "v"(u[(k + 1) & 7]), "v"(0x05040100u)); }   // two slow of different kinds
                if (OP == 80) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u2[k]) : "v"(u[(k + 1) & 7])); }   // two fast
                // packed fp32 (two results, slow class) next to plain fp32 (fast class): does the pair overlap like the integer pairs above?
                if (OP == 81) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); }
                if (OP == 82) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(m), "v"(c)); }
                if (OP == 83) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pc)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(m), "v"(c)); }
                if (OP == 84) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pm)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); }
                if (OP == 85) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f2[k]) : "v"(c)); }
                if (OP == 86) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pc)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); }
                if (OP == 87) { asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,0]" : "+v"(p[k]) : "v"(pc)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[k]) : "v"(m), "v"(c)); }
                if (OP == 88) asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(f[k]) : "v"(c));
                if (OP == 89) { asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(f[k]) : "v"(c)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f2[k]) : "v"(c)); }
                if (OP == 90) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f2[k]) : "v"(c)); }
                if (OP == 91) { asm volatile("v_rndne_f32 %0, %0" : "+v"(f[k])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2[k]) : "v"(m), "v"(c)); }
                if (OP == 92) { asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(u[k]) : "v"(f[k])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2[k]) : "v"(m), "v"(c)); }
                if (OP == 93) { asm volatile("v_add_f32_e64 %0, %0, %1 mul:2" : "+v"(f[k]) : "v"(c)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f2[k]) : "v"(c)); }
                if (OP == 94) asm volatile("v_add_f32_e64 %0, %0, %1 mul:2" : "+v"(f[k]) : "v"(c));
                if (OP == 95) { asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(f[k]) : "v"(c)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f2[k]) : "v"(c)); }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k] + f2[k] + p[k].x + p[k].y + (float)u[k] + (float)u2[k] + (float)dd[k & 3];
    if (s == 12345.678f) out[0] = s;
}

static const char *NAMES[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32", "v_mul_f32", "v_rndne_f32",
                              "v_cvt_i32_f32", "v_cvt_f32_ubyte1", "v_perm_b32", "v_max3_f32 |a| |b|", "v_fma_f32 a |b| |c|", "v_med3_f32",
                              "v_and_or_b32", "v_lshl_or_b32", "v_cvt_pk_i16_i32", "v_cvt_pk_u8_f32", "v_sub_f32", "v_pk_add_f32 neg_lo",
                              "v_pk_fma_f32 op_sel", "v_bfe_u32", "v_mov_b32", "v_fma_f32 (sgpr src)", "v_fmamk_f32 (literal)", "v_mul_f32 (sgpr src)",
                              "v_fma_f32 3 vgpr srcs", "v_fma_f32 1 chain", "v_fma_f32 2 chains", "v_fma_f32 4 chains", "v_cvt_f32_i32",
                              "v_cvt_f32_i32 sdwa", "v_cvt_u32_f32", "v_max_f32", "v_max_f32 |a| |b|", "v_and_b32", "v_or_b32", "v_lshlrev_b32",
                              "v_add_u32", "v_add_u32 sdwa bytes", "v_min_u32 sdwa byte", "v_fract_f32", "v_cvt_f32_u32", "v_alignbit_b32",
                              "v_pack_b32_f16", "v_bfi_b32", "v_mul_u32_u24", "v_pk_add_u16", "v_cndmask_b32", "v_cmp_lt_f32", "v_fmac_f32",
                              "v_cvt_f32_ubyte0 sdwa", "v_add_f32 |a|", "v_fma_f64 4 chains",
                              "v_ffbh_u32", "v_add3_u32", "v_lshl_add_u32", "v_or3_b32", "v_pk_sub_i16", "v_pk_max_i16", "v_pk_min_u16 (sgpr)", "v_lshlrev_b64",
                              "v_lshrrev_b32", "v_cmp_gt_u32", "v_min_u32", "v_sub_u32", "v_xor_b32", "v_bcnt_u32_b32", "v_mul_lo_u32", "v_mad_u32_u24",
                              "v_cndmask_b32 (sgpr pair)", "v_lshlrev_b32 (vgpr amount)", "v_ashrrev_i32", "v_ffbh_u32 sdwa", "v_and_b32 (literal)",
                              "v_mul_f32 + v_lshlrev_b32 (2 instr)", "2 fast + 1 slow (3 instr)", "1 fast + 2 slow (3 instr)", "v_lshlrev_b32 dst != src", "v_lshrrev_b32 in place", "2 slow (2 instr)", "2 fast (2 instr)", "v_pk_add_f32 + v_add_f32 (2 instr)", "v_pk_add_f32 + v_fma_f32 (2 instr)", "v_pk_fma_f32 + v_fma_f32 (2 instr)", "v_pk_mul_f32 + v_sub_f32 (2 instr)", "v_pk_add_f32 + v_add_f32 + v_sub_f32 (3 instr)", "v_pk_fma_f32 + v_add_f32 (2 instr)", "v_pk_add_f32 neg_lo + v_fmac_f32 (2 instr)", "v_add_f32 sdwa (dword selects)", "v_add_f32 sdwa + v_sub_f32 (2 instr)", "v_max_f32 + v_sub_f32 (2 instr)", "v_rndne_f32 + v_fma_f32 (2 instr)", "v_cvt_pk_u8_f32 + v_fma_f32 (2 instr)", "v_add_f32 e64 mul:2 + v_sub_f32 (2 instr)", "v_add_f32 e64 mul:2", "v_add_f32 dpp + v_sub_f32 (2 instr)"};

static double g_base = 0;

template <int OP>
void run(float *d, int waves_per_simd = 8)
{
    const int iters = 4096, grid = 256 * waves_per_simd;            // workgroups of 4 waves, one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<OP>), dim3(grid), dim3(256), 0, 0, d, 64, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<OP>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double wave_instr = (double)grid * 4 * iters * 32;           // wave-level instructions issued
    const double per_simd_per_s = wave_instr / (best * 1e-3) / (256 * 4);
    if (OP == 0 && waves_per_simd == 8) g_base = per_simd_per_s;
    printf("%-22s %d waves/SIMD %8.3f ms  %6.3f G wave-instr/s per SIMD  x%.2f of v_fma_f32 @8\n", NAMES[OP], waves_per_simd, best,
           per_simd_per_s * 1e-9, g_base > 0 ? per_simd_per_s / g_base : 0.0);
}

template <int OP> void sweep(float *d) { run<OP>(d, 8); }

int main()
{
    float *d;
    hipMalloc(&d, 64);
    run<0>(d, 8);
    sweep<1>(d); sweep<2>(d); sweep<3>(d); sweep<18>(d); sweep<19>(d);
    sweep<4>(d); sweep<5>(d); sweep<17>(d); sweep<6>(d); sweep<7>(d); sweep<8>(d); sweep<9>(d); sweep<10>(d); sweep<11>(d); sweep<12>(d);
    sweep<13>(d); sweep<14>(d); sweep<15>(d); sweep<16>(d); sweep<20>(d); sweep<21>(d);
    sweep<22>(d); sweep<23>(d); sweep<24>(d); sweep<25>(d); sweep<26>(d); sweep<27>(d); sweep<28>(d); sweep<29>(d); sweep<30>(d); sweep<31>(d);
    sweep<32>(d); sweep<33>(d); sweep<34>(d); sweep<35>(d); sweep<36>(d); sweep<37>(d); sweep<38>(d); sweep<39>(d); sweep<40>(d); sweep<41>(d);
    sweep<42>(d); sweep<43>(d); sweep<44>(d); sweep<45>(d); sweep<46>(d); sweep<47>(d); sweep<48>(d); sweep<49>(d); sweep<50>(d); sweep<51>(d); sweep<52>(d);
    if (getenv("VALU_RATE_BATCH2")) {
        sweep<53>(d); sweep<54>(d); sweep<55>(d); sweep<56>(d); sweep<57>(d); sweep<58>(d); sweep<59>(d); sweep<60>(d); sweep<61>(d); sweep<62>(d); sweep<63>(d);
        sweep<64>(d); sweep<65>(d); sweep<66>(d); sweep<67>(d); sweep<68>(d); sweep<69>(d); sweep<70>(d); sweep<71>(d); sweep<72>(d); sweep<73>(d); sweep<74>(d); sweep<75>(d); sweep<76>(d); sweep<77>(d); sweep<78>(d); sweep<79>(d); sweep<80>(d); sweep<81>(d); sweep<82>(d); sweep<83>(d); sweep<84>(d); sweep<85>(d); sweep<86>(d); sweep<87>(d); sweep<88>(d); sweep<89>(d); sweep<90>(d); sweep<91>(d); sweep<92>(d); sweep<93>(d); sweep<94>(d); sweep<95>(d);
        return 0;
    }
    for (int w : {1, 2, 4}) run<26>(d, w);
    for (int w : {1, 2, 4}) run<28>(d, w);
    for (int w : {1, 2, 3, 4, 6}) run<0>(d, w);
    for (int w : {1, 2, 4}) run<1>(d, w);
    return 0;
}
