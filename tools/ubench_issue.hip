// Micro-benchmark: issue cost of single VALU / LDS instructions on gfx950, measured in shader cycles (s_memtime) per
// wave64 instruction with W wavefronts resident per SIMD, and as whole-chip wall time.  Every body is inline asm with 8
// independent accumulators (an instruction depends on the one issued 8 earlier), 32 instructions per loop trip.
//   per-wave ticks / instruction / W  =  cycles one SIMD spends per instruction when W waves compete for it.
// Also: pairs of DIFFERENT instruction kinds run by alternate wavefronts of one SIMD ("A|B"), which shows whether two
// kinds share an issue port (time adds) or not (time of the slower one).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define TRIPS 4000

#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define R32(op) R8(op) R8(op) R8(op) R8(op)

// accumulators: d0-d7 (64-bit, "+v"), i0-i7 (32-bit, "+v"); constants: c64 a/b ("v"), c32 a/b ("v")
#define OPERANDS \
    : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), /* %0-%7 */ \
      "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]), "+v"(i[4]), "+v"(i[5]), "+v"(i[6]), "+v"(i[7])  /* %8-%15 */ \
    : "v"(ca), "v"(cb), "v"(ia), "v"(ib), "s"(sa), "s"(sb)                                           /* %16-%21 */ \
    : "vcc"

enum {
    FMA_F64, MUL_F64, ADD_F64, RCP_F64, CVT_F32_F64, CVT_F64_U32, FREXP_F64, MIN_F64, FMA_F64_S,
    FMA_F32, MUL_F32, ADD_F32, PK_FMA_F32, PK_MUL_F32, RNDNE_F32, CVT_I32_F32, CVT_F32_UBYTE, MAX_F32, MED3_F32,
    AND_B32, LSHL_B32, ADD_U32, BFE_U32, PERM_B32, ALIGNBYTE, MAD_U24, MUL_U24, DOT4_U8, DOT2_U16, LSHL_OR, AND_OR,
    CNDMASK, CMP_F32, MIN_U32, MAX3_U32, MUL_LO_U32, MAD_U64_U32, PK_MAD_U16, PK_MUL_LO_U16, MOV_B32, ADD3_U32, SAD_U8,
    LSHL_ADD, XAD_U32, MAD_I32_I24, MBCNT, OR_B32, XOR_B32, SUB_U32, CNDMASK_S, CMP_CND, FMAC_F32, SUB_F32, CVT_F32_U32, CVT_U32_F32, LSHR_B32, BFI_B32, MOV_DPP, ADD_CO, ADDC, MUL_HI_U32, MAD_U32_U16,
    DS_READ_B32, DS_READ_B64, DS_READ_B128, NOPS
};

template <int OP>
__device__ __forceinline__ void body(double (&d)[8], uint32_t (&i)[8], double ca, double cb, uint32_t ia, uint32_t ib,
                                     uint32_t sa, uint32_t sb, uint32_t lds)
{
    (void)lds;
#define A(k) "v_fma_f64 %" #k ", %" #k ", %16, %17\n"
    if (OP == FMA_F64) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_fma_f64 %" #k ", %" #k ", %16, %17\n"
    if (OP == FMA_F64_S) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_mul_f64 %" #k ", %" #k ", %16\n"
    if (OP == MUL_F64) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_add_f64 %" #k ", %" #k ", %16\n"
    if (OP == ADD_F64) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_rcp_f64 %" #k ", %" #k "\n"
    if (OP == RCP_F64) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_cvt_f32_f64 %" #k "+8, %" #k "\n"
#define B(k, j) "v_cvt_f32_f64 %" #j ", %" #k "\n"
    if (OP == CVT_F32_F64) asm volatile(B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15)
                                        B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) OPERANDS);
#undef A
#undef B
#define B(k, j) "v_cvt_f64_u32 %" #k ", %" #j "\n"
    if (OP == CVT_F64_U32) asm volatile(B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15)
                                        B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) OPERANDS);
#undef B
#define B(k, j) "v_frexp_exp_i32_f64 %" #j ", %" #k "\n"
    if (OP == FREXP_F64) asm volatile(B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15)
                                      B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) B(0, 8) B(1, 9) B(2, 10) B(3, 11) B(4, 12) B(5, 13) B(6, 14) B(7, 15) OPERANDS);
#undef B
#define A(k) "v_min_f64 %" #k ", %" #k ", %16\n"
    if (OP == MIN_F64) asm volatile(R32(A) OPERANDS);
#undef A
    // ---- 32-bit: accumulators %8-%15 ----
#define I8(op) op(8) op(9) op(10) op(11) op(12) op(13) op(14) op(15)
#define I32(op) I8(op) I8(op) I8(op) I8(op)
#define A(k) "v_fma_f32 %" #k ", %" #k ", %18, %19\n"
    if (OP == FMA_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mul_f32 %" #k ", %" #k ", %18\n"
    if (OP == MUL_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_add_f32 %" #k ", %" #k ", %18\n"
    if (OP == ADD_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_pk_fma_f32 %" #k ", %" #k ", %16, %17\n"
    if (OP == PK_FMA_F32) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_pk_mul_f32 %" #k ", %" #k ", %16\n"
    if (OP == PK_MUL_F32) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_rndne_f32 %" #k ", %" #k "\n"
    if (OP == RNDNE_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
    if (OP == CVT_I32_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
    if (OP == CVT_F32_UBYTE) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_max_f32 %" #k ", %" #k ", %18\n"
    if (OP == MAX_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_med3_f32 %" #k ", %" #k ", %18, %19\n"
    if (OP == MED3_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_and_b32 %" #k ", %" #k ", %18\n"
    if (OP == AND_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
    if (OP == LSHL_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_add_u32 %" #k ", %" #k ", %18\n"
    if (OP == ADD_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_bfe_u32 %" #k ", %" #k ", 3, 17\n"
    if (OP == BFE_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_perm_b32 %" #k ", %" #k ", %18, %19\n"
    if (OP == PERM_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_alignbyte_b32 %" #k ", %" #k ", %18, %19\n"
    if (OP == ALIGNBYTE) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mad_u32_u24 %" #k ", %" #k ", %18, %19\n"
    if (OP == MAD_U24) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mul_u32_u24 %" #k ", %" #k ", %18\n"
    if (OP == MUL_U24) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_dot4_u32_u8 %" #k ", %" #k ", %18, %19\n"
    if (OP == DOT4_U8) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_dot2_u32_u16 %" #k ", %" #k ", %18, %19\n"
    if (OP == DOT2_U16) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_lshl_or_b32 %" #k ", %" #k ", 3, %18\n"
    if (OP == LSHL_OR) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_and_or_b32 %" #k ", %" #k ", %18, %19\n"
    if (OP == AND_OR) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_cndmask_b32 %" #k ", %" #k ", %18, vcc\n"
    if (OP == CNDMASK) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_cmp_gt_f32 vcc, %" #k ", %18\n"
    if (OP == CMP_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_min_u32 %" #k ", %" #k ", %18\n"
    if (OP == MIN_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_max3_u32 %" #k ", %" #k ", %18, %19\n"
    if (OP == MAX3_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mul_lo_u32 %" #k ", %" #k ", %18\n"
    if (OP == MUL_LO_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mad_u64_u32 %" #k ", vcc, %18, %19, %" #k "\n"
    if (OP == MAD_U64_U32) asm volatile(R32(A) OPERANDS);
#undef A
#define A(k) "v_pk_mad_u16 %" #k ", %" #k ", %18, %19\n"
    if (OP == PK_MAD_U16) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_pk_mul_lo_u16 %" #k ", %" #k ", %18\n"
    if (OP == PK_MUL_LO_U16) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mov_b32 %" #k ", %18\n"
    if (OP == MOV_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_add3_u32 %" #k ", %" #k ", %18, %19\n"
    if (OP == ADD3_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_sad_u8 %" #k ", %" #k ", %18, %19\n"
    if (OP == SAD_U8) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %18\n"
    if (OP == LSHL_ADD) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_xad_u32 %" #k ", %" #k ", %18, %19\n"
    if (OP == XAD_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mad_i32_i24 %" #k ", %" #k ", %18, %19\n"
    if (OP == MAD_I32_I24) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mbcnt_lo_u32_b32 %" #k ", %18, %" #k "\n"
    if (OP == MBCNT) asm volatile(I32(A) OPERANDS);
#undef A

#define A(k) "v_or_b32 %" #k ", %" #k ", %18\n"
    if (OP == OR_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_xor_b32 %" #k ", %" #k ", %18\n"
    if (OP == XOR_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_sub_u32 %" #k ", %" #k ", %18\n"
    if (OP == SUB_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_cndmask_b32 %" #k ", %" #k ", %18, s[20:21]\n"
    if (OP == CNDMASK_S) asm volatile(I32(A) OPERANDS, "s20", "s21");
#undef A
#define A(k) "v_cmp_gt_u32 vcc, %" #k ", %18\n v_cndmask_b32 %" #k ", %" #k ", %19, vcc\n"
    if (OP == CMP_CND) asm volatile(I8(A) I8(A) OPERANDS);
#undef A
#define A(k) "v_fmac_f32 %" #k ", %18, %19\n"
    if (OP == FMAC_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_sub_f32 %" #k ", %" #k ", %18\n"
    if (OP == SUB_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
    if (OP == CVT_F32_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_cvt_u32_f32 %" #k ", %" #k "\n"
    if (OP == CVT_U32_F32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_lshrrev_b32 %" #k ", 1, %" #k "\n"
    if (OP == LSHR_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_bfi_b32 %" #k ", %" #k ", %18, %19\n"
    if (OP == BFI_B32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n"
    if (OP == MOV_DPP) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %18\n"
    if (OP == ADD_CO) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %18, vcc\n"
    if (OP == ADDC) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mul_hi_u32 %" #k ", %" #k ", %18\n"
    if (OP == MUL_HI_U32) asm volatile(I32(A) OPERANDS);
#undef A
#define A(k) "v_mad_u32_u16 %" #k ", %" #k ", %18, %19\n"
    if (OP == MAD_U32_U16) asm volatile(I32(A) OPERANDS);
#undef A
    // ---- LDS reads: address in a VGPR, results into the accumulators, one wait per 8 ----
    if (OP == DS_READ_B32) {
        asm volatile("ds_read_b32 %8, %22\n ds_read_b32 %9, %22 offset:256\n ds_read_b32 %10, %22 offset:512\n ds_read_b32 %11, %22 offset:768\n"
                     "ds_read_b32 %12, %22 offset:1024\n ds_read_b32 %13, %22 offset:1280\n ds_read_b32 %14, %22 offset:1536\n ds_read_b32 %15, %22 offset:1792\n"
                     "ds_read_b32 %8, %22\n ds_read_b32 %9, %22 offset:256\n ds_read_b32 %10, %22 offset:512\n ds_read_b32 %11, %22 offset:768\n"
                     "ds_read_b32 %12, %22 offset:1024\n ds_read_b32 %13, %22 offset:1280\n ds_read_b32 %14, %22 offset:1536\n ds_read_b32 %15, %22 offset:1792\n"
                     "ds_read_b32 %8, %22\n ds_read_b32 %9, %22 offset:256\n ds_read_b32 %10, %22 offset:512\n ds_read_b32 %11, %22 offset:768\n"
                     "ds_read_b32 %12, %22 offset:1024\n ds_read_b32 %13, %22 offset:1280\n ds_read_b32 %14, %22 offset:1536\n ds_read_b32 %15, %22 offset:1792\n"
                     "ds_read_b32 %8, %22\n ds_read_b32 %9, %22 offset:256\n ds_read_b32 %10, %22 offset:512\n ds_read_b32 %11, %22 offset:768\n"
                     "ds_read_b32 %12, %22 offset:1024\n ds_read_b32 %13, %22 offset:1280\n ds_read_b32 %14, %22 offset:1536\n ds_read_b32 %15, %22 offset:1792\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                       "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]), "+v"(i[4]), "+v"(i[5]), "+v"(i[6]), "+v"(i[7])
                     : "v"(ca), "v"(cb), "v"(ia), "v"(ib), "s"(sa), "s"(sb), "v"(lds) : "vcc", "memory");
    }
    if (OP == DS_READ_B64) {
#define L(k, o) "ds_read_b64 %" #k ", %22 offset:" #o "\n"
        asm volatile(L(0, 0) L(1, 448) L(2, 896) L(3, 1344) L(4, 1792) L(5, 2240) L(6, 2688) L(7, 3136) L(0, 0) L(1, 448) L(2, 896) L(3, 1344) L(4, 1792) L(5, 2240) L(6, 2688) L(7, 3136)
                     L(0, 0) L(1, 448) L(2, 896) L(3, 1344) L(4, 1792) L(5, 2240) L(6, 2688) L(7, 3136) L(0, 0) L(1, 448) L(2, 896) L(3, 1344) L(4, 1792) L(5, 2240) L(6, 2688) L(7, 3136)
                     "s_waitcnt lgkmcnt(0)\n"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                       "+v"(i[0]), "+v"(i[1]), "+v"(i[2]), "+v"(i[3]), "+v"(i[4]), "+v"(i[5]), "+v"(i[6]), "+v"(i[7])
                     : "v"(ca), "v"(cb), "v"(ia), "v"(ib), "s"(sa), "s"(sb), "v"(lds) : "vcc", "memory");
#undef L
    }
}

__device__ __forceinline__ void run_op(int op, double (&d)[8], uint32_t (&i)[8], double ca, double cb, uint32_t ia, uint32_t ib,
                                       uint32_t sa, uint32_t sb, uint32_t lds);

template <int OP>
__global__ __launch_bounds__(64) void k_single(double* out, const double* in, long long* ticks)
{
    constexpr bool OPSEL_B64 = OP == DS_READ_B64;
    __shared__ uint32_t s_lds[1024];
    double d[8]; uint32_t i[8];
    for (int q = 0; q < 8; ++q) { d[q] = in[q] + 1e-9 * threadIdx.x; i[q] = (uint32_t)(in[q] * 1000) + threadIdx.x; }
    const double ca = in[9], cb = in[10];
    const uint32_t ia = (uint32_t)(ca * 77) | 1u, ib = (uint32_t)(cb * 55);
    const uint32_t sa = __builtin_amdgcn_readfirstlane(ia), sb = __builtin_amdgcn_readfirstlane(ib);
    for (int q = threadIdx.x; q < 1024; q += 64) s_lds[q] = q;
    __syncthreads();
    const uint32_t lds = (uint32_t)(uintptr_t)s_lds + (OPSEL_B64 ? 8u : 4u) * threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < TRIPS; ++r) body<OP>(d, i, ca, cb, ia, ib, sa, sb, lds);
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int q = 0; q < 8; ++q) s += d[q] + i[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// two kinds on alternate wavefronts
template <int OPA, int OPB>
__global__ __launch_bounds__(64) void k_pair(double* out, const double* in, long long* ticks)
{
    constexpr bool OPSEL_B64 = false;
    __shared__ uint32_t s_lds[1024];
    double d[8]; uint32_t i[8];
    for (int q = 0; q < 8; ++q) { d[q] = in[q] + 1e-9 * threadIdx.x; i[q] = (uint32_t)(in[q] * 1000) + threadIdx.x; }
    const double ca = in[9], cb = in[10];
    const uint32_t ia = (uint32_t)(ca * 77) | 1u, ib = (uint32_t)(cb * 55);
    const uint32_t sa = __builtin_amdgcn_readfirstlane(ia), sb = __builtin_amdgcn_readfirstlane(ib);
    for (int q = threadIdx.x; q < 1024; q += 64) s_lds[q] = q;
    __syncthreads();
    const uint32_t lds = (uint32_t)(uintptr_t)s_lds + (OPSEL_B64 ? 8u : 4u) * threadIdx.x;
    // blocks go round-robin over XCDs (8), then CUs...: use a coarse split so that both kinds land on every SIMD
    const bool second = ((blockIdx.x >> 10) & 1) != 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (!second) {
#pragma unroll 1
        for (int r = 0; r < TRIPS; ++r) body<OPA>(d, i, ca, cb, ia, ib, sa, sb, lds);
    } else {
#pragma unroll 1
        for (int r = 0; r < TRIPS; ++r) body<OPB>(d, i, ca, cb, ia, ib, sa, sb, lds);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int q = 0; q < 8; ++q) s += d[q] + i[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static double *g_out, *g_in; static long long* g_ticks;

template <typename K> static void time_kernel(K kern, const char* name, int waves_per_simd, int split)
{
    const int blocks = 1024 * waves_per_simd;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, g_out, g_in, g_ticks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, g_out, g_in, g_ticks);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[8192];
    hipMemcpy(h, g_ticks, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double n = TRIPS * 32.0;
    if (!split) {
        double sum = 0, mx = 0; for (int b = 0; b < blocks; ++b) { sum += h[b]; if (h[b] > mx) mx = h[b]; }
        printf("%-34s W=%d  ticks/instr/wave: mean %.2f max %.2f  -> per-SIMD cycles/instr %.2f   wall %.3f ms (%.2f cyc/instr @2.4GHz)\n",
               name, waves_per_simd, sum / blocks / n, mx / n, sum / blocks / n / waves_per_simd, ms,
               ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
    } else {
        double sa = 0, sb = 0; int na = 0, nb = 0;
        for (int b = 0; b < blocks; ++b) { if ((b >> 10) & 1) { sb += h[b]; ++nb; } else { sa += h[b]; ++na; } }
        printf("%-34s W=%d  ticks/instr/wave: A %.2f  B %.2f   wall %.3f ms\n", name, waves_per_simd, sa / na / n, nb ? sb / nb / n : 0.0, ms);
    }
}

#define SINGLE(op, w) time_kernel(k_single<op>, #op, w, 0)
#define PAIR(a, b, w) time_kernel(k_pair<a, b>, #a "|" #b, w, 1)

int main(int argc, char** argv)
{
    hipMalloc(&g_out, 8192 * 64 * 8); hipMalloc(&g_in, 16 * 8); hipMalloc(&g_ticks, 8192 * 8);
    double h[16]; for (int q = 0; q < 16; ++q) h[q] = 1.0 + q * 0.001;
    hipMemcpy(g_in, h, sizeof(h), hipMemcpyHostToDevice);
    const int quick = argc > 1 && !strcmp(argv[1], "quick");
    if (argc > 3 && !strcmp(argv[1], "power")) {
        // power OP SECONDS: one instruction class back to back on the whole chip (8 wavefronts per SIMD) for SECONDS, for a sampler
        // of socket power and shader clock beside it (tools/ubench_power.sh); prints the achieved rate
        const double seconds = atof(argv[3]);
        void (*kern)(double*, const double*, long long*) = nullptr;
#define PICK(op) if (!strcmp(argv[2], #op)) kern = k_single<op>;
        PICK(FMA_F64) PICK(MUL_F64) PICK(ADD_F64) PICK(RCP_F64) PICK(CVT_F32_F64) PICK(FMA_F32) PICK(AND_B32) PICK(ADD_U32) PICK(MAD_U24)
        PICK(DOT2_U16) PICK(PERM_B32) PICK(LSHL_ADD) PICK(MAX3_U32) PICK(MOV_B32) PICK(DS_READ_B32) PICK(PK_MAD_U16) PICK(CVT_I32_F32)
#undef PICK
        if (!kern) { printf("unknown op %s\n", argv[2]); return 1; }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        long launches = 0; float total = 0;
        while (total < seconds * 1e3f) {
            hipEventRecord(e0);
            for (int q = 0; q < 50; ++q) hipLaunchKernelGGL(kern, dim3(1024 * 8), dim3(64), 0, 0, g_out, g_in, g_ticks);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); total += ms; launches += 50;
        }
        const double instr = (double)launches * 1024 * 8 * TRIPS * 32.0;       // wave64 instructions
        printf("%-14s %.2f s  %.3f T wave-instr/s  (%.2f ns per instruction and SIMD)\n", argv[2], total * 1e-3, instr / (total * 1e-3) / 1e12,
               total * 1e6 / (instr / 1024.0));
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "more")) {
        for (int w = 4; w <= 8; w *= 2) {
            SINGLE(AND_B32, w); SINGLE(OR_B32, w); SINGLE(XOR_B32, w); SINGLE(SUB_U32, w); SINGLE(ADD_U32, w); SINGLE(CNDMASK, w); SINGLE(CNDMASK_S, w); SINGLE(CMP_CND, w);
            SINGLE(FMAC_F32, w); SINGLE(SUB_F32, w); SINGLE(CVT_F32_U32, w); SINGLE(CVT_U32_F32, w); SINGLE(LSHR_B32, w); SINGLE(BFI_B32, w); SINGLE(MOV_DPP, w);
            SINGLE(ADD_CO, w); SINGLE(ADDC, w); SINGLE(MUL_HI_U32, w); SINGLE(MAD_U32_U16, w); SINGLE(DS_READ_B32, w); SINGLE(DS_READ_B64, w); SINGLE(FMA_F64, w); SINGLE(DOT4_U8, w);
        }
        return 0;
    }
    for (int w = 1; w <= 8; w *= 2) {
        if (quick && w != 1 && w != 8) continue;
        SINGLE(FMA_F64, w); SINGLE(MUL_F64, w); SINGLE(ADD_F64, w); SINGLE(RCP_F64, w); SINGLE(CVT_F32_F64, w);
        SINGLE(CVT_F64_U32, w); SINGLE(FREXP_F64, w); SINGLE(MIN_F64, w);
        SINGLE(FMA_F32, w); SINGLE(MUL_F32, w); SINGLE(ADD_F32, w); SINGLE(PK_FMA_F32, w); SINGLE(PK_MUL_F32, w);
        SINGLE(RNDNE_F32, w); SINGLE(CVT_I32_F32, w); SINGLE(CVT_F32_UBYTE, w); SINGLE(MAX_F32, w); SINGLE(MED3_F32, w);
        SINGLE(AND_B32, w); SINGLE(LSHL_B32, w); SINGLE(ADD_U32, w); SINGLE(BFE_U32, w); SINGLE(PERM_B32, w);
        SINGLE(ALIGNBYTE, w); SINGLE(MAD_U24, w); SINGLE(MUL_U24, w); SINGLE(DOT4_U8, w); SINGLE(DOT2_U16, w);
        SINGLE(LSHL_OR, w); SINGLE(AND_OR, w); SINGLE(CNDMASK, w); SINGLE(CMP_F32, w); SINGLE(MIN_U32, w);
        SINGLE(MAX3_U32, w); SINGLE(MUL_LO_U32, w); SINGLE(MAD_U64_U32, w); SINGLE(PK_MAD_U16, w); SINGLE(PK_MUL_LO_U16, w);
        SINGLE(MOV_B32, w); SINGLE(ADD3_U32, w); SINGLE(SAD_U8, w); SINGLE(LSHL_ADD, w); SINGLE(XAD_U32, w);
        SINGLE(MAD_I32_I24, w); SINGLE(MBCNT, w);
        SINGLE(DS_READ_B32, w); SINGLE(DS_READ_B64, w);
    }
    // which kinds share an issue port?  2 and 8 waves per SIMD, half of them each kind
    for (int w = 2; w <= 8; w *= 4) {
        PAIR(FMA_F64, FMA_F64, w); PAIR(FMA_F64, FMA_F32, w); PAIR(FMA_F64, AND_B32, w); PAIR(FMA_F64, PERM_B32, w);
        PAIR(FMA_F64, DOT4_U8, w); PAIR(FMA_F64, MAD_U24, w); PAIR(FMA_F32, AND_B32, w); PAIR(FMA_F32, DOT4_U8, w);
        PAIR(DOT4_U8, PERM_B32, w); PAIR(FMA_F64, DS_READ_B32, w); PAIR(DOT4_U8, DS_READ_B32, w); PAIR(RCP_F64, FMA_F64, w);
        PAIR(RCP_F64, DOT4_U8, w); PAIR(CVT_F32_F64, FMA_F64, w);
    }
    return 0;
}
