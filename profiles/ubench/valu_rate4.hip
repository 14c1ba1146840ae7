// Fourth microbenchmark (gfx950): which VALU ops issue at the f32 rate (2 cycles per wave instruction on a SIMD-32)
// and which at half of it.  Every op: 8 independent chains x 8 = 64 instructions per loop trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
// OPS(X): X(id, "label", "asm for chain register %N with helper operands %8 (vgpr b), %9 (vgpr c)")
#define CH(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define DEFOP(ID, TEXT)                                                                         \
  else if constexpr (OP == ID) {                                                                \
    asm volatile(REP8(TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7))          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(b), "v"(c) : "vcc");                                                     \
  }
#define T_ADD(N) "v_add_f32 %" #N ",%" #N ",%8\n"
#define T_SUB(N) "v_sub_f32 %" #N ",%" #N ",%8\n"
#define T_MUL(N) "v_mul_f32 %" #N ",%" #N ",%8\n"
#define T_FMA(N) "v_fma_f32 %" #N ",%" #N ",%8,%9\n"
#define T_FMAC(N) "v_fmac_f32 %" #N ",%8,%9\n"
#define T_MINF(N) "v_min_f32 %" #N ",%" #N ",%8\n"
#define T_MAXF(N) "v_max_f32 %" #N ",%" #N ",%8\n"
#define T_MINF_ABS(N) "v_min_f32_e64 %" #N ",|%" #N "|,%8\n"
#define T_ADD_ABS(N) "v_add_f32_e64 %" #N ",|%" #N "|,%8\n"
#define T_ADD_NEG(N) "v_add_f32_e64 %" #N ",-%" #N ",%8\n"
#define T_MIN3F(N) "v_min3_f32 %" #N ",%" #N ",%8,%9\n"
#define T_MAX3F(N) "v_max3_f32 %" #N ",%" #N ",%8,%9\n"
#define T_MED3F(N) "v_med3_f32 %" #N ",%" #N ",%8,%9\n"
#define T_MINU(N) "v_min_u32 %" #N ",%" #N ",%8\n"
#define T_MIN3U(N) "v_min3_u32 %" #N ",%" #N ",%8,%9\n"
#define T_AND(N) "v_and_b32 %" #N ",%" #N ",%8\n"
#define T_XOR(N) "v_xor_b32 %" #N ",%" #N ",%8\n"
#define T_OR(N) "v_or_b32 %" #N ",%" #N ",%8\n"
#define T_ADDU(N) "v_add_u32 %" #N ",%" #N ",%8\n"
#define T_LSHL(N) "v_lshlrev_b32 %" #N ",1,%" #N "\n"
#define T_MOV(N) "v_mov_b32 %" #N ",%8\n"
#define T_CNDVCC(N) "v_cndmask_b32 %" #N ",%" #N ",%8,vcc\n"
#define T_BFI(N) "v_bfi_b32 %" #N ",%8,%" #N ",%9\n"
#define T_ANDOR(N) "v_and_or_b32 %" #N ",%" #N ",%8,%9\n"
#define T_XAD(N) "v_xad_u32 %" #N ",%" #N ",%8,%9\n"
#define T_PERM(N) "v_perm_b32 %" #N ",%" #N ",%8,%9\n"
#define T_ALIGN(N) "v_alignbit_b32 %" #N ",%" #N ",%8,31\n"
#define T_ADD_DPP(N) "v_add_f32_dpp %" #N ",%" #N ",%8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_ADD_DPP_SHR(N) "v_add_f32_dpp %" #N ",%" #N ",%8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_MINF_DPP(N) "v_min_f32_dpp %" #N ",%" #N ",%8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_MAXF_DPP(N) "v_max_f32_dpp %" #N ",%" #N ",%8 row_mirror row_mask:0xf bank_mask:0xf\n"
#define T_XOR_DPP(N) "v_xor_b32_dpp %" #N ",%" #N ",%8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_MOV_DPP(N) "v_mov_b32_dpp %" #N ",%8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_ADD_SDWA(N) "v_add_f32_sdwa %" #N ",%" #N ",|%8| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define T_CMPF(N) "v_cmp_lt_f32 vcc,%" #N ",%8\n"
#define T_CMPEQU(N) "v_cmp_eq_u32 vcc,%" #N ",%8\n"
#define T_MULLEG(N) "v_mul_legacy_f32 %" #N ",%" #N ",%8\n"
#define T_SUBREV(N) "v_subrev_f32 %" #N ",%" #N ",%8\n"
#define T_LDEXP(N) "v_ldexp_f32 %" #N ",%" #N ",%8\n"
#define T_MADU24(N) "v_mad_u32_u24 %" #N ",%" #N ",%8,%9\n"
#define T_ADD3(N) "v_add3_u32 %" #N ",%" #N ",%8,%9\n"
#define T_OR3(N) "v_or3_b32 %" #N ",%" #N ",%8,%9\n"
#define T_LSHLOR(N) "v_lshl_or_b32 %" #N ",%" #N ",1,%9\n"
#define T_SUBU(N) "v_sub_u32 %" #N ",%" #N ",%8\n"
#define T_MAXI(N) "v_max_i32 %" #N ",%" #N ",%8\n"
#define T_CVT(N) "v_cvt_f32_i32 %" #N ",%" #N "\n"
#define T_MIX(N) "v_add_f32 %" #N ",%" #N ",%8\n v_min_f32 %" #N ",%" #N ",%9\n"

template <int OP>
__global__ void __launch_bounds__(256) ub(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP < 0) {}
    DEFOP(0, T_ADD) DEFOP(1, T_SUB) DEFOP(2, T_MUL) DEFOP(3, T_FMA) DEFOP(4, T_FMAC) DEFOP(5, T_MINF) DEFOP(6, T_MAXF)
    DEFOP(7, T_MINF_ABS) DEFOP(8, T_ADD_ABS) DEFOP(9, T_ADD_NEG) DEFOP(10, T_MIN3F) DEFOP(11, T_MAX3F) DEFOP(12, T_MED3F)
    DEFOP(13, T_MINU) DEFOP(14, T_MIN3U) DEFOP(15, T_AND) DEFOP(16, T_XOR) DEFOP(17, T_OR) DEFOP(18, T_ADDU)
    DEFOP(19, T_LSHL) DEFOP(20, T_MOV) DEFOP(21, T_CNDVCC) DEFOP(22, T_BFI) DEFOP(23, T_ANDOR) DEFOP(24, T_XAD)
    DEFOP(25, T_PERM) DEFOP(26, T_ALIGN) DEFOP(27, T_ADD_DPP) DEFOP(28, T_ADD_DPP_SHR) DEFOP(29, T_MINF_DPP)
    DEFOP(30, T_MAXF_DPP) DEFOP(31, T_XOR_DPP) DEFOP(32, T_MOV_DPP) DEFOP(33, T_ADD_SDWA) DEFOP(34, T_CMPF)
    DEFOP(35, T_CMPEQU) DEFOP(36, T_MULLEG) DEFOP(37, T_SUBREV) DEFOP(38, T_LDEXP) DEFOP(39, T_MADU24) DEFOP(40, T_ADD3)
    DEFOP(41, T_OR3) DEFOP(42, T_LSHLOR) DEFOP(43, T_SUBU) DEFOP(44, T_MAXI) DEFOP(45, T_CVT) DEFOP(46, T_MIX)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
static double clock_hz = 2.4e9;
template <int OP> void run(const char *name, int instr_per_iter = 64) {
  float *out; hipMalloc(&out, 256 * 64 * 256 * sizeof(float));
  const int iters = 4000;
  printf("%-34s", name);
  for (int w : {1, 2, 4, 8}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    ub<OP><<<256 * w, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a); ub<OP><<<256 * w, 256>>>(out, iters); hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  w=%d: %5.2f", w, ms * 1e-3 * clock_hz / ((double)iters * instr_per_iter * w));
  }
  printf("\n");
  fflush(stdout);
  hipFree(out);
}
int main() {
  printf("cycles (at an assumed 2.4 GHz) per wave instruction per SIMD, w = waves per SIMD\n");
  run<0>("v_add_f32"); run<1>("v_sub_f32"); run<2>("v_mul_f32"); run<3>("v_fma_f32"); run<4>("v_fmac_f32");
  run<5>("v_min_f32 (VOP2)"); run<6>("v_max_f32 (VOP2)"); run<7>("v_min_f32_e64 |abs|"); run<8>("v_add_f32_e64 |abs|");
  run<9>("v_add_f32_e64 -neg"); run<10>("v_min3_f32"); run<11>("v_max3_f32"); run<12>("v_med3_f32");
  run<13>("v_min_u32"); run<14>("v_min3_u32"); run<15>("v_and_b32"); run<16>("v_xor_b32"); run<17>("v_or_b32");
  run<18>("v_add_u32"); run<19>("v_lshlrev_b32"); run<20>("v_mov_b32"); run<21>("v_cndmask_b32 vcc"); run<22>("v_bfi_b32");
  run<23>("v_and_or_b32"); run<24>("v_xad_u32"); run<25>("v_perm_b32"); run<26>("v_alignbit_b32");
  run<27>("v_add_f32_dpp quad_perm"); run<28>("v_add_f32_dpp row_shr:1"); run<29>("v_min_f32_dpp quad_perm");
  run<30>("v_max_f32_dpp row_mirror"); run<31>("v_xor_b32_dpp quad_perm"); run<32>("v_mov_b32_dpp quad_perm");
  run<33>("v_add_f32_sdwa |abs|"); run<34>("v_cmp_lt_f32 vcc"); run<35>("v_cmp_eq_u32 vcc"); run<36>("v_mul_legacy_f32");
  run<37>("v_subrev_f32"); run<38>("v_ldexp_f32"); run<39>("v_mad_u32_u24"); run<40>("v_add3_u32"); run<41>("v_or3_b32");
  run<42>("v_lshl_or_b32"); run<43>("v_sub_u32"); run<44>("v_max_i32"); run<45>("v_cvt_f32_i32");
  run<46>("v_add_f32 + v_min_f32 alternating", 128);
  return 0;
}
