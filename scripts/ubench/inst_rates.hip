// Per-instruction issue rates on gfx950 via inline asm (nothing can be optimised away).
// Prints SIMD cycles per wave64 instruction assuming the measured clock of v_add_f32 = 2 cycles...
// actually prints raw T lane-ops/s and ratio to v_add_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEFK(NAME, ASM2)                                                              \
__global__ __launch_bounds__(256) void NAME(uint32_t *out) {                          \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = threadIdx.x * 77u + 1u, c = 0x3effffffu;                              \
    for (int i = 0; i < ITERS; ++i) {                                                  \
        asm volatile(ASM2("%0") ASM2("%1") ASM2("%2") ASM2("%3") ASM2("%4") ASM2("%5") ASM2("%6") ASM2("%7") \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    }                                                                                  \
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;             \
}
#define A_ADDF(D) "v_add_f32 " D ", " D ", %8\n"
#define A_MULF(D) "v_mul_f32 " D ", " D ", %8\n"
#define A_FMA(D) "v_fma_f32 " D ", " D ", %8, %9\n"
#define A_DOT4(D) "v_dot4_i32_i8 " D ", " D ", %8, " D "\n"
#define A_PERM(D) "v_perm_b32 " D ", " D ", %8, %9\n"
#define A_BFI(D) "v_bfi_b32 " D ", %9, %8, " D "\n"
#define A_ANDOR(D) "v_and_or_b32 " D ", " D ", %8, %9\n"
#define A_MED3F(D) "v_med3_f32 " D ", " D ", %8, %9\n"
#define A_MAXF(D) "v_max_f32 " D ", " D ", %8\n"
#define A_MAXI(D) "v_max_i32 " D ", " D ", %8\n"
#define A_CVTFI(D) "v_cvt_f32_i32 " D ", " D "\n"
#define A_CVTIF(D) "v_cvt_i32_f32 " D ", " D "\n"
#define A_TRUNC(D) "v_trunc_f32 " D ", " D "\n"
#define A_RNDNE(D) "v_rndne_f32 " D ", " D "\n"
#define A_CVTPKU8(D) "v_cvt_pk_u8_f32 " D ", %8, 1, " D "\n"
#define A_ALIGNB(D) "v_alignbyte_b32 " D ", " D ", %8, 3\n"
#define A_ANDB(D) "v_and_b32 " D ", " D ", %8\n"
#define A_XORB(D) "v_xor_b32 " D ", " D ", %8\n"
#define A_ADDU(D) "v_add_u32 " D ", " D ", %8\n"
#define A_LSHLOR(D) "v_lshl_or_b32 " D ", " D ", 8, %8\n"
#define A_CNDMASK(D) "v_cndmask_b32 " D ", " D ", %8, vcc\n"
#define A_MADI24(D) "v_mad_i32_i24 " D ", " D ", %8, %9\n"
#define A_MULI24(D) "v_mul_i32_i24 " D ", " D ", %8\n"
#define A_BFE(D) "v_bfe_i32 " D ", " D ", 8, 8\n"
#define A_CVTUB0(D) "v_cvt_f32_ubyte0 " D ", " D "\n"
#define A_SDWA(D) "v_mul_i32_i24_sdwa " D ", %8, " D " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define A_LSHR(D) "v_lshrrev_b32 " D ", 8, " D "\n"
#define A_ADD3(D) "v_add3_u32 " D ", " D ", %8, %9\n"
DEFK(k_addf, A_ADDF) DEFK(k_mulf, A_MULF) DEFK(k_fma, A_FMA) DEFK(k_dot4, A_DOT4) DEFK(k_perm, A_PERM) DEFK(k_bfi, A_BFI)
DEFK(k_andor, A_ANDOR) DEFK(k_med3f, A_MED3F) DEFK(k_maxf, A_MAXF) DEFK(k_maxi, A_MAXI) DEFK(k_cvtfi, A_CVTFI) DEFK(k_cvtif, A_CVTIF)
DEFK(k_trunc, A_TRUNC) DEFK(k_rndne, A_RNDNE) DEFK(k_cvtpku8, A_CVTPKU8) DEFK(k_alignb, A_ALIGNB) DEFK(k_andb, A_ANDB) DEFK(k_xorb, A_XORB)
DEFK(k_addu, A_ADDU) DEFK(k_lshlor, A_LSHLOR) DEFK(k_cndmask, A_CNDMASK) DEFK(k_madi24, A_MADI24) DEFK(k_muli24, A_MULI24) DEFK(k_bfe, A_BFE)
DEFK(k_cvtub0, A_CVTUB0) DEFK(k_sdwa, A_SDWA) DEFK(k_lshr, A_LSHR) DEFK(k_add3, A_ADD3)
// packed f32: 64-bit register pairs
__global__ __launch_bounds__(256) void k_pkmul(uint32_t *out) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, b = {1.0001f, 0.9999f};
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n"
                     "v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    if (a0.x + a1.x + a2.x + a3.y == 1.2345f) out[0] = 1;
}
template <typename K> double run(const char *name, K kern, double base) {
    uint32_t *d; hipMalloc(&d, 4);
    const int grid = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 5.0 * grid * 4.0 * ITERS * 8;      // wave-instructions
    const double rate = insts / (ms * 1e-3);
    printf("%-22s %7.2f G wave-inst/s  %5.2fx v_add_f32 time\n", name, rate / 1e9, base > 0 ? base / rate : 1.0);
    hipFree(d);
    return rate;
}
int main() {
    double b = run("v_add_f32", k_addf, 0);
#define RUN(n) run(#n, n, b);
    RUN(k_mulf) RUN(k_fma) RUN(k_dot4) RUN(k_perm) RUN(k_bfi) RUN(k_andor) RUN(k_med3f) RUN(k_maxf) RUN(k_maxi) RUN(k_cvtfi) RUN(k_cvtif)
    RUN(k_trunc) RUN(k_rndne) RUN(k_cvtpku8) RUN(k_alignb) RUN(k_andb) RUN(k_xorb) RUN(k_addu) RUN(k_lshlor) RUN(k_cndmask) RUN(k_madi24)
    RUN(k_muli24) RUN(k_bfe) RUN(k_cvtub0) RUN(k_sdwa) RUN(k_lshr) RUN(k_add3)
    run("v_pk_mul+v_pk_add (x8)", k_pkmul, b);
    return 0;
}
