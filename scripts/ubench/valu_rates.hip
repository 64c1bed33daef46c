// Microbenchmark: sustained wave-instruction rates on gfx950 for the ops the epilogue
// and the depthwise inner loop are made of.  Prints lane-ops/s and the rate relative to
// v_fma_f32.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 2048
#define NACC 8

template <int OP>
__global__ __launch_bounds__(256) void bench(uint32_t *out, uint32_t seed) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    uint32_t a[NACC];
    float f[NACC];
    v2f pk[NACC];
    v2f pg = {1.0001f, 0.9999f};
    const uint32_t t = threadIdx.x + seed;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = t * (i + 3) + 1; f[i] = (float)(t + i) * 0.001f; pk[i] = v2f{f[i], f[i] + 1.0f}; }
    uint32_t w = t * 2654435761u | 1u;
    float g = 1.0001f, h = 0.49999997f;
    __shared__ uint32_t lds[4096];
    if (OP == 10 || OP == 11) { for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * seed; __syncthreads(); }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) f[i] = __builtin_fmaf(f[i], g, h);
            if (OP == 1) a[i] = (uint32_t)__builtin_amdgcn_sdot4((int)a[i], (int)w, (int)a[i], false);
            if (OP == 2) a[i] = __builtin_amdgcn_perm(a[i], w, 0x05040100u + i);
            if (OP == 3) f[i] = __fmul_rn(f[i], g);
            if (OP == 4) f[i] = __fadd_rn(f[i], h);
            if (OP == 5) { f[i] = (float)(int)a[i]; a[i] += (uint32_t)f[i]; }            // cvt_f32_i32 + cvt back + add (3 ops)
            if (OP == 6) f[i] = __builtin_amdgcn_fmed3f(f[i], g, h);
            if (OP == 7) f[i] = __builtin_copysignf(h, f[i]) ;
            if (OP == 8) a[i] = a[i] * w + 7u;                                            // v_mad_u32_u24? (mul_lo + add)
            if (OP == 9) a[i] = (uint32_t)(((int)(a[i] << 8)) >> 24) + w;  // bfe
            if (OP == 10) a[i] += lds[(a[i] + threadIdx.x) & 4095];                       // ds_read_b32 dependent
            if (OP == 11) a[i] += lds[(threadIdx.x + i * 256 + it) & 4095];               // ds_read_b32 streaming
            if (OP == 12) a[i] = (uint32_t)(((int)(a[i] << 8) >> 8) * ((int)(w << 8) >> 8)) + a[i]; // mad_i24
            if (OP == 13) { int v = (int)f[i]; a[i] ^= (uint32_t)v; }                       // cvt_i32_f32 + xor
            if (OP == 14) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "s"(0x80000000u), "v"(w));
            if (OP == 15) asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w));
            if (OP == 16) asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "s"(0x80u), "v"(w));
            if (OP == 17) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(a[i]) : "v"(f[i]), "v"(a[i]));
            if (OP == 18) asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 19) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "s"(0x80u), "v"(w));
            if (OP == 20) asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "s"(0x80u), "v"(w));
            if (OP == 21) asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w));
            if (OP == 22) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w));
            if (OP == 23) asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 24) asm volatile("v_trunc_f32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
            if (OP == 25) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w));
            if (OP == 26) asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(g));
            if (OP == 27) asm volatile("v_max_i32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w));
            if (OP == 29) asm volatile("v_or_b32 %0, 1, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 30) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "v"(f[i]), "v"(g));
            if (OP == 31) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(a[i]) : "v"(a[i]), "v"(g));
            if (OP == 32) asm volatile("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(a[i]) : "v"(w));
            if (OP == 33) asm volatile("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(a[i]) : "v"(f[i]));
            if (OP == 34) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w));
            if (OP == 35) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(w));
            if (OP == 36) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(g));
            if (OP == 37) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(f[i]) : "v"(f[i]), "v"(g));
            if (OP == 38) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(a[i]) : "v"(w));
            if (OP == 39) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(g), "v"(f[i]));
            if (OP == 40) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(w) : );
            if (OP == 41) asm volatile("v_min_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(g));
            if (OP == 42) asm volatile("v_rndne_f32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
            if (OP == 43) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 44) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(pk[i]), "v"(pg));
            if (OP == 45) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(pk[i]), "v"(pg));
            if (OP == 46) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pk[i]) : "v"(pk[i]), "v"(pg), "v"(pg));
            if (OP == 47) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pk[i]) : "v"(pk[i]), "v"(pg));
        }
        if (OP == 7) { h = -h; }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r ^= a[i] ^ __float_as_uint(f[i]) ^ __float_as_uint(pk[i].x) ^ __float_as_uint(pk[i].y);
    if (r == 0x12345678u) out[0] = r;
}

template <int OP> double run(const char *name, double ops_per_iter, double base) {
    uint32_t *d; hipMalloc(&d, 4);
    const int grid = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(256), 0, 0, d, 2u + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double laneops = 5.0 * grid * 256.0 * ITERS * NACC * ops_per_iter;
    const double rate = laneops / (ms * 1e-3);
    printf("%-28s %8.2f T lane-ops/s  (%.2fx fma)  %.3f ms\n", name, rate / 1e12, base > 0 ? rate / base : 1.0, ms / 5);
    hipFree(d);
    return rate;
}

int main() {
    double b = run<0>("v_fma_f32", 1, 0);
    run<1>("v_dot4_i32_i8", 1, b);
    run<2>("v_perm_b32", 1, b);
    run<3>("v_mul_f32", 1, b);
    run<4>("v_add_f32", 1, b);
    run<5>("cvt_f32_i32+cvt_u32+add", 3, b);
    run<6>("v_med3_f32", 1, b);
    run<7>("v_bfi copysign", 1, b);
    run<8>("mul_lo_u32 + add", 2, b);
    run<9>("shl+ashr+add (bfe)", 3, b);
    run<10>("ds_read_b32 dependent", 1, b);
    run<11>("ds_read_b32 streaming", 1, b);
    run<12>("v_mul_i24 + add (mad_i24)", 1, b);
    run<13>("cvt_i32_f32 + xor", 2, b);
    run<14>("v_and_or_b32", 1, b);
    run<15>("v_lshl_or_b32", 1, b);
    run<16>("v_xad_u32", 1, b);
    run<17>("v_cvt_pk_u8_f32", 1, b);
    run<18>("v_cvt_rpi_i32_f32", 1, b);
    run<19>("v_add3_u32", 1, b);
    run<20>("v_or3_b32", 1, b);
    run<21>("v_cvt_pk_i16_i32", 1, b);
    run<22>("v_pk_max_i16", 1, b);
    run<23>("v_sat_pk_u8_i16", 1, b);
    run<24>("v_trunc_f32", 1, b);
    run<25>("v_xor_b32", 1, b);
    run<26>("v_max_f32", 1, b);
    run<27>("v_max_i32", 1, b);
    run<29>("v_or_b32 x, 1", 1, b);
    run<30>("v_add_f32_sdwa BYTE_1 preserve", 1, b);
    run<31>("v_add_f32_sdwa WORD_0 pad", 1, b);
    run<32>("v_sat_pk_u8_i16_sdwa WORD_1", 1, b);
    run<33>("v_cvt_i32_f32_sdwa BYTE_1", 1, b);
    run<34>("v_and_b32", 1, b);
    run<35>("v_add_u32", 1, b);
    run<36>("v_add_f32 (asm)", 1, b);
    run<37>("v_add_f32_sdwa DWORD", 1, b);
    run<38>("v_mov_b32_sdwa byte insert", 1, b);
    run<39>("v_fmac_f32", 1, b);
    run<44>("v_pk_add_f32 (2 results)", 1, b);
    run<45>("v_pk_mul_f32 (2 results)", 1, b);
    run<46>("v_pk_fma_f32 (2 results)", 1, b);
    run<47>("v_pk_add_f32 op_sel_hi:[1,0]", 1, b);
    run<40>("v_cndmask_b32", 1, b);
    run<41>("v_min_f32", 1, b);
    run<42>("v_rndne_f32", 1, b);
    run<43>("v_lshlrev_b32", 1, b);
    return 0;
}
