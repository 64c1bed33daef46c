// k_common.hpp -- what every kernel family of libmicroflow_amd shares (gfx950 / CDNA4 only).
//
// The kernels live in one file per family -- k_generic.hip, k_depthwise.hip, k_pointwise.hip,
// k_fused_mm.hip, k_gemm.hip, ... -- each citing the upstream file:line it implements.  All of them
// share one arithmetic contract:
//
//   acc (i32)  = sum over ALL window taps of (v' - izp) * (w - wzp)
//                with v' = izp at out-of-range taps          [== x0 - x1 - k2 + k3]
//   y          = sat_i8(roundf((f32(ozp) + c0[c]) + c1[c] * f32(acc)))   then activation
//
// The device never evaluates `izp`-dependent terms per pixel: the host folds
//   Kc[c] = -izp * sum_all_taps w[c] + T * izp * wzp[c]   (T = taps contributing to c)
// so that acc = dot(v', w) - wzp[c] * sum(v') + Kc[c]; padding is realised by
// filling the halo with izp, which makes every pixel -- border or interior -- run
// the same code.  A[c] = fl32(f32(ozp) + c0[c]) and S[c] = c1[c or 0] are folded on
// the host too, and the activation is a clamp [lo, hi] (relu: lo = ozp; relu6:
// hi = quantize(6.0)).
//
// f32 rules: no contraction (this file is built with -ffp-contract=off AND uses the
// explicit __fmul_rn/__fadd_rn/__fdiv_rn forms), int->float is v_cvt_f32_i32 (RNE),
// roundf is trunc(x + copysign(pred(0.5), x)) which is exact for every finite x.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kernels.hpp"
#include "mf_switches.hpp"

namespace mf {
namespace k {

typedef int v4i __attribute__((ext_vector_type(4)));
// native vector type for register-resident staging arrays: arrays of HIP's struct-based
// uint4 that are conditionally re-assigned are NOT promoted to registers by hipcc (they
// end up in scratch memory); ext_vector_type arrays are.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------
// The workgroup barrier of every kernel here -- NOT a bare __syncthreads().  __syncthreads() is fence(release) + s_barrier +
// fence(acquire), and the `s_waitcnt lgkmcnt(0)` in front of s_barrier comes from the release fence as a SOFT wait: hipcc's
// wait-count pass deletes it wherever its scoreboard shows no LDS operation pending.  At a loop header the pass first sees the
// pre-header's state only, deletes the wait, and does not re-create it when the back edge then brings the previous iteration's
// pending ds_writes (gfx950 backs off barriers, so the pass never forces a wait in front of one either).  Round 6: quad_mm's
// barrier at the top of a step was reached with the B.pw stores of the step before still in flight, and the output copy behind
// it read stale bytes in about one launch of twenty; every step-queue kernel had its `slot` store pending at the same place.
// The explicit wait cannot be deleted.  scripts/asm_barrier_waits.py scans the listings for barriers reached with LDS operations
// pending (tests/test_host_logic.py keeps it at zero outside the GEMM's counted-wait schedule).
// MF_JITTER_DIAG=n (a diagnostic build, never shipped): every wave sleeps a pseudo-random 0 .. 64 n cycles in front of every barrier,
// behind it and in front of every LDS-DMA issue.  A correct kernel's results cannot depend on it; a kernel that relies on waves
// arriving "soon enough" (a missing barrier or wait) fails the race screens far more often (scripts/r06_jitter.sh).
#ifndef MF_JITTER_DIAG
#define MF_JITTER_DIAG 0
#endif
__device__ __forceinline__ void mf_jitter() {
#if MF_JITTER_DIAG
    uint32_t t = (uint32_t)__builtin_readcyclecounter();
    t = (t ^ (t >> 7) ^ (threadIdx.x >> 6) * 0x9E3779B1u) * 0x85EBCA6Bu;
    const uint32_t n = __builtin_amdgcn_readfirstlane(t >> 28); // 0 .. 15
    for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(MF_JITTER_DIAG);
#endif
}
#ifndef MF_SYNC_KO
#define MF_SYNC_KO 0 // 1: the bare __syncthreads() again (WRONG: the positive control of the race screens, never shipped)
#endif
__device__ __forceinline__ void wg_sync() {
    mf_jitter();
#if !MF_SYNC_KO
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    mf_jitter();
}

__device__ __forceinline__ int requant(int acc, float A, float S, float lo_f, float hi_f) {
    // (f32(ozp) + c0) + c1 * f32(acc): two roundings, like the reference (conv_2d.rs:93-98)
    const float x = __fadd_rn(A, __fmul_rn(S, (float)acc));
    // libm::roundf = half away from zero = trunc(x + copysign(0x1.fffffep-2, x))
    float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    // saturating cast + activation clamp, done in f32 so that the int conversion below
    // is always in range (it truncates toward zero)
    r = __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
    return (int)r;
}

// "Magic" accumulators (MG = true): the accumulator is started at Kc + 0x4B400000, so that its
// bit pattern read as f32 is 12582912 + acc, exactly, whenever |acc| < 2^22; f32(acc) is then
// one v_add_f32 (full rate) instead of v_cvt_f32_i32 (half rate).  Both are exact, so the
// epilogue's value is unchanged.  The host enables it per operator only when the worst-case
// |acc| over ALL inputs -- max|v - izp| * sum|w - wzp| per channel -- is below 2^22 (ops.hip).
// MG (template parameter of every fast kernel) = the epilogue mode the host chose for the operator(s) of a launch:
//   0  accumulator converted with v_cvt_f32_i32                         (|acc| may reach 2^22)
//   1  bit-pattern int -> f32 (above)
//   2  = 1, and the clamp [lo, hi] is exactly the element type's range and |x| < 2^15 for every input, so that the
//      clamp is done by a saturating pack (v_sat_pk_u8_i16) instead of v_med3_f32 (requant_pack4 below)
//   3  the single-fma form: the host found, per channel, (S', C', d) whose staircase  acc -> v_cvt_pk_u8_f32(v_fma_f32(S', F, C'))
//      -- both instructions executed with MODE.FP_ROUND = toward zero, which the kernel sets on entry (epi_enter) -- equals the
//      reference's on every accumulator the operator can produce (epi_fma.cpp: search; k_generic.hip verify_fma_form: exhaustive
//      check on the device).  The kernels then receive C' in place of A, S' in place of S and Kc + d in place of Kc; the clamp is
//      the element type's whole range (the conversion saturates).  Two instructions per byte.
template <int MG>
__device__ __forceinline__ int requant_t(int acc, float A, float S, float lo_f, float hi_f) {
    static_assert(MG != 3, "the single-fma form exists for the packed epilogues only (requant_pack4)");
    if constexpr (MG == 0) {
        return requant(acc, A, S, lo_f, hi_f);
    } else {
        const float f = __fsub_rn(__int_as_float(acc), 12582912.0f);
        const float x = __fadd_rn(A, __fmul_rn(S, f));
        float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
        r = __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
        return (int)r;
    }
}
// Mode 3 kernels run with the f32 rounding mode TOWARD ZERO for their whole life (MODE is per-wave state, initialised from the
// kernel descriptor at every wave launch, so nothing has to be restored).  Both v_fma_f32 and v_cvt_pk_u8_f32 follow it (measured
// over all 2^32 inputs: scripts/ubench/cvt_pk_probe.hip, mf_selftest_cvt_pk), which makes the form y = floor(S' F + C') of the
// EXACT real value -- the step for output k sits exactly where S' F + C' reaches k, with no grid or tie effects of its own.  (In
// the default mode the conversion rounds half to even and the steps of odd and even k shift against each other by an f32
// spacing: fewer channels have a solution.)  Such a kernel must contain no other f32 arithmetic that needs round-to-nearest.
template <int MG> __device__ __forceinline__ void epi_enter() {
    if constexpr (MG == 3) __builtin_amdgcn_s_setreg(0x801 /* hwreg(HW_REG_MODE, 0, 2): FP_ROUND, single precision */, 3u /* toward zero */);
}
// Patched accumulators of mode 3 (kernels.hpp EpiPatchRec): `rec` is the record of the tile the four accumulators belong to (wave-
// uniform: scalar registers), g the lane's 16-lane group.  P == 0 -- no patched channel in this tile -- is the common case and
// costs one scalar compare and branch.
#ifndef MF_EPI_PATCH_KO
#define MF_EPI_PATCH_KO 0 // 1: knock-out timing experiment (WRONG results where an operator has patched accumulators): no patch code
#endif
__device__ __forceinline__ EpiPatchRec epi_patch_load(const EpiPatchRec *tab, int idx) { // a scalar load (idx is wave-uniform)
    typedef __attribute__((address_space(4))) const EpiPatchRec c_rec;
    if (MF_EPI_PATCH_KO || !tab) return EpiPatchRec{0, 0};
    c_rec *t = (c_rec *)(uintptr_t)tab;
    if (MF_EPI_PATCH_KO == 2) return EpiPatchRec{t[idx].P & 0, t[idx].meta}; // (knock-out 2: the loads and branches, never the replacement)
    return EpiPatchRec{t[idx].P, t[idx].meta};
}
__device__ __forceinline__ void epi_patch_apply(v4i &a, const EpiPatchRec &rec, int g) {
    if (MF_EPI_PATCH_KO) return;
    if (__builtin_expect(rec.P != 0, 0)) { // this tile has a patched channel (scalar branch; zero for almost every tile)
        // ... whose accumulator is P about once in 10^5 values: four compares and one more scalar branch decide that NO lane holds
        // P in any of its four accumulators; only then does the replacement itself run
        const bool any = (a[0] == rec.P) | (a[1] == rec.P) | (a[2] == rec.P) | (a[3] == rec.P);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0ull, 0)) {
            asm volatile("" ::: "memory"); // (keeps the block from being speculated or turned into selects on the common path)
            const bool grp = g == ((rec.meta >> 2) & 3);
            const int delta = (rec.meta & 16) ? -1 : 1, reg = rec.meta & 3;
            // (four selects with a uniform "is this the register" folded in, NOT a switch over the register: hipcc 7.2 was seen to
            // miscompile the uniform switch here -- every accumulator of the tile came out wrong; scripts/layer_parity.py found it)
            a[0] += (grp && a[0] == rec.P) ? (reg == 0 ? delta : 0) : 0;
            a[1] += (grp && a[1] == rec.P) ? (reg == 1 ? delta : 0) : 0;
            a[2] += (grp && a[2] == rec.P) ? (reg == 2 ? delta : 0) : 0;
            a[3] += (grp && a[3] == rec.P) ? (reg == 3 ? delta : 0) : 0;
        }
    }
}
template <int MG> __device__ __forceinline__ int4 magic4(int4 k) {
    if constexpr (MG != 0) k.x += MF_MAGIC_I, k.y += MF_MAGIC_I, k.z += MF_MAGIC_I, k.w += MF_MAGIC_I;
    return k;
}

// The shape-generic kernels also honour Rust's `NaN as i8 == 0` (a NaN can only come from
// non-finite constants, i.e. a degenerate model); operators with non-finite constants are never
// routed to the shape-specialised kernels (ops.hip), whose requant() skips this test.
__device__ __forceinline__ int requant_any(int acc, float A, float S, float lo_f, float hi_f) {
    const float x = __fadd_rn(A, __fmul_rn(S, (float)acc));
    float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    r = (r != r) ? 0.0f : r; // NaN -> 0, then the activation clamp like any other value
    r = __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
    return (int)r;
}

// Output stores of the fast kernels.  NT = non-temporal: the written tensor is far larger than the L2 and is not read
// again by this launch.  A plain copy gains 2 .. 12 % from it at these sizes (scripts/ubench/hbm_copy.hip); the layer-wise
// kernels gain 2 .. 20 % where a wave's store instruction covers whole lines, and lose 3 .. 8 % where it does not (the
// pointwise kernels with N = 2K, whose 128-byte lines are completed by two instructions), so each kernel picks
// (profiles/r03/nt_store_ab.txt).  The fused kernels are bounded by the VALU and do not care: MF_NT_STORE (default 0).
#ifndef MF_NT_STORE
#define MF_NT_STORE 0
#endif
template <bool NT> __device__ __forceinline__ void st_out_t(void *p, uint4 v) {
    // (a native vector store in both forms: assigning HIP's struct-based uint4 through a pointer was seen split into four dword
    // stores when the value came straight out of an LDS load -- the stage kernel's output copy, round 4)
    if constexpr (NT) __builtin_nontemporal_store(v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w}, (v4i *)p);
    else *(v4i *)p = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
}
template <bool NT> __device__ __forceinline__ void st_out_t(void *p, uint2 v) {
    typedef int v2i_ __attribute__((ext_vector_type(2)));
    if constexpr (NT) __builtin_nontemporal_store(v2i_{(int)v.x, (int)v.y}, (v2i_ *)p);
    else *(v2i_ *)p = v2i_{(int)v.x, (int)v.y};
}
template <bool NT> __device__ __forceinline__ void st_out_t(void *p, uint32_t v) {
    if constexpr (NT) __builtin_nontemporal_store(v, (uint32_t *)p);
    else *(uint32_t *)p = v;
}
template <typename V> __device__ __forceinline__ void st_out(void *p, V v) { st_out_t<MF_NT_STORE != 0>(p, v); }

// 4 ints in [-128,127] -> one dword of int8 (byte 0 = a)
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// ------------------------------------------------------------------------
// The epilogue of the fast kernels: four accumulators -> one packed dword.
//
// x = fl(A + fl(S * f32(acc))) is the reference's value (two roundings, conv_2d.rs:93-98); what follows it in the
// reference is roundf (half away from zero), the activation clamp and `as T`.  The exact forms of that tail (round 2's
// x + copysign(pred(0.5), x); v_med3; truncating v_cvt form -- 10 issue units per byte -- lives on in requant() for the
// shape-generic kernels only):
//
// Modes 0 / 1, "sticky bit": roundf(x) == RNE_int(x | 1) for every finite x with |x| < 2^22, where `x | 1` sets the
//   lowest mantissa bit.  Proof sketch: a tie k + 1/2 has an even mantissa (its ulp is < 1/2), so the OR moves it one
//   ulp AWAY from zero and round-to-nearest then goes away from zero, as roundf does; a non-tie x with an even
//   mantissa moves by one ulp towards the next float of the same sign, and the nearest tie (even mantissa, at least
//   two ulps away) is not reached; odd mantissas are unchanged; 0 becomes a denormal that rounds to 0.  RNE_int itself
//   is the f32 addition of 1.5 * 2^23, whose result's low mantissa bits are the two's-complement integer, so the
//   low BYTE of the sum -- written to its place in the dword by the SDWA destination select of that same v_add_f32 --
//   is the stored value.  The clamp moves in front (clamping to integers commutes with rounding to integers):
//   v_sub (bit pattern -> f32), v_mul, v_add, v_med3, v_or, v_add_sdwa: 7 issue units per byte.
//   mf_selftest_rounding (k_generic.hip) checks the identity over all 2^32 bit patterns on the device.
//
// MG == 2 (host: clamp == the element type's whole range, |x| < 2^15): the magic constant carries +128 (i8), the low
//   16 bits of two sums are written into the halves of one dword, and v_sat_pk_u8_i16 saturates both to [0, 255] =
//   the clamp; XOR 0x80 per byte returns to the stored i8 domain.  No v_med3: 6.25 issue units per byte.
// ------------------------------------------------------------------------
template <int MG> __device__ __forceinline__ float requant_x(int acc, float A, float S) {
    static_assert(MG != 3, "mode 3 has no two-rounding value: epi_value");
    float f;
    if constexpr (MG != 0) f = __fsub_rn(__int_as_float(acc), 12582912.0f);
    else f = (float)acc;
    return __fadd_rn(A, __fmul_rn(S, f));
}
// x + copysign(pred(0.5), x), clamped: the value whose truncation is the reference's result (for epilogues that go on with
// the integer instead of packing it: the pooled sum of k_tail3.hip)
template <int MG> __device__ __forceinline__ float requant_clamped(int acc, float A, float S, float lo_f, float hi_f) {
    const float x = requant_x<MG>(acc, A, S);
    const float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    return __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
}
// gfx950 has a destination-forwarding hazard on partial (dst_sel) writes: an instruction that reads a VGPR in the
// issue slot right after an SDWA byte write of it may see the old value (observed: an MFMA fed such a dword).  hipcc
// pads this for its own instructions but cannot see into inline asm, so the packs below are ONE asm block each in which
// every partial write is followed by an independent instruction (the OR of a later value, the other chain of a two-dword
// pack, or s_nop): nothing the compiler schedules next to them can break the rule.
// ---- sticky-bit forms ----
// RNE_int of four values (|v| < 2^22, lowest mantissa bit set) into the four bytes of a dword: the low byte of
// v + 1.5 * 2^23.  Every SDWA write is followed by an independent instruction (the OR of a later value / s_nop).
__device__ __forceinline__ uint32_t rne_pack4(float r0, float r1, float r2, float r3) {
    uint32_t d;
    const float M = 12582912.0f;
    asm("v_or_b32 %1, 1, %1\n\t"
        "v_or_b32 %2, 1, %2\n\t"
        "v_add_f32_sdwa %0, %1, %5 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %3, 1, %3\n\t"
        "v_add_f32_sdwa %0, %2, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %4, 1, %4\n\t"
        "v_add_f32_sdwa %0, %3, %5 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_add_f32_sdwa %0, %4, %5 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(d), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(M));
    return d;
}
__device__ __forceinline__ void rne_pack4x2(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3,
                                            uint32_t &da, uint32_t &db) {
    const float M = 12582912.0f;
    asm("v_or_b32 %2, 1, %2\n\t"
        "v_or_b32 %6, 1, %6\n\t"
        "v_or_b32 %3, 1, %3\n\t"
        "v_add_f32_sdwa %0, %2, %10 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %7, 1, %7\n\t"
        "v_add_f32_sdwa %1, %6, %10 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %4, 1, %4\n\t"
        "v_add_f32_sdwa %0, %3, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %8, 1, %8\n\t"
        "v_add_f32_sdwa %1, %7, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %5, 1, %5\n\t"
        "v_add_f32_sdwa %0, %4, %10 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %9, 1, %9\n\t"
        "v_add_f32_sdwa %1, %8, %10 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_add_f32_sdwa %0, %5, %10 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_add_f32_sdwa %1, %9, %10 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(da), "=&v"(db), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
        : "v"(M));
}
// MG == 2: the same rounding with the clamp done by a saturating pack.  MS = 1.5 * 2^23 + (128 for i8, 0 for u8):
// the low 16 bits of v + MS are the value in the u8 domain as an i16 (|v| < 2^15), two of them per dword;
// v_sat_pk_u8_i16 clamps both to [0, 255] and packs them into 16 bits.  Result: four u8-domain bytes.
__device__ __forceinline__ uint32_t sat_pack4(float r0, float r1, float r2, float r3, float MS) {
    uint32_t d, p, q;
    asm("v_or_b32 %3, 1, %3\n\t"
        "v_or_b32 %5, 1, %5\n\t"
        "v_add_f32_sdwa %1, %3, %7 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %4, 1, %4\n\t"
        "v_add_f32_sdwa %2, %5, %7 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %6, 1, %6\n\t"
        "v_add_f32_sdwa %1, %4, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_add_f32_sdwa %2, %6, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_sat_pk_u8_i16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(d), "=&v"(p), "=&v"(q), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(MS));
    return d;
}
__device__ __forceinline__ void sat_pack4x2(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3,
                                            float MS, uint32_t &da, uint32_t &db) {
    uint32_t p, q, r, s;
    asm("v_or_b32 %6, 1, %6\n\t"
        "v_or_b32 %8, 1, %8\n\t"
        "v_or_b32 %10, 1, %10\n\t"
        "v_add_f32_sdwa %2, %6, %14 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %12, 1, %12\n\t"
        "v_add_f32_sdwa %3, %8, %14 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %7, 1, %7\n\t"
        "v_add_f32_sdwa %4, %10, %14 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %9, 1, %9\n\t"
        "v_add_f32_sdwa %5, %12, %14 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %11, 1, %11\n\t"
        "v_add_f32_sdwa %2, %7, %14 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_or_b32 %13, 1, %13\n\t"
        "v_add_f32_sdwa %3, %9, %14 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_add_f32_sdwa %4, %11, %14 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_add_f32_sdwa %5, %13, %14 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_sat_pk_u8_i16_sdwa %0, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
        "v_sat_pk_u8_i16_sdwa %1, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
        "v_sat_pk_u8_i16_sdwa %0, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_sat_pk_u8_i16_sdwa %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(da), "=&v"(db), "=&v"(p), "=&v"(q), "=&v"(r), "=&v"(s), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0),
          "+v"(b1), "+v"(b2), "+v"(b3)
        : "v"(MS));
}

// The value an epilogue hands to its pack (epi_pack4 / epi_pack4x2): kernels that interleave the epilogue with MFMAs
// by hand (k_stage.hip, k_tail3.hip) call the two halves themselves.
template <int MG> __device__ __forceinline__ float epi_value(int acc, float A, float S, float lo_f, float hi_f) {
    if constexpr (MG == 3) return __fmaf_rn(S, __int_as_float(acc), A); // S = S', A = C', acc carries the pivot: ONE rounding
    else if constexpr (MG == 2) return requant_x<MG>(acc, A, S); // clamp = the saturating pack
    else return __builtin_amdgcn_fmed3f(requant_x<MG>(acc, A, S), lo_f, hi_f);
}
// MG == 3: conversion in the current rounding mode (toward zero: truncation) + saturation to [0, 255] + byte insert in one
// instruction per value (VOP3, whole-dword write: no SDWA hazard, free for the compiler to schedule)
__device__ __forceinline__ uint32_t cvtpk_pack4(float r0, float r1, float r2, float r3) {
    uint32_t d = __builtin_amdgcn_cvt_pk_u8_f32(r0, 0u, 0u);
    d = __builtin_amdgcn_cvt_pk_u8_f32(r1, 1u, d);
    d = __builtin_amdgcn_cvt_pk_u8_f32(r2, 2u, d);
    return __builtin_amdgcn_cvt_pk_u8_f32(r3, 3u, d);
}
template <int MG, uint32_t XR4> __device__ __forceinline__ uint32_t epi_pack4(float r0, float r1, float r2, float r3) {
    if constexpr (MG == 3) return cvtpk_pack4(r0, r1, r2, r3) ^ 0x80808080u; // u8 domain -> the stored i8 domain, either element type
    else if constexpr (MG == 2) return sat_pack4(r0, r1, r2, r3, XR4 ? 12582912.0f : 12583040.0f) ^ 0x80808080u;
    else return rne_pack4(r0, r1, r2, r3) ^ XR4;
}
template <int MG, uint32_t XR4>
__device__ __forceinline__ void epi_pack4x2(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3,
                                            uint32_t &da, uint32_t &db) {
    if constexpr (MG == 3) {
        da = cvtpk_pack4(a0, a1, a2, a3) ^ 0x80808080u, db = cvtpk_pack4(b0, b1, b2, b3) ^ 0x80808080u;
    } else if constexpr (MG == 2) {
        sat_pack4x2(a0, a1, a2, a3, b0, b1, b2, b3, XR4 ? 12582912.0f : 12583040.0f, da, db);
        da ^= 0x80808080u, db ^= 0x80808080u;
    } else {
        rne_pack4x2(a0, a1, a2, a3, b0, b1, b2, b3, da, db);
        da ^= XR4, db ^= XR4;
    }
}
template <int MG, uint32_t XR4>
__device__ __forceinline__ uint32_t requant_pack4(int a0, int a1, int a2, int a3, const float4 &A, const float4 &S,
                                                  float lo_f, float hi_f) {
    return epi_pack4<MG, XR4>(epi_value<MG>(a0, A.x, S.x, lo_f, hi_f), epi_value<MG>(a1, A.y, S.y, lo_f, hi_f),
                              epi_value<MG>(a2, A.z, S.z, lo_f, hi_f), epi_value<MG>(a3, A.w, S.w, lo_f, hi_f));
}

// requant_pack4 of two accumulator quads (two dwords) with the alternating chains of the x2 packs
template <int MG, uint32_t XR4>
__device__ __forceinline__ void requant_pack4x2(const v4i &a, const float4 &aA, const float4 &aS, const v4i &b, const float4 &bA,
                                                const float4 &bS, float lo_f, float hi_f, uint32_t &da, uint32_t &db) {
    epi_pack4x2<MG, XR4>(epi_value<MG>(a[0], aA.x, aS.x, lo_f, hi_f), epi_value<MG>(a[1], aA.y, aS.y, lo_f, hi_f),
                         epi_value<MG>(a[2], aA.z, aS.z, lo_f, hi_f), epi_value<MG>(a[3], aA.w, aS.w, lo_f, hi_f),
                         epi_value<MG>(b[0], bA.x, bS.x, lo_f, hi_f), epi_value<MG>(b[1], bA.y, bS.y, lo_f, hi_f),
                         epi_value<MG>(b[2], bA.z, bS.z, lo_f, hi_f), epi_value<MG>(b[3], bA.w, bS.w, lo_f, hi_f), da, db);
}

// pack4 for either element type: XR4 = 0x80808080 moves u8-domain epilogue results (0..255) back
// to the stored i8 domain (kernels.hpp), XR4 = 0 is plain i8 (the XOR disappears at compile time)
template <uint32_t XR4>
__device__ __forceinline__ uint32_t pack4x(int a, int b, int c, int d) {
    return pack4(a, b, c, d) ^ XR4;
}

__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
}
// First tap of an accumulator: d = dot4(a, b) + c with c a value that stays live (the folded
// constant Kc).  hipcc otherwise copies c into d and uses the destructive v_dot4c (one v_mov per
// accumulator per task); the three-address form needs no copy.
__device__ __forceinline__ int sdot4_first(uint32_t a, uint32_t b, int c) {
    int d;
    asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ------------------------------------------------------------------------
// HBM -> LDS staging for the depthwise kernels: LDS-DMA (`global_load_lds_dwordx4`,
// gfx950), 16 bytes per lane straight into LDS with no VGPR round trip.
//
// Why not registers: (1) a predicated register prefetch makes hipcc branch around every
// element and wait vmcnt(0) after each one; (2) an array of HIP's struct-based uint4 that
// is conditionally re-assigned is not promoted to registers (it went to scratch); (3) even
// with both fixed, hipcc flushed vmcnt(0) in the pre-header of the compute loop, i.e. the
// prefetch never overlapped the compute.  A DMA has no destination register, so nothing
// waits on it except the one explicit `s_waitcnt vmcnt(0)` + barrier per step below.
// The LDS destination of a DMA instruction is wave-uniform base + lane * 16.
// ------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// MF_DMA_ASM (default 1): the instruction is issued from inline asm, so that hipcc's wait-count pass does NOT know an LDS-DMA is
// in flight.  With the builtin it orders every later LDS *store* of the wave behind the DMA (it cannot tell the two LDS ranges
// apart: `s_waitcnt vmcnt(0)` in front of the first ds_write after a DMA issue -- the HBM round trip of a prefetch that was meant to
// fly under the phase) and completes every __syncthreads() fence with vmcnt(0) while one is outstanding.  Every kernel here waits for
// its staged tile itself (`s_waitcnt vmcnt(..)` in front of the barrier that hands the tile over: the builtin's bookkeeping was
// never what made them correct), and hidden vector-memory operations only make the compiler's own counted waits stricter (vmcnt
// retires in order).  Round 6, same box: the five-operator launch 0.81 -> 0.71 ms, scripts/asm_dma_waits.py lists the waits.
#ifndef MF_DMA_ASM
#define MF_DMA_ASM 1
#endif
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
__device__ __forceinline__ void dma16(const int8_t *src_lane, uint8_t *lds_wave_base) {
    mf_jitter();
#if MF_DMA_ASM
    // M0 = the wave-uniform LDS base; one wait state between an SALU write of M0 and the LDS-DMA that reads it
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t *)lds_wave_base);
    // (no "memory" clobber: the barriers and explicit waits on either side of a staged tile's life carry one, and a clobber here
    // would pin every LDS read of the phase the DMA flies under behind its issue)
#if MF_DMA_ASM == 2
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src_lane), "s"(base) : "memory");
#else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src_lane), "s"(base));
#endif
#else
    __builtin_amdgcn_global_load_lds((gbl_void_t *)src_lane, (lds_void_t *)lds_wave_base, 16, 0, 0);
#endif
}
// The same with the source as (wave-uniform base in scalar registers) + (32-bit lane offset): no 64-bit address pair per lane and
// per piece to keep or re-compute (pair3_tail's FRONT instance spilled its five hoisted piece addresses).
__device__ __forceinline__ void dma16_sb(const int8_t *src_wave, uint32_t lane_off, uint8_t *lds_wave_base) {
    mf_jitter();
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u8_t *)lds_wave_base);
    // (both halves through v_readfirstlane: hipcc does not legalise a 64-bit "s" operand that its allocator left in vector registers)
    const uint64_t a = (uint64_t)(uintptr_t)src_wave;
    const uint64_t sa = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32; // (the builtin returns a signed int)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(sa), "s"(base));
}

// ------------------------------------------------------------------------
// Dynamic step queue of the persistent kernels.
//
// A persistent kernel launches exactly the resident workgroups and walks "steps" (G images each).  With the steps
// dealt statically (step += gridDim.x) the workgroups of one CU do NOT finish together: the CU's VALU issue is
// arbitrated by wave age, so the workgroup that was dispatched first runs ahead and the younger one finishes up to
// 1.4x later -- alone on its CU, at less than half the CU's throughput (measured on the stage kernel, round 3: half of
// the 512 workgroups ended at 560-590 us, the other half at 780-820 us).  Here only the first two steps of a workgroup
// are static; every further step index is drawn from a device counter, two iterations ahead so that the next step's
// staging DMA can still be issued a whole step early.  The last workgroup to leave resets the counters, so a launch
// needs no memset.  `ctr` = DynSteps::INTS ints in device memory, zero before the first launch.
// ------------------------------------------------------------------------
#ifndef MF_DYNQ
#define MF_DYNQ 1 // 0: static striding (A/B switch)
#endif
struct DynSteps {
    static constexpr int HEADS = 8, PITCH = 32, INTS = HEADS * PITCH + PITCH; // 8 counters on their own 128-byte lines + [done]
    int *slot; // LDS: two ints
    int *ctr;  // device: ctr[h * PITCH] = draws from head h; ctr[HEADS * PITCH] = workgroups finished
    int step, nxt, nn, it, fetched;
    int K, nheads, head;                  // steps per draw; counters in use (1 or 8) and this workgroup's
    int S0;                               // the first S0 steps of every workgroup are static (blockIdx.x + i gridDim.x), drawn ones follow
    int pool_next, pool_left, sel, write_at;
    // cfg = K | nheads << 8 (dq_config below).  A draw is a chunk of K consecutive steps and has K iterations to return.
    // One device-wide counter sustains ~90 draws per microsecond (MI355X_MICROARCH.md) while the kernels with one-image
    // steps run 200 steps per microsecond, and a returning device-scope atomic takes 1-3 us under load -- as long as one
    // of their steps; hence K > 1 and/or 8 counters for those (workgroup b draws from head b % 8: workgroups are dealt
    // round-robin over the 8 XCDs, so a head is mostly one XCD's line; head h's d-th draw starts at step
    // 2 grid + K (8 d + h)).  Kernels with long steps use one counter and K = 1, which balances best.
    // Call before the kernel's first __syncthreads().
    __device__ __forceinline__ void init(uint8_t *lds_slot, int *counters, int tid, int cfg) {
        slot = (int *)lds_slot, ctr = counters, step = blockIdx.x, nxt = blockIdx.x + gridDim.x, nn = 0, it = 0, fetched = 0;
        K = cfg & 0xff, nheads = (cfg >> 8) & 0xff; // K == 0: static striding
        // (a power of two -- 1 or 8 from dq_config -- and a MASK, not `%`: hipcc lowers a modulo by a run-time value through f32
        // arithmetic, which a kernel that has switched its rounding mode (epi_enter) must not contain)
        nheads = nheads >= 8 ? 8 : (nheads >= 4 ? 4 : (nheads >= 2 ? 2 : 1));
        head = (int)blockIdx.x & (nheads - 1);
        S0 = cfg >> 16;
        if (S0 < 2) S0 = 2;
        pool_next = 0, pool_left = 0, sel = 0, write_at = -1;
        if (MF_DYNQ && K != 0 && tid == 0) slot[0] = draw();
    }
    __device__ __forceinline__ int draw() { return index_of(atomicAdd(ctr + head * PITCH, 1)); }
    // first step of the d-th chunk drawn from this workgroup's head
    __device__ __forceinline__ int index_of(int d) const { return S0 * (int)gridDim.x + K * (nheads * d + head); }
    // right after the barrier at the top of an iteration
    __device__ __forceinline__ void top(int tid) {
        if (MF_DYNQ && K != 0) {
            if (it + 2 < S0) {    // still inside the static prefix: the step after next is a stride away
                nn = nxt + (int)gridDim.x;
            } else if (pool_left == 0) { // start the chunk drawn earlier, draw the one after it (due K iterations from now)
                nn = __builtin_amdgcn_readfirstlane(slot[sel]);
                pool_next = nn + 1, pool_left = K - 1, sel ^= 1, write_at = it + K - 1;
                // The RAW result of the atomic is kept and nothing is computed from it here (the index arithmetic happens in
                // advance()).  Note that hipcc's atomic optimizer still broadcasts the result with a v_readfirstlane right behind
                // the atomic, so the drawing wave waits for the round trip here; switching the optimizer off and deferring the
                // wait to the end of the step was measured (round 4, profiles/r04/dq_ab.txt): no gain.
#if defined(MF_DQ_EAGER) && MF_DQ_EAGER
                if (tid == 0) { // (A/B: round 3's eager use of the result -- the wave waits for the atomic right here)
                    fetched = atomicAdd(ctr + head * PITCH, 1);
                    asm volatile("" : "+v"(fetched));
                }
#else
                if (tid == 0) fetched = atomicAdd(ctr + head * PITCH, 1);
#endif
            } else {
                nn = pool_next++, --pool_left;
            }
        }
    }
    // end of an iteration (before the next iteration's barrier)
    __device__ __forceinline__ void advance(int tid) {
        if (MF_DYNQ && K != 0) {
            if (it == write_at && tid == 0) slot[sel] = index_of(fetched); // (the wait for the atomic's return is here)
            step = nxt, nxt = nn, ++it;
        } else {
            step = nxt, nxt += gridDim.x;
        }
    }
    __device__ __forceinline__ void finish(int tid) {
        if (MF_DYNQ && K != 0 && tid == 0) {
            // this workgroup's last (speculative, unconsumed) draw must have landed before its arrival is counted: otherwise it
            // could hit the counter after the last workgroup's reset and the next launch on this set would skip a chunk
            // (A returning atomic that has returned has been performed, and the arrival below is issued after the wait: that is all
            // the order the reset needs.  The arrival itself stays RELAXED: an agent-scope acquire/release here writes back and
            // invalidates the XCD's L2 once per workgroup -- 512 times per launch, under the workgroups that are still running;
            // round 4 measured 0.208 vs 0.183 ms on a 0.2 ms launch, profiles/r04/chain_dq_ab.txt.)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (__hip_atomic_fetch_add(ctr + HEADS * PITCH, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
#pragma unroll
                for (int h = 0; h <= HEADS; ++h) ctr[h * PITCH] = 0;
            }
        }
    }
};
// Queue configuration of a launch: `est_us` = the launch's expected duration (its bytes over the rates the kernels reach).
unsigned long dq_next_launch(); // (k_generic.hip) one process-wide counter of queue launches, for MF_DQ_CFGS
static inline int dq_config(int nsteps, int grid, double est_us) {
    // tuning: MF_DQ_CFG = K | heads << 8 for every launch (0x100 = static striding).  With MF_DQ_TUNE set it is re-read per
    // launch, and MF_DQ_CFGS = "c0,c1,..." gives the k-th queue launch of the process configuration c[k % n] (0 = automatic)
    const Switches &sw = switches();
    int forced = sw.dq_cfg;
    if (sw.dq_tune) {
        const unsigned long launches = dq_next_launch();
        const Switches now = switches_parse(); // the tuning scripts change these between launches of one process
        forced = now.dq_cfg;
        const char *l = now.dq_cfgs.c_str();
        if (*l) {
            int n = 1;
            for (const char *q = l; *q; ++q) n += *q == ',';
            int k = (int)(launches % (unsigned long)n);
            const char *q = l;
            while (k > 0 && *q) k -= *q++ == ',';
            forced = (int)strtol(q, nullptr, 0);
        }
    }
    if (forced) return forced;
    const double t_step = est_us * grid / (nsteps > 0 ? nsteps : 1);   // one workgroup's step, us
    const int K = t_step >= 8.0 ? 1 : (t_step >= 4.0 ? 2 : 4);
    const double draws_per_us = nsteps / (K * (est_us > 1.0 ? est_us : 1.0));
    // Static prefix: a draw's returning atomic is waited for where it is issued (hipcc broadcasts the result at once), 600 - 1 200
    // cycles of one wave at the top of a step that the next barrier makes the whole workgroup's.  What the queue is for -- workgroups
    // of one CU drifting apart, the tail -- needs only the last part of a launch to be dealt dynamically: the first
    // MF_DQ_STATIC (default 0.6) of every workgroup's share is a stride walk.
    const double sfrac = sw.dq_static;
    int S0 = (int)(sfrac * (double)nsteps / (grid > 0 ? grid : 1));
    S0 = S0 < 2 ? 2 : (S0 > 32767 ? 32767 : S0);
    const int cfg = K | (draws_per_us > 60.0 ? 8 : 1) << 8 | S0 << 16;
    if (sw.dq_verbose) fprintf(stderr, "[microflow_amd] step queue: %d steps on %d workgroups, est %.0f us -> cfg 0x%x\n", nsteps, grid, est_us, cfg);
    return cfg;
}
// the counter set of this launch: the operator's ring (kernels.hpp DYNQ_RING sets of DynSteps::INTS zeroed ints), next slot.
// Every ring counts its OWN launches -- `launches` lives in host memory next to the ring, in the operator that owns it -- so a
// handle's slots are only ever used up by launches of that handle: DYNQ_RING launches of one handle may be in flight whatever else
// the process launches in between.  No process-wide lock or table on the launch path.
static inline int *dq_slot(int *ring, unsigned long *launches) {
    if (!ring) return nullptr;
    const unsigned long n = launches ? __atomic_fetch_add(launches, 1ul, __ATOMIC_RELAXED) : dq_next_launch();
    return ring + (size_t)(n % (unsigned long)DYNQ_RING) * DynSteps::INTS;
}
// expected duration of a depthwise / pair launch from its HBM bytes and requantised bytes (4.5 and 4.0 TB/s: what the kernels reach)
static inline double dq_est_us(double hbm_bytes, double requant_bytes) {
    const double a = hbm_bytes / 4.5e6, b = requant_bytes / 4.0e6;
    return a > b ? a : b;
}

// ---- helpers of the matrix-pipe depthwise kernels (k_fused_mm.hip, k_stage.hip) ----
constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
// tile swizzle: output bit i = input bit (nibble i of TS) - 1, nibble 0 = no bit
template <int TS> __device__ __forceinline__ constexpr int tile_swz(int x) {
    int r = 0;
    if constexpr ((TS & 0xf) != 0) r |= ((x >> ((TS & 0xf) - 1)) & 1);
    if constexpr (((TS >> 4) & 0xf) != 0) r |= ((x >> (((TS >> 4) & 0xf) - 1)) & 1) << 1;
    if constexpr (((TS >> 8) & 0xf) != 0) r |= ((x >> (((TS >> 8) & 0xf) - 1)) & 1) << 2;
    return r;
}

// ------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------
// Persistent-workgroup kernels launch exactly as many workgroups as are resident (LDS-,
// VGPR- or wave-limited, asked of the runtime once per kernel), so every workgroup walks the
// same number of steps and none queues behind a finished one.
// x / scale of the boundary quantisation (src/quantize.rs:16-18), correctly rounded, in 3 instructions instead of the
// ~10 of the IEEE division expansion: q0 = x * r, e = fma(-scale, q0, x) (the exact remainder), q = fma(e, r, q0), with
// r = 1 / scale rounded on the host (Markstein's correction step).  `fast` is set only after the host has VERIFIED,
// exhaustively over all 2^32 input bit patterns, that the quantised byte is the same as with the true division for
// this (scale, zero point, element type): ops.hip quant_div_verified / k_generic.hip verify_quant_div.  Infinities
// and NaNs (q not finite) take the true division.
__device__ __forceinline__ float quant_div(float x, float scale, float rcp, bool fast) {
    if (!fast) return __fdiv_rn(x, scale);
    const float q0 = __fmul_rn(x, rcp);
    float q = __fmaf_rn(__fmaf_rn(-scale, q0, x), rcp, q0);
    if (!(__builtin_fabsf(q) <= 3.0e38f)) q = __fdiv_rn(x, scale);
    return q;
}

// 4 x 4 transpose between four registers and the four 16-lane groups of a wave: on return register i of lane
// group g holds what register g of lane group i held (v_permlane32_swap: lanes 32..63 of the first <-> lanes 0..31 of
// the second operand; v_permlane16_swap: odd 16-lane rows of the first <-> even rows of the second).
__device__ __forceinline__ void lane_group_transpose4(uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t &r3) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 t;
    t = __builtin_amdgcn_permlane32_swap(r0, r2, false, false), r0 = t[0], r2 = t[1];
    t = __builtin_amdgcn_permlane32_swap(r1, r3, false, false), r1 = t[0], r3 = t[1];
    t = __builtin_amdgcn_permlane16_swap(r0, r1, false, false), r0 = t[0], r1 = t[1];
    t = __builtin_amdgcn_permlane16_swap(r2, r3, false, false), r2 = t[0], r3 = t[1];
}

// Function attributes and occupancy are per DEVICE, and one process may drive several GPUs, so
// each launcher keeps one slot per device (benign race: two threads may both prepare a slot).
struct LaunchState {
    static constexpr int MAX_DEV = 64;
    std::atomic<int> per_cu[MAX_DEV];
};
template <typename Kern> static int prepared(LaunchState &st, Kern kern, int threads, int lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LaunchState::MAX_DEV) dev = LaunchState::MAX_DEV - 1;
    int n = st.per_cu[dev].load(std::memory_order_relaxed);
    if (n > 0 && dev != LaunchState::MAX_DEV - 1) return n;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, (size_t)lds_bytes) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
    }
    st.per_cu[dev].store(n, std::memory_order_relaxed);
    return n;
}

static inline int grid_for(size_t total, int per_block = 256, int cap = 256 * 8) {
    size_t g = (total + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g < (size_t)cap ? g : (size_t)cap);
}

// ---- fast-path dispatch tables ------------------------------------------------
// the instances of a fast kernel: epilogue mode MG in {0, 1, 2} (requant_t above)  x  {i8, u8} element type ...
#define MF_DISPATCH4(magic, xr, FN, ARGS, ...)                         \
    do {                                                               \
        const int mg_ = (magic);                                       \
        if (xr) {                                                      \
            if (mg_ == 2) FN<__VA_ARGS__, 2, 0x80808080u> ARGS;        \
            else if (mg_) FN<__VA_ARGS__, 1, 0x80808080u> ARGS;        \
            else FN<__VA_ARGS__, 0, 0x80808080u> ARGS;                 \
        } else {                                                       \
            if (mg_ == 2) FN<__VA_ARGS__, 2, 0u> ARGS;                 \
            else if (mg_) FN<__VA_ARGS__, 1, 0u> ARGS;                 \
            else FN<__VA_ARGS__, 0, 0u> ARGS;                          \
        }                                                              \
    } while (0);
// ... plus mode 3 (the single-fma form; its code does not depend on the element type: one instance, XR4 = 0) for the kernels
// of the fused step and their layer-wise counterparts
#define MF_DISPATCH5(magic, xr, FN, ARGS, ...)                         \
    do {                                                               \
        if ((magic) == 3) FN<__VA_ARGS__, 3, 0u> ARGS;                 \
        else MF_DISPATCH4(magic, xr, FN, ARGS, __VA_ARGS__)            \
    } while (0);

} // namespace k
} // namespace mf
