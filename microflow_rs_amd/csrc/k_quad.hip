// k_quad.hip -- TWO consecutive DepthwiseConv2D 3x3 + Conv2D 1x1 pairs in one launch ("quad"): person_detect ops 1..4
// (48x48x8 stride 1 -> 16, then 48x48x16 stride 2 -> 24x24x32) and ops 5..8 (24x24x32 stride 1 -> 32, then stride 2 ->
// 12x12x64).  (src/ops/depthwise_conv_2d.rs:28-105, src/ops/conv_2d.rs:28-108; four reference operators per launch, each
// with the reference's own requantisation -- the three intermediate int8 tensors just never leave the CU.)
//
// Why.  The pair kernels alternate between two regimes: a stride-1 pair requantises as many bytes as it moves and is bounded
// by the VALU (valu_frac 0.77-0.81, HBM at 0.6), the stride-2 pair that follows moves twice what it requantises and is
// bounded by HBM (0.65-0.71 of peak, VALU at 0.45) -- and between the two the largest tensors of the network (36 864 and
// 18 432 bytes per image) are written to HBM and read back.  Fused, the second pair's input never crosses HBM: ops 1..4
// move 36 864 B per image instead of 110 592, ops 5..8 27 648 instead of 64 512, and the HBM-bound half disappears under
// the VALU-bound one.
//
// How.  Both pairs are dwpw_rr pairs (k_fused_mm.hip: depthwise taps on the matrix pipe, the depthwise result requantised
// in registers straight into the pointwise MFMA's B operand).  A step is one image:
//     phase A : staged input tile -> pair A -> its 4..16 output bytes per lane are written into tile B, the halo'd,
//               swizzled LDS tile pair B's tap loads expect (instead of to HBM)
//     barrier ; the next image's DMA into tile A is issued here and lands during phase B (one staging buffer suffices)
//     phase B : tile B -> pair B -> HBM
//     barrier (top of the next step: the DMA has landed, everyone is done with tile B)
// Column grids, row pads and swizzles are the table rows of the single pairs (kernels.hpp MF_DWRR_SHAPES); the workgroup
// size is a multiple of four waves (waves go to the SIMDs round-robin per workgroup: anything else loads them unevenly and
// every barrier waits for the doubled-up ones) over which pair A's unit grid divides; a pair whose grid does not divide runs
// on fewer waves (ACT_B).  The dynamic step queue of k_common.hpp deals the images.
// With STEM the network's first operator (one input channel -> 8, stride 2) is a third phase in front of pair A: ops 0..4 in
// one launch (see quad_rr below).
#include "k_common.hpp"

#include <algorithm>
#include <type_traits>

// Wave priority 1 while a wave issues the MFMAs of its depthwise taps: the matrix pipe is fed ahead of the other waves'
// requantisation VALU work, which fills the issue slots behind it (0: off).  Same box, back to back: ops 0..4 1.247 -> 1.226 ms,
// ops 5..8 0.710 -> 0.700; priority on the pointwise MFMAs as well, or 2 / 3 instead of 1: no better.
#ifndef MF_QUAD_PRIO
#define MF_QUAD_PRIO 1
#endif
#ifndef MF_Q13_X2A
#define MF_Q13_X2A 0 // (tuning: RrPhase X2 of ops 1..4's two pairs and of ops 5..8's)
#define MF_Q13_X2B 0
#define MF_Q57_X2A 3
#define MF_Q57_X2B 3
#endif

// hipcc orders every LDS *store* behind all of the wave's outstanding LDS-DMA loads (SIInsertWaitcnts cannot tell the two LDS ranges
// apart): in the stem instance the next image's DMA is issued right in front of phase A, whose first store into tile B then waited
// for the HBM round trip (scripts/asm_dma_waits.py lists such waits).  1: in the STEM instances pair A's tile-B stores are inline asm
// (the pass does not look into it); the kernel's own waits -- vmcnt(0) in front of the barrier that hands the staged image over,
// lgkmcnt(0) in front of every barrier -- are what orders them.  2: in every instance (measured on ops 5..8, where no DMA is in
// flight during phase A: 0.445 -> 0.475 ms, the asm statements hem the scheduler in).  Same box, ops 0..4: 0.814 -> 0.766 ms.
// Superseded by MF_DMA_ASM (k_common.hpp: the DMA itself is what the compiler no longer sees); kept as an A/B switch.
#ifndef MF_QUAD_ASM_LDS
#define MF_QUAD_ASM_LDS 0
#endif
#ifndef MF_QUAD_ASM_CLOBBER
#define MF_QUAD_ASM_CLOBBER 1 // (tuning: 0 = the asm stores carry no "memory" clobber)
#endif
// 1: the stem's per-lane operands are read from an LDS copy made once per launch instead of from device memory every step: a
// vector-memory load returns in order behind phase B's output stores, so the stem phase used to start with a wait for those
// (same box, ops 0..4: 0.814 -> 0.755 ms; with the asm stores 0.713).
#ifndef MF_QUAD_STEM_LDS
#define MF_QUAD_STEM_LDS 1
#endif
// 1 (instances without the stem): the wait at the top of a step leaves phase B's output stores in flight -- they were issued after
// the staging DMAs, and vmcnt retires in order -- instead of draining them (same box, ops 5..8: 0.441 -> 0.4265 ms).
#ifndef MF_QUAD_CNT_WAIT
#define MF_QUAD_CNT_WAIT 1
#endif
// 1: the depthwise taps of both pairs run as ONE structured-sparse v_smfmac_i32_16x16x128_i8 (eight of the nine 16-byte tap chunks)
// + one v_mfma_i32_16x16x32_i8 (the ninth) per 16-channel unit instead of three v_mfma_i32_16x16x64_i8: the sparse instruction takes
// the time of ONE dense one (scripts/ubench/mfma_rates.hip), and these launches' time is close to the SUM of their matrix-pipe and
// VALU time (the knock-out with two of the three dense MFMAs: -7.5 % / -8 %, profiles/r06/d2_mfma_knockouts.txt).  Operands: ops.hip
// build_dw_sp_weights; the instruction's operand map: scripts/ubench/smfmac_probe.hip / smfmac_check.hip.  Bit-exact (the whole -m gpu
// suite) -- and NOT faster: ops 0..4 0.76 against 0.725 ms dense, ops 5..8 0.434 against 0.432 (profiles/r06/k_smfmac_variants.txt;
// with an earlier chunk map 0.42 against 0.43).  What eats the saved MFMA: the dense -> sparse pair of one chain needs 8 idle states
// between its two instructions (a hazard hipcc 7.2 does not pad: see the comment at the MFMAs), a third tap load per unit, and two
// more address registers per 16-channel group.  Default 0: the three dense instructions.
#ifndef MF_RR_SPARSE
#define MF_RR_SPARSE 0
#endif
#ifndef MF_RR_SP_PAD
#define MF_RR_SP_PAD 7
#endif
#ifndef MF_Q57_DB
#define MF_Q57_DB 0
#endif
#ifndef MF_QUAD_KO
#define MF_QUAD_KO 0 // knock-out timing experiments (WRONG results, never shipped): 1 no depthwise requantisation, 2 no pointwise requantisation,
                     // 4 no HBM stores, 8 no staging after the first step, 16 no barriers inside a step, 32 one of the three tap-row LDS loads only,
                     // (f32 instance) 64 no quantisation pass, 128 the pass without its arithmetic; 256 two of the three tap-row MFMAs
#endif

namespace mf {
namespace k {
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS stores the compiler's wait-count pass does not see (MF_QUAD_ASM_LDS).  `p` points into the workgroup's LDS; OFF is a
// compile-time byte offset (< 65536: the instruction's offset field).
// (`off` must fold to a constant -- the unit loops are fully unrolled -- or the build fails at the "i" constraint)
__device__ __forceinline__ void lds_store_asm(uint8_t *p, int off, uint2 v) {
    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
    const uint32_t a = (uint32_t)(uintptr_t)(lds_u8_t *)p;
    const u32x2_ d = {v.x, v.y};
#if MF_QUAD_ASM_CLOBBER
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(d), "i"(off) : "memory");
#else
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(d), "i"(off));
#endif
}
__device__ __forceinline__ void lds_store_asm(uint8_t *p, int off, uint4 v) {
    const uint32_t a = (uint32_t)(uintptr_t)(lds_u8_t *)p;
    const u32x4 d = {v.x, v.y, v.z, v.w};
#if MF_QUAD_ASM_CLOBBER
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(a), "v"(d), "i"(off) : "memory");
#else
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(a), "v"(d), "i"(off));
#endif
}
// the barrier of a kernel that stores with lds_store_asm: the compiler does not count those stores, so the wait is spelled out
template <bool ASMST> __device__ __forceinline__ void quad_barrier() {
    if constexpr (ASMST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_sync();
}
// the bare instruction: __syncthreads() carries a workgroup-scope fence, which hipcc completes with vmcnt(0) while it believes an
// LDS-DMA may be outstanding -- that would drain the output stores a counted wait (MF_QUAD_CNT_WAIT) has just left in flight
__device__ __forceinline__ void quad_barrier_raw() {
    mf_jitter();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    mf_jitter();
}

template <int H_, int W_, int C_, int S_, int N_, int CG_, int CY_, int ORD_, int ROWPAD_, int TS_>
struct RrGeom {
    static constexpr int H = H_, W = W_, C = C_, S = S_, N = N_, CG = CG_, CY = CY_, ORD = ORD_, ROWPAD = ROWPAD_, TS = TS_;
    static constexpr bool PAIR = C == 8;
    static constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S, OWC = PAIR ? OW / 2 : OW;
    static constexpr int NQ = PAIR ? 1 : C / 16, CX = 16 / (CG * CY);
    static constexpr int LP = C < 16 ? 16 : C;
    static constexpr int ROWB = W * C, ROW = LP + ROWB + LP + ROWPAD, TILE = (H + 2) * ROW;
    static constexpr int IMG = H * ROWB, ROWCH = ROWB / 16, OPIX = OH * OW;
    static constexpr int NT = (PAIR ? 2 * N : N) / 16, LB = 4 * NT;
    static_assert(C == 8 || C == 16 || C == 32, "register-resident pairs: C <= 32");
    static_assert(CG * CY * CX == 16 && OH % CY == 0 && OWC % CX == 0, "column grid");
    static_assert(ROWB % 16 == 0 && ROWCH <= 64 && ROW % 16 == 0, "staging geometry");
    static_assert(!PAIR || (S == 1 && OW % 2 == 0), "pair columns");
    static_assert(LB == 8 || LB == 16, "one 8- or 16-byte store per lane");
};

// geometry of the tile a pair writes into (void = HBM: never looked at)
template <typename T> struct DstTile {
    static constexpr int TILE = T::TILE, ROW = T::ROW, C = T::C;
};
template <> struct DstTile<void> {
    static constexpr int TILE = 0, ROW = 0, C = 0;
};

// one pair of a quad: the per-lane constants (init) and the unit loop (run)
// NACT: the waves of the workgroup the unit grid is dealt over (waves >= NACT sit the phase out: a 3 x 3 unit grid does
// not divide over 4 waves, and 4 waves per workgroup is what fills the four SIMDs evenly)
// X2: requantise two dwords at a time with the interleaved packs of k_common.hpp (epi_pack4x2): bit 0 the two depthwise
// dwords of a 32-channel pair, bit 1 the pointwise tiles in pairs.  Fewer hazard no-ops and two independent chains per wave:
// ops 5..8 0.698 -> 0.654 ms; ops 0..4 (168 VGPRs with it: spills) 1.218 -> 1.260, so a quad chooses per pair.
template <typename Ge, int G, int NACT, int MG, uint32_t XR4, int X2 = 0>
struct RrPhase {
    static constexpr int NWAVE = NACT;
    static constexpr int UG = G / Ge::CG, UY = Ge::OH / Ge::CY, UX = Ge::OWC / Ge::CX;
    static constexpr int PSY = cgcd(UY, NWAVE), PSX = cgcd(UX, NWAVE / PSY), PSG = cgcd(UG, NWAVE / PSY / PSX);
    static_assert(G % Ge::CG == 0 && PSY * PSX * PSG == NWAVE, "the unit grid does not divide over the waves");
    static constexpr int NUG = UG / PSG, NUY = UY / PSY, NUX = UX / PSX, NU = NUG * NUY * NUX;
    static constexpr int T_UG = Ge::CG * Ge::TILE, T_UY = Ge::CY * Ge::S * Ge::ROW, T_UX = Ge::PAIR ? Ge::CX * 16 : Ge::CX * Ge::S * Ge::C;
    static constexpr int NQ = Ge::NQ, NT = Ge::NT;

    static constexpr bool SP = MF_RR_SPARSE == 1 || (MF_RR_SPARSE == 2 && !Ge::PAIR) || (MF_RR_SPARSE == 3 && Ge::PAIR); // (2 / 3: debug)
    int tbase[NQ];
    v4i Adw[NQ][3];
    // SP: the chunk of this lane group in half 0 / half 1 of the sparse instruction's B operand, its 8 bytes of the ninth chunk;
    // the stored sparse A, its index register, the ninth chunk's dense A
    int tb0[NQ], tb1[NQ], tb9[NQ];
    v4i As[NQ];
    int Ai[NQ];
    long A9[NQ];
    float4 dA[NQ], dS[NQ];
    int4 dK[NQ];
    long Apw[NT];
    float4 cA[NT], cS[NT];
    int4 cK[NT];
    int dst_lane;   // this lane's first output byte: in the output tensor of a step (HBM) or in the next pair's LDS tile
    int cg, wug;    // image column of this lane, image offset of this wave (ragged steps)
    float dlo, dhi, plo, phi;

    // NEXT = the geometry whose input tile this pair's output is written into (void: the output goes to HBM)
    template <typename NEXT>
    __device__ __forceinline__ void init(const DwPwArgs &p, int lane, int wave) {
        const int col = lane & 15, g = lane >> 4;
        int cy, cx;
        {
            constexpr int ORD = Ge::ORD, CGc = Ge::CG, CYc = Ge::CY, CXc = Ge::CX;
            constexpr int D0 = (ORD == 0 || ORD == 1) ? CGc : (ORD == 2 || ORD == 3) ? CYc : CXc;
            constexpr int D1 = (ORD == 2 || ORD == 4) ? CGc : (ORD == 0 || ORD == 5) ? CYc : CXc;
            const int i0 = col % D0, i1 = (col / D0) % D1, i2 = col / (D0 * D1);
            cg = (ORD == 0 || ORD == 1) ? i0 : (ORD == 2 || ORD == 4) ? i1 : i2;
            cy = (ORD == 2 || ORD == 3) ? i0 : (ORD == 0 || ORD == 5) ? i1 : i2;
            cx = (ORD == 4 || ORD == 5) ? i0 : (ORD == 1 || ORD == 3) ? i1 : i2;
        }
        const int wpy = wave % PSY, wpx = (wave / PSY) % PSX, wpg = wave / (PSY * PSX);
        wug = wpg * Ge::CG;
        const int wave_t = wpg * T_UG + wpy * T_UY + wpx * T_UX;
        // first byte of the 16-byte tap chunk (filter row ty, chunk column gch) of unit (0, 0, 0) in this lane's column
        auto chunk_addr = [&](int q, int ty, int gch) {
            if constexpr (Ge::PAIR) {
                return cg * Ge::TILE + (cy + ty) * Ge::ROW + Ge::LP + (2 * cx - 2) * 8 + gch * 16 + wave_t;
            } else {
                const int xl = cx * Ge::S + gch - 1;
                return cg * Ge::TILE + (cy * Ge::S + ty) * Ge::ROW + Ge::LP + xl * Ge::C + 16 * (q ^ tile_swz<Ge::TS>(xl)) + wave_t;
            }
        };
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            tbase[q] = chunk_addr(q, 0, g);
            // (ops.hip DW_SP_CHUNK: half 0 = chunks (0,0) (0,1) (0,2) (1,0) of lane groups 0..3, half 1 = (1,1) (1,2) (2,0) (2,1): neighbouring
            // lane groups read neighbouring chunk columns of one filter row wherever nine chunks allow it, like the dense form, whose
            // tile pitches were chosen for exactly that)
            tb0[q] = g < 3 ? chunk_addr(q, 0, g) : chunk_addr(q, 1, 0);
            tb1[q] = g < 2 ? chunk_addr(q, 1, g + 1) : chunk_addr(q, 2, g - 2);
            tb9[q] = chunk_addr(q, 2, 2) + 8 * (g & 1);
        }
        const int opar = Ge::PAIR ? (g >> 1) : 0;
        const int n0 = Ge::PAIR ? 8 * (g & 1) : (Ge::N / 4) * g;
        const int ox = (Ge::PAIR ? 2 * cx + opar : cx) + wpx * (Ge::PAIR ? 2 : 1) * Ge::CX; // output pixel of unit (0, 0, 0)
        const int oy = cy + wpy * Ge::CY;
        if constexpr (std::is_void<NEXT>::value) {
            dst_lane = ((cg + wug) * Ge::OPIX + oy * Ge::OW + ox) * Ge::N + n0;
        } else {
            static_assert(NEXT::H == Ge::OH && NEXT::W == Ge::OW && NEXT::C == Ge::N, "the next pair consumes this pair's output");
            // the unit steps must not touch the x bits the next tile's swizzle looks at
            static_assert(tile_swz<NEXT::TS>(((Ge::PAIR ? 2 : 1) * Ge::CX) * 255) == 0 || NEXT::TS == 0, "x step vs swizzle");
            const int chunk = n0 >> 4, within = n0 & 15;
            dst_lane = (cg + wug) * NEXT::TILE + (oy + 1) * NEXT::ROW + NEXT::LP + ox * NEXT::C +
                       16 * (NEXT::NQ > 1 ? (chunk ^ tile_swz<NEXT::TS>(ox)) : chunk) + within;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if constexpr (SP) {
                const v4i w0 = ((const v4i *)p.dw.wsp)[(q * 64 + lane) * 2], w1 = ((const v4i *)p.dw.wsp)[(q * 64 + lane) * 2 + 1];
                As[q] = w0, Ai[q] = w1[0];
                A9[q] = (long)(((unsigned long)(uint32_t)w1[2] << 32) | (unsigned long)(uint32_t)w1[1]);
            } else {
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) Adw[q][ty] = ((const v4i *)p.dw.wmm)[(q * 3 + ty) * 64 + lane];
            }
            const int ch4 = Ge::PAIR ? (g & 1) : 4 * q + g;
            dA[q] = ((const float4 *)p.dw.A)[ch4];
            dS[q] = ((const float4 *)p.dw.S)[ch4];
            dK[q] = magic4<MG>(((const int4 *)p.dw.Kc)[ch4]);
        }
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            Apw[m] = ((const long *)p.pw.wrr)[m * 64 + lane];
            cA[m] = *(const float4 *)(p.pw.A + n0 + 4 * m);
            cS[m] = *(const float4 *)(p.pw.S + n0 + 4 * m);
            cK[m] = magic4<MG>(*(const int4 *)(p.pw.Kc + n0 + 4 * m));
        }
        dlo = p.dw.lo_f, dhi = p.dw.hi_f, plo = p.pw.lo_f, phi = p.pw.hi_f;
    }

    // tb: this pair's staged input tile(s); dst: the step's output tensor in HBM, or the next pair's LDS tile
    // ASMST: stores into the next pair's tile as inline asm (MF_QUAD_ASM_LDS)
    template <typename NEXT, bool ASMST = false>
    __device__ __forceinline__ void run(const uint8_t *tb, uint8_t *dst, int gvalid) const {
        constexpr bool TO_LDS = !std::is_void<NEXT>::value;
        // strides of one unit step at the destination
        constexpr int XF = Ge::PAIR ? 2 : 1;
        constexpr int D_UG = TO_LDS ? Ge::CG * DstTile<NEXT>::TILE : Ge::CG * Ge::OPIX * Ge::N;
        constexpr int D_UY = TO_LDS ? Ge::CY * DstTile<NEXT>::ROW : Ge::CY * Ge::OW * Ge::N;
        constexpr int D_UX = TO_LDS ? XF * Ge::CX * DstTile<NEXT>::C : XF * Ge::CX * Ge::N;
        // units in flight (explicitly double-buffered tap rows): both pairs' operands are resident, so the two-channel-group pairs
        // (C = 32) keep one
#ifdef MF_QUAD_UB // (tuning: force the units in flight where the unit count allows it)
        constexpr int UB = NU % MF_QUAD_UB == 0 ? MF_QUAD_UB : 1;
#else
        // (three in flight -- pair B of ops 1..4 has three units per wave -- measured 3 % slower for the whole launch than one at a
        // time since a phase is one block: the scheduler overlaps the units as far as the registers go by itself)
        constexpr int UB = NQ > 1 ? 1 : (NU % 2 == 0 ? 2 : 1);
#endif
        auto coords = [](int iu, int &ug, int &uy, int &ux) constexpr {
            ug = (iu / (NUY * NUX)) * PSG, uy = ((iu / NUX) % NUY) * PSY, ux = (iu % NUX) * PSX;
        };
        auto toff_of = [&](int iu) constexpr {
            int ug = 0, uy = 0, ux = 0;
            coords(iu, ug, uy, ux);
            return ug * T_UG + uy * T_UY + ux * T_UX;
        };
        v4i bq[UB][NQ][3], bn[UB][NQ][3]; // (SP: [0] / [1] = the lane's two chunks of the sparse instruction's B operand; [2] unused)
        long b9[UB][NQ], n9[UB][NQ];       // (SP) the lane's 8 bytes of the ninth chunk
        auto load_taps = [&](v4i (&b)[3], long &e, int q, int toff) {
            if constexpr (SP) {
                b[0] = *(const v4i *)(tb + tb0[q] + toff), b[1] = *(const v4i *)(tb + tb1[q] + toff);
                e = *(const long *)(tb + tb9[q] + toff);
            } else {
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) b[ty] = ((MF_QUAD_KO & 32) && ty > 0) ? b[0] : *(const v4i *)(tb + tbase[q] + toff + ty * Ge::ROW);
            }
        };
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int q = 0; q < NQ; ++q) load_taps(bq[u][q], b9[u][q], q, toff_of(u));
#pragma unroll
        for (int t0 = 0; t0 < NU; t0 += UB) {
            v4i acc[UB][NQ];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[u][q] = v4i{dK[q].x, dK[q].y, dK[q].z, dK[q].w};
#if MF_QUAD_PRIO
            __builtin_amdgcn_s_setprio(MF_QUAD_PRIO);
#endif
            if constexpr (SP) {
                typedef int v8i_ __attribute__((ext_vector_type(8)));
                // the ninth chunk first: a dense MFMA takes the start values as its srcC, the sparse one accumulates in place
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[u][q] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A9[q], b9[u][q], acc[u][q], 0, 0, 0);
                // HAZARD (gfx950, hipcc 7.2): a v_smfmac whose accumulator is the result of the v_mfma issued right in front of it reads
                // it too early -- the hardware forwards an accumulator between dense MFMAs, not from a dense to a sparse one, and hipcc
                // pads nothing here.  Found as wrong, run-to-run varying results with one unit per batch; 4 idle states between the two
                // are not enough, 8 are (profiles/r06/smfmac_hazard.txt).  The next units' tap loads go here -- they were due anyway --
                // and idle states make it >= 12 where no other chain's MFMA separates the pair.
                if (t0 + UB < NU) {
#pragma unroll
                    for (int u = 0; u < UB; ++u)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) load_taps(bn[u][q], n9[u][q], q, toff_of(t0 + UB + u));
                }
                // (the idle states hang on the accumulator itself -- a data dependence between the two instructions of a chain -- so the
                // scheduler stays free to interleave everything else; a sched_barrier here cost the five-operator launch 6 %)
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
#if MF_RR_SP_PAD == 11
                        asm volatile("s_nop 11" : "+v"(acc[u][q]));
#else
                        asm volatile("s_nop 7" : "+v"(acc[u][q]));
#endif
                    }
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const v4i &lo = bq[u][q][0], &hi = bq[u][q][1];
                        const v8i_ B = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        acc[u][q] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(As[q], B, acc[u][q], Ai[q], 0, 0);
                    }
            } else {
#pragma unroll
                for (int ty = 0; ty < ((MF_QUAD_KO & 256) ? 2 : 3); ++ty) // (knock-out 256: two of the three tap-row MFMAs)
#pragma unroll
                    for (int u = 0; u < UB; ++u)
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
                            acc[u][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Adw[q][ty], bq[u][q][ty], acc[u][q], 0, 0, 0);
            }
#if MF_QUAD_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            if (!SP && t0 + UB < NU) {
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) load_taps(bn[u][q], n9[u][q], q, toff_of(t0 + UB + u));
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                uint32_t d[2] = {0u, 0u};
                if constexpr ((MF_QUAD_KO & 1) != 0) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) d[q] = (uint32_t)(acc[u][q][0] ^ acc[u][q][1] ^ acc[u][q][2] ^ acc[u][q][3]);
                } else if constexpr (NQ == 2 && (X2 & 1) != 0) {
                    requant_pack4x2<MG, XR4>(acc[u][0], dA[0], dS[0], acc[u][1], dA[1], dS[1], dlo, dhi, d[0], d[1]);
                } else {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        d[q] = requant_pack4<MG, XR4>(acc[u][q][0], acc[u][q][1], acc[u][q][2], acc[u][q][3], dA[q], dS[q], dlo, dhi);
                }
                const long bop = (long)(((unsigned long)d[1] << 32) | (unsigned long)d[0]);
                uint32_t packed[NT];
                if constexpr ((MF_QUAD_KO & 2) != 0) {
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        v4i pa = {cK[m].x, cK[m].y, cK[m].z, cK[m].w};
                        pa = __builtin_amdgcn_mfma_i32_16x16x32_i8(Apw[m], bop, pa, 0, 0, 0);
                        packed[m] = (uint32_t)(pa[0] ^ pa[1] ^ pa[2] ^ pa[3]);
                    }
                } else if constexpr ((X2 & 2) != 0) {
                    static_assert((X2 & 2) == 0 || NT % 2 == 0, "pointwise tiles in pairs");
#pragma unroll
                    for (int m = 0; m < NT; m += 2) { // two tiles at a time
                        v4i pa = {cK[m].x, cK[m].y, cK[m].z, cK[m].w}, pb2 = {cK[m + 1].x, cK[m + 1].y, cK[m + 1].z, cK[m + 1].w};
                        pa = __builtin_amdgcn_mfma_i32_16x16x32_i8(Apw[m], bop, pa, 0, 0, 0);
                        pb2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(Apw[m + 1], bop, pb2, 0, 0, 0);
                        requant_pack4x2<MG, XR4>(pa, cA[m], cS[m], pb2, cA[m + 1], cS[m + 1], plo, phi, packed[m], packed[m + 1]);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        v4i pa = {cK[m].x, cK[m].y, cK[m].z, cK[m].w};
                        pa = __builtin_amdgcn_mfma_i32_16x16x32_i8(Apw[m], bop, pa, 0, 0, 0);
                        packed[m] = requant_pack4<MG, XR4>(pa[0], pa[1], pa[2], pa[3], cA[m], cS[m], plo, phi);
                    }
                }
                int ug = 0, uy = 0, ux = 0;
                coords(t0 + u, ug, uy, ux);
                const int doff = ug * D_UG + uy * D_UY + ux * D_UX;
                // (one image per step: nothing to test, and no branch between the units -- the scheduler then works across them)
                if (G == 1 || cg + wug + ug * Ge::CG < gvalid) { // a ragged last step stages fewer than G images
                    if constexpr (TO_LDS && ASMST) {
                        static_assert(G * DstTile<NEXT>::TILE < 65536, "tile offsets fit the ds offset field");
                        if constexpr (Ge::LB == 8) lds_store_asm(dst + dst_lane, doff, make_uint2(packed[0], packed[1]));
                        else lds_store_asm(dst + dst_lane, doff, make_uint4(packed[0], packed[1], packed[2], packed[3]));
                    } else if constexpr (TO_LDS) {
                        if constexpr (Ge::LB == 8) *(uint2 *)(dst + dst_lane + doff) = make_uint2(packed[0], packed[1]);
                        else *(uint4 *)(dst + dst_lane + doff) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
                    } else if (!(MF_QUAD_KO & 4) || packed[0] == 0x12345678u) {
                        if constexpr (Ge::LB == 8) st_out(dst + dst_lane + doff, make_uint2(packed[0], packed[1]));
                        else st_out(dst + dst_lane + doff, make_uint4(packed[0], packed[1], packed[2], packed[3]));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty) bq[u][q][ty] = bn[u][q][ty];
                    b9[u][q] = n9[u][q];
                }
        }
    }
};


// the two quads of person_detect: pair geometries = the MF_DWRR_SHAPES rows of the single pairs
#ifndef MF_Q13B_ROWPAD
#define MF_Q13B_ROWPAD 32 // (tuning: row pitch padding of tile B in ops 1..4 -- pair A's 8-byte stores into it vs pair B's tap reads)
#endif
#ifndef MF_Q13A_ROWPAD
#define MF_Q13A_ROWPAD 32
#endif
struct Quad13 {
    using A = RrGeom<48, 48, 8, 1, 16, 1, 4, 0, MF_Q13A_ROWPAD, 0x000>;
    using B = RrGeom<48, 48, 16, 2, 32, 1, 2, 0, MF_Q13B_ROWPAD, 0x000>;
    static constexpr int G = 1, NTHR = 768, WPE = 3, ACT_A = 12, ACT_B = 12, X2A = MF_Q13_X2A, X2B = MF_Q13_X2B;
    static constexpr bool DBA = false; // one staging buffer (the stem instance stages the stem tile instead)
    static constexpr const char *name = "quad_rr<48,48,8,1,16|48,48,16,2,32>";
};
// (measured alternatives for ops 1..4: two 6-wave workgroups per CU -- 1.51 ms, the waves land 4/4/2/2 on the SIMDs; two
// images per step = half the barriers per image -- 1.087 vs 1.086 ms: the barriers are not what bounds this kernel)
struct Quad57 {
    using A = RrGeom<24, 24, 32, 1, 32, 1, 2, 0, 0, 0x002>;
    using B = RrGeom<24, 24, 32, 2, 64, 1, 4, 0, 32, 0x002>;
    // 4 waves = one per SIMD, two workgroups per CU (248 VGPRs); pair B's 3 x 3 unit grid runs on three of them.
    // (3-wave workgroups load the SIMDs 2/2/1/1 and every barrier waits for the doubled-up ones: 0.85 ms vs 0.71)
    static constexpr int G = 1, NTHR = 256, WPE = 2, ACT_A = 4, ACT_B = 3, X2A = MF_Q57_X2A, X2B = MF_Q57_X2B;
    // MF_Q57_DB = 1: two staging buffers, the next image's DMA issued at the TOP of a step so that it has the whole step to land
    // instead of phase B only.  Measured, same box: 0.436 against 0.4285 ms with one buffer (the loop is unrolled twice for the two
    // buffers; what the knock-out "no staging after the first step" gains -- 0.05 ms -- is the DMA's own cost, not its latency).
    static constexpr bool DBA = MF_Q57_DB != 0;
    static constexpr const char *name = "quad_rr<24,24,32,1,32|24,24,32,2,64>";
};



// STEM: the network's first operator in front of pair A (person_detect op 0: DepthwiseConv2D 3x3 stride 2 with one input channel
// and 8 outputs, src/ops/depthwise_conv_2d.rs:28-105 with Cin = 1) as a third phase, dw3x3_stem8_mm's arithmetic (k_depthwise.hip):
// the 96x96 image is DMA-staged into its own tile; a tile of the stem = 16 pixel pairs x (2 pixels x 8 channels) is ONE
// v_mfma_i32_16x16x32_i8 whose K bytes are two aligned dwords of an image row; the lane's packed dword goes straight into tile A.
// The stem's output tensor (18 432 B per image, the input of pair A) then never crosses HBM either.  Its per-lane operands are
// re-read from device memory every step (L2 hits) instead of living in registers next to those of the two pairs.
template <typename Q, bool STEM> constexpr int quad_stem_bytes() {
    return STEM ? 16 + (2 * Q::A::H + 2) * (2 * Q::A::W) : 0;
}
// F32IN (STEM only): the model's f32 entry (M::predict, microflow-macros/src/lib.rs:186-190: Tensor::quantize of the input, then
// predict_inner).  The f32 image is DMA-staged into its own 36 KB buffer; between phase B and the stem phase every wave quantises
// the nine image rows its own stem tiles read (its eight + the row above, which the wave above also writes, with the same bytes:
// no barrier) into the stem tile, with dw3x3_stem8_mm's arithmetic (src/quantize.rs:16-22; the verified 3-instruction division of
// k_common.hpp quant_div) -- in round-to-nearest: the single-fma epilogue's round-toward-zero is switched off around these lines.
template <typename Q, bool STEM, int MG, uint32_t XR4, bool F32IN = false>
__global__ __launch_bounds__(Q::NTHR, Q::WPE) void quad_rr(const int8_t *__restrict__ in, int8_t *__restrict__ out, QuadArgs p, int batch) {
    static_assert(!F32IN || STEM, "the f32 entry is the stem's");
    using GA = typename Q::A;
    using GB = typename Q::B;
    epi_enter<MG>();
    constexpr int G = Q::G, NTHR = Q::NTHR, NWAVE = NTHR / 64;
    constexpr bool DBA = Q::DBA && !STEM; // two tile-A buffers, the next image staged a whole step ahead
    constexpr int BUF_A = G * GA::TILE, OFF_B = (DBA ? 2 : 1) * BUF_A + 512, BUF_B = G * GB::TILE, OFF_S = OFF_B + BUF_B + 512;
    constexpr int S_GUARD = 16, SW = 2 * GA::W, SH = 2 * GA::H, S_TILE = quad_stem_bytes<Q, STEM>(), OFF_Q = OFF_S + S_TILE;
    constexpr int OFF_F = OFF_Q + 16; // (F32IN) the staged f32 image, verbatim: SH rows of SW floats
    constexpr int OFF_C = OFF_F + (F32IN ? 4 * SH * SW : 0); // (STEM, MF_QUAD_STEM_LDS) the stem's operands: QuadArgs::stem's 152 dwords
    // (the f32 instance and the two-rounding epilogue forms sit on the 168-register step, where the copy's LDS pointer costs spills: they
    // keep the device-memory reads unless MF_QUAD_STEM_LDS is 2)
    constexpr bool ASMST = MF_QUAD_ASM_LDS == 2 || (MF_QUAD_ASM_LDS == 1 && STEM);
    constexpr bool STEM_LDS = STEM && MF_QUAD_STEM_LDS != 0 && ((!F32IN && MG == 3) || MF_QUAD_STEM_LDS > 1);
    static_assert(!STEM || (G == 1 && GA::C == 8 && GA::W == 48 && (GA::H / 2) % NWAVE == 0 && (SH * SW) % 1024 == 0 && S_TILE % 16 == 0),
                  "stem phase: 8 channels, 24 pixel pairs per output row, whole row pairs per wave");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DynSteps dq;
    dq.init(lds + OFF_Q, p.a.dw.queue, tid, p.a.dw.qcfg);
    if constexpr (STEM) // the stem tile's guard and padding rows hold the stem's input zero point
        for (int i = tid; i < S_TILE / 16; i += NTHR) ((uint4 *)(lds + OFF_S))[i] = make_uint4(p.stem_izp4, p.stem_izp4, p.stem_izp4, p.stem_izp4);
    if constexpr (STEM_LDS) // the stem's operands, the accumulator start values with the epilogue form's offset folded in
        if (tid < 152) ((uint32_t *)(lds + OFF_C))[tid] = p.stem[tid] + ((MG != 0 && tid >= 144) ? (uint32_t)MF_MAGIC_I : 0u);
    // tile A's halo holds pair A's input zero point, tile B's pair B's (= the zero point of pair A's output tensor)
    for (int i = tid; i < OFF_B / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.a.dw.izp4, p.a.dw.izp4, p.a.dw.izp4, p.a.dw.izp4);
    for (int i = tid; i < (BUF_B + 512) / 16; i += NTHR) ((uint4 *)(lds + OFF_B))[i] = make_uint4(p.b.dw.izp4, p.b.dw.izp4, p.b.dw.izp4, p.b.dw.izp4);
    RrPhase<GA, G, Q::ACT_A, MG, XR4, Q::X2A> pa;
    RrPhase<GB, G, Q::ACT_B, MG, XR4, Q::X2B> pb;
    pa.template init<GB>(p.a, lane, wave);
    pb.template init<void>(p.b, lane, wave);
    wg_sync(); // halo fills complete before any DMA lands

    auto stage = [&](int st, int buf = 0) {
        if constexpr (F32IN) { // the 96 x 96 f32 image, verbatim, in 1 KiB pieces
            constexpr int NI = SH * SW * 4 / 1024;
            static_assert(NI % NWAVE == 0, "f32 staging: whole pieces per wave");
#pragma unroll
            for (int k = 0; k < NI / NWAVE; ++k) {
                const int r = k * NWAVE + wave;
                dma16(in + ((size_t)st * (SH * SW * 4) + r * 1024 + lane * 16), lds + OFF_F + r * 1024);
            }
            return;
        }
        if constexpr (STEM) { // the 96 x 96 x 1 image, verbatim, in 1 KiB pieces behind the guard and the padding row
            constexpr int NI = SH * SW / 1024;
#pragma unroll
            for (int k = 0; k < (NI + NWAVE - 1) / NWAVE; ++k) {
                const int r = k * NWAVE + wave;
                if (r < NI) dma16(in + ((size_t)st * (SH * SW) + r * 1024 + lane * 16), lds + OFF_S + S_GUARD + SW + r * 1024);
            }
            return;
        }
        constexpr int NROWS = G * GA::H;
        const int src_lane = GA::NQ > 1 ? (lane ^ tile_swz<GA::TS>(lane / (GA::NQ > 1 ? GA::NQ : 1))) : lane;
#pragma unroll
        for (int k = 0; k < (NROWS + NWAVE - 1) / NWAVE; ++k) {
            const int r = k * NWAVE + wave;
            const int gi = r / GA::H, y = r % GA::H;
            if (r < NROWS && st * G + gi < batch && lane < GA::ROWCH)
                dma16(in + ((size_t)(st * G + gi) * GA::IMG + y * GA::ROWB + src_lane * 16), lds + buf * BUF_A + gi * GA::TILE + (y + 1) * GA::ROW + GA::LP);
        }
    };
    // (F32IN) boundary quantisation of the image rows this wave's stem tiles read: rows 4 RPW wave - 1 .. 4 RPW wave + 4 RPW - 1
    auto quant_rows = [&]() {
        if constexpr (F32IN && !(MF_QUAD_KO & 64)) { // (knock-out 64: no quantisation pass at all)
            constexpr int RPW = GA::H / 2 / NWAVE, NR = 4 * RPW + 1, C4 = SW / 4, NIT = (NR * C4 + 63) / 64;
            asm volatile("" ::: "memory");
            if constexpr (MG == 3) __builtin_amdgcn_s_setreg(0x801, 0); // MODE.FP_ROUND (f32) = nearest even, for these lines
            asm volatile("" ::: "memory");
            const int y0 = 4 * RPW * wave - 1;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = it * 64 + lane, row = item / C4, c4 = item - row * C4, y = y0 + row;
                if (item < NR * C4 && y >= 0) {
                    const f32x4 v = *(const f32x4 *)(lds + OFF_F + (y * SW + 4 * c4) * 4);
                    // Four pixels at a time.  The verified 3-instruction division (k_common.hpp quant_div) for all four; its quotient is
                    // finite for every input the verification let through, and then nothing below can produce a NaN -- so ONE test per
                    // four pixels (not a branch per pixel: those cut this pass into blocks and were a third of its time) decides whether
                    // the lane redoes its four with the general arithmetic (true division, NaN -> 0): non-finite inputs, or parameters
                    // the 3-instruction form was not verified for.
                    int qv[4];
                    float q[4];
                    bool plain = p.in_fast != 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float q0 = __fmul_rn(v[e], p.in_rcp);
                        q[e] = __fmaf_rn(__fmaf_rn(-p.in_scale, q0, v[e]), p.in_rcp, q0);
                        plain = plain && (__builtin_fabsf(q[e]) <= 3.0e38f);
                    }
                    if constexpr ((MF_QUAD_KO & 128) != 0) { // (knock-out: the staging without the arithmetic)
#pragma unroll
                        for (int e = 0; e < 4; ++e) qv[e] = __float_as_int(v[e]);
                    } else if (__builtin_expect(plain, 1)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = __fadd_rn(q[e], p.in_zp_f);
                            const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
                            qv[e] = (int)__builtin_amdgcn_fmed3f(r, p.in_sat_lo, p.in_sat_hi);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = __fadd_rn(__fdiv_rn(v[e], p.in_scale), p.in_zp_f);
                            const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
                            qv[e] = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.in_sat_lo, p.in_sat_hi);
                        }
                    }
                    *(uint32_t *)(lds + OFF_S + S_GUARD + (y + 1) * SW + 4 * c4) = pack4(qv[0], qv[1], qv[2], qv[3]) ^ p.in_xr4;
                }
            }
            asm volatile("" ::: "memory");
            if constexpr (MG == 3) __builtin_amdgcn_s_setreg(0x801, 3); // back to toward zero
            asm volatile("" ::: "memory");
        }
    };
    // stem phase: three tile types per row pair (pairs 0..15 of row r0 | 16..23 of r0 and 0..7 of r0 + 1 | 8..23 of r0 + 1)
    auto stem_phase = [&](const uint32_t *sc) {
        constexpr int RPW = GA::H / 2 / NWAVE;        // output row pairs per wave
        const int col = lane & 15, g = lane >> 4;
        // (the pointer is laundered every step, so the compiler no longer knows it is global: say so, or these become flat loads
        // that count on lgkmcnt as well and make every LDS wait of the phase wait for them)
        const int cq = (g & 1) * 4;                   // this lane's channels within its pixel
        long Aw;
        v4i cA_, cS_, cK;
        if constexpr (STEM_LDS) { // from the LDS copy (made before the first barrier): no vector-memory wait in this phase
            const uint32_t *lsc = (const uint32_t *)(lds + OFF_C);
            Aw = *(const long *)(lsc + 2 * lane);
            cA_ = *(const v4i *)(lsc + 128 + cq), cS_ = *(const v4i *)(lsc + 136 + cq), cK = *(const v4i *)(lsc + 144 + cq);
        } else {
            typedef __attribute__((address_space(1))) const uint32_t g_u32;
            g_u32 *gsc = (g_u32 *)(uintptr_t)sc;
            typedef long i64x1 __attribute__((ext_vector_type(1)));
            Aw = (*(__attribute__((address_space(1))) const i64x1 *)(gsc + 2 * lane))[0];
            cA_ = *(__attribute__((address_space(1))) const v4i *)(gsc + 128 + cq), cS_ = *(__attribute__((address_space(1))) const v4i *)(gsc + 136 + cq);
            const v4i ck_ = *(__attribute__((address_space(1))) const v4i *)(gsc + 144 + cq);
            const int4 ck = magic4<MG>(make_int4(ck_[0], ck_[1], ck_[2], ck_[3]));
            cK = v4i{ck.x, ck.y, ck.z, ck.w};
        }
        const float4 cA = make_float4(__int_as_float(cA_[0]), __int_as_float(cA_[1]), __int_as_float(cA_[2]), __int_as_float(cA_[3]));
        const float4 cS = make_float4(__int_as_float(cS_[0]), __int_as_float(cS_[1]), __int_as_float(cS_[2]), __int_as_float(cS_[3]));
        const int oy1 = col >= 8 ? 1 : 0, j1 = col >= 8 ? col - 8 : 16 + col;
        // operand: the aligned dwords at columns 4j - 4 and 4j of tile row 2 oy + ky (ky = g; tile row 0 = input row -1)
        const int off0 = S_GUARD + g * SW + 4 * col - 4, off1 = S_GUARD + (2 * oy1 + g) * SW + 4 * j1 - 4, off2 = S_GUARD + (2 + g) * SW + 4 * (8 + col) - 4;
        // result: pixel 2j + (g >> 1) of output row oy -> tile A row oy + 1, 8 bytes per pixel, this lane's 4 channels
        const int px = (g >> 1) * 8 + cq;
        const int dst0 = GA::ROW + GA::LP + 16 * col + px, dst1 = (1 + oy1) * GA::ROW + GA::LP + 16 * j1 + px, dst2 = 2 * GA::ROW + GA::LP + 16 * (8 + col) + px;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rp = wave * RPW + i;
            const uint8_t *t0 = lds + OFF_S + rp * (4 * SW);
            uint8_t *d0 = lds + rp * (2 * GA::ROW);
            long B[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const uint32_t *q = (const uint32_t *)(t0 + (n == 0 ? off0 : (n == 1 ? off1 : off2)));
                uint32_t lo = q[0];
                const uint32_t hi = q[1];
                if (n == 0) lo = col == 0 ? p.stem_izp4 : lo;  // pair 0: column -1 is padding
                if (n == 1) lo = col == 8 ? p.stem_izp4 : lo;
                B[n] = (long)(((unsigned long)hi << 32) | (unsigned long)lo);
            }
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const v4i acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(Aw, B[n], cK, 0, 0, 0);
                *(uint32_t *)(d0 + (n == 0 ? dst0 : (n == 1 ? dst1 : dst2))) =
                    requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], cA, cS, p.stem_lo, p.stem_hi);
            }
        }
    };
    const int nsteps = (batch + G - 1) / G;
    if (dq.step < nsteps) stage(dq.step);
    int ko_steps = 0; // (steps done: knock-out switch 8 acts from the second step on)
    const uint32_t *sc = p.stem;
    if constexpr (STEM) {
        // Two barriers per image: the stem phase of the NEXT image follows phase B of this one without a barrier between them
        // (it reads the stem tile, landed by barrier Y, and writes tile A, free since barrier Y).
        //     X : tile A complete (stem phase), tile B and the stem tile free  -> DMA of the next image, phase A
        //     Y : tile B complete, tile A free, the next image landed          -> phase B, then the next image's stem phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();
        if (dq.step < nsteps) quant_rows(), stem_phase(sc);
        for (; dq.step < nsteps; dq.advance(tid)) {
            const int step = dq.step;
            if (!(MF_QUAD_KO & 16)) quad_barrier<ASMST>(); // X
            dq.top(tid);
            if (dq.nxt < nsteps && !((MF_QUAD_KO & 8) && ko_steps > 0)) stage(dq.nxt);
            if (Q::ACT_A == NWAVE || wave < Q::ACT_A) pa.template run<GB, ASMST>(lds, lds + OFF_B, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(MF_QUAD_KO & 16)) quad_barrier<ASMST>(); // Y
            if (Q::ACT_B == NWAVE || wave < Q::ACT_B)
                pb.template run<void>(lds + OFF_B, (uint8_t *)out + (size_t)step * GB::OPIX * GB::N, 1);
            asm volatile("" : "+s"(sc)); // (the stem's operands are fetched here, every step, not hoisted into registers)
            if (dq.nxt < nsteps) quant_rows(), stem_phase(sc);
            ++ko_steps;
        }
    } else {
        bool first = true;
        int cur = 0;
        for (; dq.step < nsteps; dq.advance(tid)) {
            const int step = dq.step;
            // this step's image must have landed.  Outstanding, oldest first: the staging DMAs (issued behind the barrier in the middle
            // of the previous step), then that step's NU output stores of phase B -- vmcnt retires in order, so the stores can stay in flight
            if (MF_QUAD_CNT_WAIT && G == 1 && !first && (Q::ACT_B == NWAVE || wave < Q::ACT_B)) {
                constexpr int NSTORE = decltype(pb)::NU;
                static_assert(NSTORE >= 0 && NSTORE < 16, "counted wait");
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            first = false;
            // this step's image is staged; every wave is done reading tile B (the previous step's phase B)
            if constexpr (MF_QUAD_CNT_WAIT != 0 && G == 1) quad_barrier_raw();
            else quad_barrier<ASMST>();
            dq.top(tid);
            const int gvalid = min(G, batch - step * G);
            if constexpr (DBA) { // the other buffer was last read in the previous step's phase A: refill it now, a whole step ahead
                if (dq.nxt < nsteps && !((MF_QUAD_KO & 8) && ko_steps > 0)) stage(dq.nxt, cur ^ 1);
            }
            if (Q::ACT_A == NWAVE || wave < Q::ACT_A) pa.template run<GB, ASMST>(lds + (DBA ? cur * BUF_A : 0), lds + OFF_B, gvalid); // pair A: tile A -> tile B
            if (!(MF_QUAD_KO & 16)) quad_barrier<ASMST>(); // tile B is complete; tile A is free
            if constexpr (!DBA) {
                if (dq.nxt < nsteps && !((MF_QUAD_KO & 8) && ko_steps > 0)) stage(dq.nxt); // lands during phase B
            }
            if (Q::ACT_B == NWAVE || wave < Q::ACT_B)
                pb.template run<void>(lds + OFF_B, (uint8_t *)out + (size_t)step * G * GB::OPIX * GB::N, gvalid); // pair B: tile B -> HBM
            cur ^= 1;
            ++ko_steps;
        }
    }
    dq.finish(tid);
}

template <typename Q, bool STEM, int MG, uint32_t XR4, bool F32IN = false>
static void launch_quad_t(const int8_t *in, int8_t *out, const QuadArgs &a, int batch, hipStream_t s) {
    constexpr int lds = ((Q::DBA && !STEM) ? 2 : 1) * Q::G * Q::A::TILE + 512 + Q::G * Q::B::TILE + 512 + quad_stem_bytes<Q, STEM>() + 16 +
                        (F32IN ? 4 * (2 * Q::A::H) * (2 * Q::A::W) : 0) + (STEM ? 640 : 0);
    static_assert(lds <= 163840, "quad tiles do not fit the LDS");
    static LaunchState st;
    const int per_cu = prepared(st, quad_rr<Q, STEM, MG, XR4, F32IN>, Q::NTHR, lds);
    const int nsteps = (batch + Q::G - 1) / Q::G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    QuadArgs b = a;
    using GA = typename Q::A;
    using GB = typename Q::B;
    const double hbm = (double)batch * ((STEM ? (F32IN ? 16 : 4) * GA::H * GA::W : GA::H * GA::W * GA::C) + GB::OPIX * GB::N);
    const double rq = (double)batch * ((STEM ? GA::H * GA::W * GA::C : 0) + GA::OPIX * (GA::C + GA::N) + GB::OPIX * (GB::C + GB::N));
    b.a.dw.qcfg = dq_config(nsteps, grid, dq_est_us(hbm, rq));
    b.a.dw.queue = dq_slot(b.a.dw.queue, b.a.dw.qlaunch);
    hipLaunchKernelGGL((quad_rr<Q, STEM, MG, XR4, F32IN>), dim3(grid), dim3(Q::NTHR), lds, s, in, out, b, batch);
}
template <typename Q> static bool quad_matches(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2) {
    using GA = typename Q::A;
    using GB = typename Q::B;
    return H == GA::H && W == GA::W && C == GA::C && S == GA::S && N == GA::N && H2 == GB::H && W2 == GB::W && C2 == GB::C && S2 == GB::S &&
           N2 == GB::N;
}
static int quad_mask() { // MF_QUADS: bit 0 = ops 1..4, bit 1 = ops 5..8, bit 2 = the stem in front of ops 1..4 (tuning)
    const int m = switches().quads;
    return m;
}
const char *quad_name(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2) {
    if ((quad_mask() & 1) && quad_matches<Quad13>(H, W, C, S, N, H2, W2, C2, S2, N2)) return Quad13::name;
    if ((quad_mask() & 2) && quad_matches<Quad57>(H, W, C, S, N, H2, W2, C2, S2, N2)) return Quad57::name;
    return nullptr;
}
const char *quad_stem_name(int SH, int SW, int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2) {
    if ((quad_mask() & 4) && SH == 2 * Quad13::A::H && SW == 2 * Quad13::A::W && quad_matches<Quad13>(H, W, C, S, N, H2, W2, C2, S2, N2))
        return "penta_rr<96,96,1,2,8|48,48,8,1,16|48,48,16,2,32>";
    return nullptr;
}
bool launch_quad(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2, const int8_t *in, int8_t *out, const QuadArgs &a,
                 int batch, hipStream_t s) {
    if (!a.a.dw.wmm || !a.a.pw.wrr || !a.b.dw.wmm || !a.b.pw.wrr) return false;
    if (MF_RR_SPARSE && (!a.a.dw.wsp || !a.b.dw.wsp)) return false; // (no 2:4-sparse form of these taps: the pairs run one by one)
    int mg = std::min(std::min(a.a.dw.magic, a.a.pw.magic), std::min(a.b.dw.magic, a.b.pw.magic));
    if (a.stem) mg = std::min(mg, a.stem_magic);
    if (mg == 0) return false; // (the quads exist for the bit-pattern epilogues only)
#define MF_QUAD_GO2(Q, ST)                                                                         \
    do {                                                                                           \
        if (mg == 3) {                                                                             \
            launch_quad_t<Q, ST, 3, 0u>(in, out, a, batch, s);                                     \
        } else if (a.b.pw.xr) {                                                                           \
            if (mg == 2) launch_quad_t<Q, ST, 2, 0x80808080u>(in, out, a, batch, s);               \
            else launch_quad_t<Q, ST, 1, 0x80808080u>(in, out, a, batch, s);                       \
        } else {                                                                                   \
            if (mg == 2) launch_quad_t<Q, ST, 2, 0u>(in, out, a, batch, s);                        \
            else launch_quad_t<Q, ST, 1, 0u>(in, out, a, batch, s);                                \
        }                                                                                          \
        return true;                                                                               \
    } while (0)
#define MF_QUAD_GO(Q) MF_QUAD_GO2(Q, false)
    if (quad_matches<Quad13>(H, W, C, S, N, H2, W2, C2, S2, N2)) {
        if (a.stem) MF_QUAD_GO2(Quad13, true);
        MF_QUAD_GO(Quad13);
    }
    if (quad_matches<Quad57>(H, W, C, S, N, H2, W2, C2, S2, N2)) {
        MF_QUAD_GO(Quad57);
    }
#undef MF_QUAD_GO
#undef MF_QUAD_GO2
    return false;
}
// the f32 entry of the stem instance: `in` holds batch x (2 H) x (2 W) floats (16-byte aligned), a.in_* the boundary quantisation
bool launch_quad_f32(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2, const float *in, int8_t *out, const QuadArgs &a,
                     int batch, hipStream_t s) {
    if (!a.stem || !a.f32_ok || !(quad_mask() & 4) || !quad_matches<Quad13>(H, W, C, S, N, H2, W2, C2, S2, N2)) return false;
    if (!a.a.dw.wmm || !a.a.pw.wrr || !a.b.dw.wmm || !a.b.pw.wrr) return false;
    if (MF_RR_SPARSE && (!a.a.dw.wsp || !a.b.dw.wsp)) return false;
    const int mg = std::min(std::min(std::min(a.a.dw.magic, a.a.pw.magic), std::min(a.b.dw.magic, a.b.pw.magic)), a.stem_magic);
    if (mg == 0) return false;
    const int8_t *src = (const int8_t *)in;
    if (mg == 3) launch_quad_t<Quad13, true, 3, 0u, true>(src, out, a, batch, s);
    else if (a.b.pw.xr) {
        if (mg == 2) launch_quad_t<Quad13, true, 2, 0x80808080u, true>(src, out, a, batch, s);
        else launch_quad_t<Quad13, true, 1, 0x80808080u, true>(src, out, a, batch, s);
    } else {
        if (mg == 2) launch_quad_t<Quad13, true, 2, 0u, true>(src, out, a, batch, s);
        else launch_quad_t<Quad13, true, 1, 0u, true>(src, out, a, batch, s);
    }
    return true;
}

} // namespace k
} // namespace mf
