// k_chain.hip -- RUN-TIME-GEOMETRY fused chains: 1 .. CHAIN_MAX consecutive DepthwiseConv2D 3x3 + Conv2D 1x1 pairs of ANY
// height / width (C % 16 == 0, N % 16 == 0) in ONE launch, every tensor between the chain's input and output in LDS.
//
// (src/ops/depthwise_conv_2d.rs:28-105 + src/ops/conv_2d.rs:28-108.  The reference is generic over every dimension
// (const generics, depthwise_conv_2d.rs:28-49, conv_2d.rs:28-49); the fused kernels of k_fused_mm.hip / k_quad.hip /
// k_stage.hip are instantiated for person_detect's rows only.  This kernel is their run-time-geometry form: the same
// arithmetic -- depthwise taps as three v_mfma_i32_16x16x64_i8 against block-diagonal weights, the reference's
// requantisation of the intermediate tensor, the 1x1 convolution as an MFMA product over the pixel matrix, the reference's
// requantisation again -- with every extent a kernel argument.  Each operator's result is the reference's int8 tensor; it
// just never leaves the CU.)
//
//   step      : G whole images per workgroup (8 waves); the dynamic step queue of k_common.hpp deals the steps.
//   staging   : the first pair's input rows by LDS-DMA into a halo'd tile (halo = that pair's input zero point, filled once);
//               the next step's images are issued as soon as the first depthwise phase has read the tile.
//   depthwise : unit = 16 MFMA columns (CG images x CY rows x CX columns, chosen by the host as divisors of the tensor)
//               x one 16-channel group; a lane's operand is ONE ds_read_b128 per filter row; the 16-byte group index inside
//               a pixel is XOR-swizzled with bits of the tile column (on the DMA source, on the tap reads and on the
//               pointwise writes of the previous pair), which makes 16 consecutive columns conflict-free for every C.
//               The unit list (channel group major) is cut into 8 contiguous ranges, one per wave; the walk is scalar.
//               Result -> requantise -> planar MID [16-channel group][pixel][16 B].
//   pointwise : item = 16 pixels x one block of TB <= 4 output tiles; B operand = KS ds_read_b128 of MID planes; operand A in
//               registers (loaded once for a single pair, per phase for a chain: L2 hits); a lane ends with 4 TB consecutive
//               output bytes of its pixel (row permutation of build_pw_rt_reg_weights) -> one store, to HBM or -- through a
//               pixel -> tile-offset table built once per launch -- into the next pair's halo'd, swizzled tile.
//   barriers  : depthwise | pointwise | next pair ...: two per pair.
#include "k_common.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

// tuning switches (scripts/variants.py)
#ifndef MF_CHAIN_TBM1
#define MF_CHAIN_TBM1 2   // output tiles per wave block with one k step (K <= 64): 2, or 4 (a lane then ends with 16 consecutive bytes, but the
                          // kernel sits on the 128-register limit and spills: its reloads are vector-memory loads whose wait also waits for the
                          // staging DMAs; round 4, profiles/r04/chain_ab.txt: 32x32x32 s2 -> 64 0.395 -> 0.336 ms, 64x64x16 s2 -> 32 0.757 -> 0.667 with 2)
#endif
#ifndef MF_CHAIN_RAW_BARRIER
#define MF_CHAIN_RAW_BARRIER 1 // 0: __syncthreads() between the phases of a step (A/B)
#endif
#ifndef MF_CHAIN_STAGE_FAST
#define MF_CHAIN_STAGE_FAST 1 // 0: every step's staging DMAs by the generic walk (A/B)
#endif
#ifndef MF_CHAIN_KEEP
#define MF_CHAIN_KEEP 1 // 0: the single pair's record may be re-materialised by scalar loads inside the step loop (A/B)
#endif
#ifndef MF_CHAIN_KO
#define MF_CHAIN_KO 0   // knock-out timing experiments (WRONG results, never shipped): 1 no depthwise requantisation, 2 no pointwise
                          // requantisation, 4 no HBM stores, 8 no staging after the first step, 16 no phase barriers, 32 no top-of-step drain,
                          // 64 a chain's per-phase operand loads only in the first step, 128 its pair records only in the first step
#endif
#ifndef MF_CHAIN_WPE
#define MF_CHAIN_WPE 4    // waves per SIMD the register allocation leaves room for (K <= 128)
#endif

#ifndef MF_CHAIN_DIAG
#define MF_CHAIN_DIAG 0   // 1: shader-clock stamps of workgroup 0 / wave 0 at the phase boundaries of its 3rd step (never shipped)
#endif

namespace mf {
namespace k {

#if MF_CHAIN_DIAG
__device__ long long g_chain_trace[128]; // [0..63] wave 0 (the wave that draws from the step queue), [64..127] wave 3
#define MF_CTR(k) do { if (blockIdx.x == 0 && (wave == 0 || wave == 3) && lane == 0 && (k) < 62) { if (trace_step == 2) g_chain_trace[(k) + (wave ? 64 : 0)] = (long long)__builtin_readcyclecounter(); if (trace_step == 3 && (k) == 0) g_chain_trace[62 + (wave ? 64 : 0)] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define MF_CTR(k) do { } while (0)
#endif

namespace {
struct CDwW {        // depthwise operands of one 16-channel group
    v4i A[3];
    float4 a, s;
    int4 k;
};
// output tiles per wave block: 4 (a lane ends with 16 consecutive bytes) while one k step's operands fit the register budget of four
// waves per SIMD, 2 beyond
template <int KSC> constexpr int chain_tbm() { return KSC == 1 ? MF_CHAIN_TBM1 : 2; }
template <int KSC> struct CPwW { // pointwise operands of one block of output tiles
    static constexpr int TBM = chain_tbm<KSC>();
    v4i A[TBM][KSC];
    float4 a[TBM], s[TBM];
    int4 k[TBM];
};
} // namespace

// NW: waves per workgroup.  8 by default (two workgroups per CU = four waves per SIMD); 16 where the LDS plan admits one workgroup per CU
// only (K <= 64).
// SOLO: the chain is ONE pair (the common case).  The pair record is then loop-invariant: its fields are read once and kept in
// scalar registers (for a chain the record pointer is laundered every step so that the per-phase operand addresses are formed
// where they are used -- and every phase entry pays a few dependent scalar-load latencies: stamps of round 4, MF_CHAIN_DIAG,
// showed 2 000 - 3 000 cycles between a step's top barrier and its first tap load).
// RES (with SOLO): every wave's depthwise units lie inside ONE channel group, so its operands are loaded once per launch and the
// depthwise loop carries no reload code -- hence no vmcnt wait inside a step.  That matters beyond the load itself: the step queue's
// returning atomic (issued at a step's top, consumed at its end) would otherwise be waited for by the first such wait, 1 - 3 us
// under load, by the one wave that drew -- and the phase barrier makes everyone wait for that wave (stamps: MF_CHAIN_DIAG).
template <int KSC, int NW, bool SOLO, bool RES, int MG, uint32_t XR4>
__global__ __launch_bounds__(NW * 64, KSC == 4 ? 2 : (NW == 8 ? MF_CHAIN_WPE : 4)) void chain_rt(const int8_t *__restrict__ in, int8_t *__restrict__ out, ChainArgs p, int batch) {
    constexpr int NTHR = NW * 64, NWAVE = NW, TBM = chain_tbm<KSC>();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(4))) const ChainPair c_pair;
    c_pair *pairs = (c_pair *)(uintptr_t)p.pairs;
    typedef __attribute__((address_space(1))) const v4i g_v4i;
    auto ld16 = [](const void *base, uint32_t off) { return *(g_v4i *)((uintptr_t)base + off); };
    auto ldf4 = [&](const void *base, uint32_t off) {
        const v4i v = ld16(base, off);
        return make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3]));
    };
    auto ldi4 = [&](const void *base, uint32_t off) {
        const v4i v = ld16(base, off);
        return magic4<MG>(make_int4(v[0], v[1], v[2], v[3]));
    };
    const int G = p.G, NP = SOLO ? 1 : p.npairs;
    const int col = lane & 15, g = lane >> 4;

    DynSteps dq;
    dq.init(lds + p.q_off, p.queue, tid, p.qcfg);
    // halo fills: every tile region holds the input zero point of the depthwise that reads it
#pragma unroll
    for (int f = 0; f < CHAIN_MAX; ++f) { // (constant indices: a run-time index into the argument struct would move it to scratch)
        if (f < p.nfill) {
            const uint32_t z = p.fill_izp4[f];
            uint4 *dst = (uint4 *)(lds + p.fill_off[f]);
            for (int i = tid; i < p.fill_bytes[f] / 16; i += NTHR) dst[i] = make_uint4(z, z, z, z);
        }
    }
    // pixel -> next-tile offset tables of the pairs whose output stays in LDS: offset of the pixel's 16-byte group 0 (24 bits)
    // | the group-index XOR of its tile column (8 bits)
    for (int pi = 0; pi + 1 < NP; ++pi) {
        c_pair &cp = pairs[pi];
        if (cp.otab_off < 0) continue;
        uint32_t *tab = (uint32_t *)(lds + cp.otab_off);
        const int OW = cp.OW, OPIX = cp.plimit_img;
        for (int pix = tid; pix < cp.P; pix += NTHR) {
            const int img = pix / OPIX, r = pix - img * OPIX, y = r / OW, x = r - y * OW;
            const uint32_t off = (uint32_t)(img * cp.dTILE + (y + 1) * cp.dROW + (x + 1) * cp.dC);
            const uint32_t sw = (uint32_t)(((x + 1) >> cp.dswz_sh) & cp.dswz_mask);
            tab[pix] = off | (sw << 24);
        }
    }

    // A workgroup barrier between two phases of a step: every wave's LDS writes of the phase are complete (lgkmcnt) and visible
    // behind it.  NOT __syncthreads(): its fence makes hipcc wait vmcnt(0) whenever an LDS-DMA is in flight (a DMA is a pending LDS
    // write on the VM counter: cdna_hip_programming.md, "Pipelining across barriers"), which would drain the next step's staging DMAs
    // -- and this step's HBM stores -- at every phase boundary instead of at the next step's top.
    auto phase_barrier = [&]() {
#if MF_CHAIN_KO & 16
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif MF_CHAIN_RAW_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#else
        wg_sync();
#endif
    };
    // A pair's record as the phases use it: plain registers.  SOLO: fetched ONCE before the step loop and laundered, so that the
    // register allocator keeps (or lane-spills) the values instead of re-materialising them -- it otherwise re-issued ~50 scalar
    // loads of the record per step, behind ~15 dependent lgkmcnt(0) waits (which also wait for the wave's LDS traffic), in front
    // of every phase.  A chain fetches the record of the pair it enters (one batch of loads, one wait).
    struct PR {
        int H, W, C, S, OH, OW, N, NQ, lgNQ, KS, TB, NBLK, ROW, TILE, tile_off, swz_sh, swz_mask, lgCX, lgCY, UG, UY, UX, P, NCH, PLANE,
            plimit_img, dst_off, dROW, dTILE, dC, dswz_sh, dswz_mask, otab_off;
        float dw_lo, dw_hi, pw_lo, pw_hi;
        int us0, us1, us2, ucnt;
        const int *rtab;
        const void *dw_wmm, *pw_w;
        const float *dwA, *dwS, *pwA, *pwS;
        const int *dwK, *pwK;
    };
    auto fetch_pair = [&](int pi) {
        c_pair &c = pairs[pi];
        PR r;
        r.H = c.H, r.W = c.W, r.C = c.C, r.S = c.S, r.OH = c.OH, r.OW = c.OW, r.N = c.N, r.NQ = c.NQ, r.lgNQ = c.lgNQ, r.KS = c.KS, r.TB = c.TB;
        r.NBLK = c.NBLK, r.ROW = c.ROW, r.TILE = c.TILE, r.tile_off = c.tile_off, r.swz_sh = c.swz_sh, r.swz_mask = c.swz_mask;
        r.lgCX = c.lgCX, r.lgCY = c.lgCY, r.UG = c.UG, r.UY = c.UY, r.UX = c.UX, r.P = c.P, r.NCH = c.NCH, r.PLANE = c.PLANE;
        r.plimit_img = c.plimit_img, r.dst_off = c.dst_off, r.dROW = c.dROW, r.dTILE = c.dTILE, r.dC = c.dC, r.dswz_sh = c.dswz_sh;
        r.dswz_mask = c.dswz_mask, r.otab_off = c.otab_off;
        r.dw_lo = c.dw_lo, r.dw_hi = c.dw_hi, r.pw_lo = c.pw_lo, r.pw_hi = c.pw_hi;
        r.us0 = c.ustart[wave][0], r.us1 = c.ustart[wave][1], r.us2 = c.ustart[wave][2], r.ucnt = c.ucount[wave];
        r.rtab = c.rtab, r.dw_wmm = c.dw_wmm, r.pw_w = c.pw_w, r.dwA = c.dwA, r.dwS = c.dwS, r.pwA = c.pwA, r.pwS = c.pwS, r.dwK = c.dwK, r.pwK = c.pwK;
        if constexpr (SOLO && MF_CHAIN_KEEP) {
#define MF_KEEP(x) asm volatile("" : "+s"(x))
            MF_KEEP(r.H); MF_KEEP(r.W); MF_KEEP(r.C); MF_KEEP(r.S); MF_KEEP(r.OH); MF_KEEP(r.OW); MF_KEEP(r.N); MF_KEEP(r.NQ); MF_KEEP(r.lgNQ);
            MF_KEEP(r.KS); MF_KEEP(r.TB); MF_KEEP(r.NBLK); MF_KEEP(r.ROW); MF_KEEP(r.TILE); MF_KEEP(r.tile_off); MF_KEEP(r.swz_sh);
            MF_KEEP(r.swz_mask); MF_KEEP(r.lgCX); MF_KEEP(r.lgCY); MF_KEEP(r.UG); MF_KEEP(r.UY); MF_KEEP(r.UX); MF_KEEP(r.P); MF_KEEP(r.NCH);
            MF_KEEP(r.PLANE); MF_KEEP(r.plimit_img); MF_KEEP(r.dst_off); MF_KEEP(r.otab_off);
            MF_KEEP(r.dw_lo); MF_KEEP(r.dw_hi); MF_KEEP(r.pw_lo); MF_KEEP(r.pw_hi);
            MF_KEEP(r.us0); MF_KEEP(r.us1); MF_KEEP(r.us2); MF_KEEP(r.ucnt); MF_KEEP(r.rtab);
#undef MF_KEEP
        }
        return r;
    };
    auto load_dw = [&](const PR &cp, int q) {
        CDwW w;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) w.A[ty] = ld16(cp.dw_wmm, (uint32_t)(((q * 3 + ty) * 64 + lane) * 16));
        const uint32_t co = (uint32_t)((4 * q + g) * 16);
        w.a = ldf4(cp.dwA, co), w.s = ldf4(cp.dwS, co), w.k = ldi4(cp.dwK, co);
        return w;
    };
    auto load_pw = [&](const PR &cp) {
        CPwW<KSC> w;
        const int TB = cp.TB, KS = cp.KS, blk = wave & (cp.NBLK - 1);
#pragma unroll
        for (int t = 0; t < TBM; ++t) {
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) {
                w.A[t][ks] = v4i{0, 0, 0, 0};
                if (t < TB && ks < KS) w.A[t][ks] = ld16(cp.pw_w, (uint32_t)((((blk * TB + t) * KS + ks) * 64 + lane) * 16));
            }
            const uint32_t co = (uint32_t)((blk * 16 * TB + g * 4 * TB + 4 * t) * 4);
            if (t < TB) w.a[t] = ldf4(cp.pwA, co), w.s[t] = ldf4(cp.pwS, co), w.k[t] = ldi4(cp.pwK, co);
            else w.a[t] = w.s[t] = make_float4(0.f, 0.f, 0.f, 0.f), w.k[t] = make_int4(0, 0, 0, 0);
        }
        return w;
    };

    // ---- staging of the chain's input (pair 0's tile) ----
    auto stage = [&](const PR &cp, int st, int buf) {
        const int H = cp.H, ROWB = cp.W * cp.C, ROWCH = ROWB >> 4, IMG = H * ROWB;
        const int lgNQ = cp.lgNQ, nqm = cp.NQ - 1, sh = cp.swz_sh, mask = cp.swz_mask;
        // rows of the step's G images are one contiguous run of the input (image after image): wave w takes rows w, w + NWAVE, ...
        const int nrows = (int)min((long)G, (long)batch - (long)st * G) * H;
        const int8_t *src0 = in + (long)st * G * IMG;
        uint8_t *tile0 = lds + cp.tile_off + buf * p.dbuf_stride + cp.C;
        int gi = 0, y = wave;
        while (y >= H) y -= H, ++gi;
        for (int r = wave; r < nrows; r += NWAVE) {
            const int8_t *src = src0 + (long)r * ROWB;
            uint8_t *dst = tile0 + gi * cp.TILE + (y + 1) * cp.ROW;
            for (int o = 0; o < ROWCH; o += 64) {
                const int i = o + lane; // 16-byte group i of the row lands at LDS group i; it must hold source group (x, c ^ swz(x))
                int sidx = i;
                if (mask != 0) {
                    const int x = i >> lgNQ, c = i & nqm;
                    sidx = (x << lgNQ) + (c ^ (((x + 1) >> sh) & mask));
                }
                if (i < ROWCH) dma16(src + sidx * 16, dst + o * 16);
            }
            y += NWAVE;
            while (y >= H) y -= H, ++gi;
        }
    };

    // The same DMAs from per-wave descriptors made once per launch: a full step of a wave is `nd` instructions whose lane source
    // offsets (relative to the step's first byte) and LDS destinations never change.  The generic walk above costs ~40 scalar and
    // vector instructions per DMA -- ~1 100 cycles per wave and step for two DMAs on an 8x8x128 tensor (stamps), an eighth of the step.
    constexpr int KD = 4;
    int sd_src[KD], sd_dst[KD], nd = 0;
    auto stage_plan = [&](const PR &cp) {
        const int H = cp.H, ROWB = cp.W * cp.C, ROWCH = ROWB >> 4;
        const int lgNQ = cp.lgNQ, nqm = cp.NQ - 1, sh = cp.swz_sh, mask = cp.swz_mask;
#pragma unroll
        for (int t = 0; t < KD; ++t) sd_src[t] = -1, sd_dst[t] = 0;
        int gi = 0, y = wave;
        while (y >= H) y -= H, ++gi;
        for (int r = wave; r < G * H; r += NWAVE) {
            for (int o = 0; o < ROWCH; o += 64) {
                const int i = o + lane;
                int sidx = i;
                if (mask != 0) {
                    const int x = i >> lgNQ, c = i & nqm;
                    sidx = (x << lgNQ) + (c ^ (((x + 1) >> sh) & mask));
                }
                const int so = i < ROWCH ? r * ROWB + sidx * 16 : -1, dd = cp.C + gi * cp.TILE + (y + 1) * cp.ROW + o * 16;
#pragma unroll
                for (int t = 0; t < KD; ++t)
                    if (t == nd) sd_src[t] = so, sd_dst[t] = dd;
                ++nd;
            }
            y += NWAVE;
            while (y >= H) y -= H, ++gi;
        }
    };
    auto stage_fast = [&](const PR &cp, int st, int buf) {
        const int8_t *src0 = in + (long)st * G * (cp.H * cp.W * cp.C);
        uint8_t *tile0 = lds + cp.tile_off + buf * p.dbuf_stride;
#pragma unroll
        for (int t = 0; t < KD; ++t)
            if (t < nd && sd_src[t] >= 0) dma16(src0 + sd_src[t], tile0 + __builtin_amdgcn_readfirstlane(sd_dst[t]));
    };

    // ---- depthwise phase of one pair: tile -> MID ----
    // A wave's units are a contiguous range of the list (channel group q, column ux, j = (image group, row)): it is walked in
    // segments of constant (q, ux) -- inside one, the lane's operand address is a per-segment VGPR plus a scalar offset from
    // the host's table, and the unit loop is one branch-free block.
    // RES (compile time): the wave's operands stay what they are for the whole launch -- no reload code in the loop, hence no
    // vmcnt wait there (a wait for an operand reload would also wait for the step-ahead staging DMAs: both count on vmcnt)
    auto dw_phase = [&](const PR &cp, CDwW &wd, int tile_base, auto resc) { // wd: the operands of this wave's first channel group, already on their way
        constexpr bool RESD = decltype(resc)::value;
        const int S = cp.S, C = cp.C, ROW = cp.ROW, sh = cp.swz_sh, mask = cp.swz_mask;
        const int lgCX = cp.lgCX, lgCY = cp.lgCY, CXv = 1 << lgCX, CYv = 1 << lgCY;
        const int gg = g < 2 ? g : 2; // tap column of this lane group (g == 3 meets zero weights: any readable bytes will do)
        const int cx = col & (CXv - 1), cy = (col >> lgCX) & (CYv - 1), cg = col >> (lgCX + lgCY);
        const int xin0 = cx * S + gg;
        const int tb0 = tile_base + cg * cp.TILE + cy * S * ROW + xin0 * C;
        const int mb0 = p.mid_off + (((cg * cp.OH + cy) * cp.OW + cx) << 4) + 4 * g;
        const int UGY = cp.UG * cp.UY, UX = cp.UX;
        const int T_UX = CXv * S * C, XSTEP = CXv * S, M_UX = CXv * 16, PLANE = cp.PLANE;
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(4))) const i32x2 c_int2;
        c_int2 *rt = (c_int2 *)(uintptr_t)cp.rtab;
        int q = cp.us0, ux = cp.us1, j = cp.us2;
        int n = cp.ucnt;
        const float lo = cp.dw_lo, hi = cp.dw_hi;
        while (n > 0) {
            const int seg = min(n, UGY - j);
            const int xin = xin0 + ux * XSTEP;
            const int a_seg = tb0 + ux * T_UX + ((q ^ ((xin >> sh) & mask)) << 4);
            const int m_seg = mb0 + q * PLANE + ux * M_UX;
            // unit offsets come from the host's table by scalar loads issued two units ahead (the wait that covers a unit's tap
            // loads then finds the entry of the unit after next already there)
            i32x2 e0 = rt[j], e1 = rt[j + (seg > 1 ? 1 : 0)];
            v4i t0 = *(const v4i *)(lds + a_seg + e0[0]), t1 = *(const v4i *)(lds + a_seg + e0[0] + ROW), t2 = *(const v4i *)(lds + a_seg + e0[0] + 2 * ROW);
            for (int k = 0; k < seg; ++k) {
                const i32x2 e2 = rt[j + (k + 2 < seg ? k + 2 : seg - 1)];
                v4i acc = {wd.k.x, wd.k.y, wd.k.z, wd.k.w};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[0], t0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[1], t1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[2], t2, acc, 0, 0, 0);
                // the next unit's taps (the last unit of a segment prefetches itself: no branch)
                t0 = *(const v4i *)(lds + a_seg + e1[0]), t1 = *(const v4i *)(lds + a_seg + e1[0] + ROW), t2 = *(const v4i *)(lds + a_seg + e1[0] + 2 * ROW);
#if MF_CHAIN_KO & 1
                *(uint32_t *)(lds + m_seg + e0[1]) = (uint32_t)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
#else
                *(uint32_t *)(lds + m_seg + e0[1]) = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], wd.a, wd.s, lo, hi);
#endif
                e0 = e1, e1 = e2;
            }
            n -= seg, j = 0;
            if (++ux == UX) {
                ux = 0, ++q;
                if constexpr (!RESD) {
                    if (n > 0) wd = load_dw(cp, q); // (a wave's range crosses into the next channel group)
                }
            }
        }
    };

    // ---- pointwise phase of one pair: MID -> the next pair's tile, or HBM ----
    // The chunk loop is instantiated per (tiles per block, k steps, destination): with run-time bounds every MFMA sat in its own
    // basic block behind a branch and the TB independent chains of a chunk could not be interleaved.
    auto pw_items = [&](const PR &cp, const CPwW<KSC> &wp, int step, int gvalid, auto tbc, auto ksc, auto dstc) {
        constexpr int TB = decltype(tbc)::value, KS = decltype(ksc)::value;
        constexpr bool TO_TILE = decltype(dstc)::value;
        const int NBLK = cp.NBLK, N = cp.N;
        const int lgB = NBLK == 1 ? 0 : (NBLK == 2 ? 1 : (NBLK == 4 ? 2 : 3));
        const int blk = wave & (NBLK - 1), slot = wave >> lgB, SLOTS = NWAVE >> lgB;
        const int P = cp.P, NCH = cp.NCH, PLANE = cp.PLANE, NQ = cp.NQ;
        const float lo = cp.pw_lo, hi = cp.pw_hi;
        int poff[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int pl = 4 * ks + g; // plane = 16-channel group; a k step hanging over K meets zero weights
            poff[ks] = p.mid_off + (pl < NQ ? pl : NQ - 1) * PLANE;
        }
        const int ch0 = blk * 16 * TB + g * 4 * TB;
        const int plim = gvalid * cp.plimit_img;
        int8_t *obase = out + (size_t)step * G * cp.plimit_img * N;
        const int voff0 = col * N + ch0;
        const uint32_t *otab = (const uint32_t *)(lds + (TO_TILE ? cp.otab_off : 0));
        const int chq = ch0 >> 4, chw = ch0 & 15, dst = cp.dst_off;
        constexpr bool PF = KSC <= 2; // operand prefetch of the next chunk while the registers allow it
        v4i B[KS], Bn[KS];
        auto fetch = [&](int px, v4i (&d)[KS]) {
            px = px < P ? px : P - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) d[ks] = *(const v4i *)(lds + poff[ks] + px * 16);
        };
        fetch(16 * slot + col, B);
        for (int c = slot; c < NCH; c += SLOTS) {
            const int pix = 16 * c + col;
            if constexpr (PF) fetch(pix + 16 * SLOTS, Bn); // the next chunk's operand (the last chunk re-reads pixel P - 1)
            else if (c != slot) fetch(pix, B);
            uint32_t tent = 0;
            if constexpr (TO_TILE) tent = otab[pix < P ? pix : P - 1];
            v4i acc[TB];
#pragma unroll
            for (int t = 0; t < TB; ++t) acc[t] = v4i{wp.k[t].x, wp.k[t].y, wp.k[t].z, wp.k[t].w};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < TB; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[t][ks], B[ks], acc[t], 0, 0, 0);
            uint32_t packed[TB];
#if MF_CHAIN_KO & 2
#pragma unroll
            for (int t = 0; t < TB; ++t) packed[t] = (uint32_t)(acc[t][0] ^ acc[t][1] ^ acc[t][2] ^ acc[t][3]);
            if constexpr (false) {
#else
            if constexpr (TB == 2) {
#endif // (with four tiles the interleaved packs cost more registers than four waves per SIMD leave)
#pragma unroll
                for (int t = 0; t < TB; t += 2)
                    requant_pack4x2<MG, XR4>(acc[t], wp.a[t], wp.s[t], acc[t + 1], wp.a[t + 1], wp.s[t + 1], lo, hi, packed[t], packed[t + 1]);
            } else {
#if !(MF_CHAIN_KO & 2)
#pragma unroll
                for (int t = 0; t < TB; ++t) packed[t] = requant_pack4<MG, XR4>(acc[t][0], acc[t][1], acc[t][2], acc[t][3], wp.a[t], wp.s[t], lo, hi);
#endif
            }
            if constexpr (TO_TILE) {
                if (pix < P) {
                    uint8_t *d = lds + dst + (tent & 0xffffffu) + ((chq ^ (int)(tent >> 24)) << 4) + chw;
                    if constexpr (TB == 4) *(uint4 *)d = make_uint4(packed[0], packed[1], packed[2], packed[3]);
                    else if constexpr (TB == 2) *(uint2 *)d = make_uint2(packed[0], packed[1]);
                    else *(uint32_t *)d = packed[0];
                }
            } else if (pix < plim && (!(MF_CHAIN_KO & 4) || packed[0] == 0x12345678u)) {
                int8_t *o = obase + voff0 + c * 16 * N;
                if constexpr (TB == 4) st_out(o, make_uint4(packed[0], packed[1], packed[2], packed[3]));
                else if constexpr (TB == 2) st_out(o, make_uint2(packed[0], packed[1]));
                else {
#pragma unroll
                    for (int t = 0; t < TB; ++t) st_out(o + 4 * t, packed[t]);
                }
            }
            if constexpr (PF) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) B[ks] = Bn[ks];
            }
        }
    };
    auto pw_phase = [&](const PR &cp, const CPwW<KSC> &wp, int step, int gvalid) {
        using std::integral_constant;
        const int TB = cp.TB, KS = cp.KS;
        const bool to_tile = cp.dst_off >= 0;
        bool done = false;
        auto go = [&](auto tbc, auto ksc) {
            if (done || TB != decltype(tbc)::value || KS != decltype(ksc)::value) return;
            done = true;
            if constexpr (decltype(tbc)::value != 3) {
                if (to_tile) {
                    pw_items(cp, wp, step, gvalid, tbc, ksc, integral_constant<bool, true>{});
                    return;
                }
            }
            pw_items(cp, wp, step, gvalid, tbc, ksc, integral_constant<bool, false>{});
        };
        auto go_tb = [&](auto ksc) {
            go(integral_constant<int, 1>{}, ksc);
            go(integral_constant<int, 2>{}, ksc);
            if constexpr (TBM == 4) {
                go(integral_constant<int, 3>{}, ksc);
                go(integral_constant<int, 4>{}, ksc);
            }
        };
        go_tb(integral_constant<int, 1>{});
        if constexpr (KSC >= 2) go_tb(integral_constant<int, 2>{});
        if constexpr (KSC >= 4) {
            go_tb(integral_constant<int, 3>{});
            go_tb(integral_constant<int, 4>{});
        }
    };

    wg_sync(); // halo fills and tables complete before any DMA lands
    const int nsteps = (batch + G - 1) / G;
    PR cp = fetch_pair(0);
    // (with one k step the kernel sits on the 128-register limit of four waves per SIMD: four more registers are spills there --
    // 32x32x32 s2 -> 64 0.40 -> 0.54 ms with the descriptors; profiles/r04/chain_ab.txt)
    constexpr bool FASTST = SOLO && (RES || KSC == 1) && (KSC >= 2 || MF_CHAIN_TBM1 <= 2) && MF_CHAIN_STAGE_FAST;
    // (the descriptors are a prologue: not for a launch of a few steps per workgroup -- 3x3x128 0.063 -> 0.067 ms with them)
    const bool fast_stage = FASTST && ((batch + G - 1) / G) > 24 * (int)gridDim.x;
    if constexpr (FASTST) {
        if (fast_stage) stage_plan(cp);
    }
    auto stage_any = [&](const PR &c0, int st, int buf) {
        if constexpr (FASTST) {
            if (fast_stage && nd <= KD && (st + 1) * G <= batch) { // (a full step; the ragged last one and large tensors take the generic walk)
                stage_fast(c0, st, buf);
                return;
            }
        }
        stage(c0, st, buf);
    };
    if (dq.step < nsteps) stage_any(cp, dq.step, 0);
    // a single pair keeps its operands in registers for the whole launch; a chain fetches them per phase (L2 hits): the
    // pointwise operands before the depthwise phase in front of them, the next depthwise's (first channel group) before
    // the pointwise phase in front of it -- so that both land under a phase of work
    const bool persist = NP == 1;
    CDwW wd = load_dw(cp, cp.us0);
    CPwW<KSC> wp;
    if (persist) wp = load_pw(cp);
    const bool dw_resident = RES; // (the launcher picks the RES instance when the plan says single_q)
    const bool dbuf = p.dbuf != 0;
    int cur = 0;
    int trace_ko = 0; // (steps done: the knock-out switches 64 / 128 act from the second step on)
#if MF_CHAIN_DIAG
    int trace_step = 0;
#endif
    for (; dq.step < nsteps; dq.advance(tid)) {
        const int step = dq.step;
        MF_CTR(0);
#if !(MF_CHAIN_KO & 32)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MF_CTR(1);
        wg_sync(); // this step's images are in pair 0's tile; every wave has left the previous step's last phase
        MF_CTR(2);
        dq.top(tid);
        if constexpr (!SOLO) asm volatile("" : "+s"(pairs));
        MF_CTR(60);
        if constexpr (!SOLO) cp = fetch_pair(0);
        if (dbuf && dq.nxt < nsteps && !(MF_CHAIN_KO & 8)) stage_any(cp, dq.nxt, cur ^ 1); // double buffered: the next step's images have this whole step to land
        MF_CTR(61);
        const int gvalid = min(G, batch - step * G);
        PR cp0 = cp; // (a chain stages pair 0's tile while it is inside a later pair)
        for (int pi = 0; pi < NP; ++pi) {
            if constexpr (!SOLO) {
                if (pi > 0 && !((MF_CHAIN_KO & 128) && trace_ko > 0)) cp = fetch_pair(pi);
            }
            if (!persist && !((MF_CHAIN_KO & 64) && trace_ko > 0)) wp = load_pw(cp); // lands during the depthwise phase
            MF_CTR(3 + 4 * pi);
            dw_phase(cp, wd, cp.tile_off + (pi == 0 ? cur * p.dbuf_stride : 0), std::integral_constant<bool, RES>{});
            MF_CTR(4 + 4 * pi);
            phase_barrier(); // MID complete; the tile has been read
            MF_CTR(5 + 4 * pi);
            if (!dbuf && pi == p.stage_after && dq.nxt < nsteps && !(MF_CHAIN_KO & 8)) stage_any(cp0, dq.nxt, 0); // pair 0's tile region is free: the next step's images fly under the rest of this step
            if (!dw_resident && !((MF_CHAIN_KO & 64) && trace_ko > 0)) {
                c_pair &nx = pairs[pi + 1 < NP ? pi + 1 : 0];
                PR nr;
                nr.dw_wmm = nx.dw_wmm, nr.dwA = nx.dwA, nr.dwS = nx.dwS, nr.dwK = nx.dwK;
                wd = load_dw(nr, nx.ustart[wave][0]); // lands during the pointwise phase
            }
            pw_phase(cp, wp, step, gvalid);
            MF_CTR(6 + 4 * pi);
            if (pi + 1 < NP) phase_barrier(); // the next pair's tile is complete; MID is free
        }
        if (dbuf) cur ^= 1;
        ++trace_ko;
#if MF_CHAIN_DIAG
        ++trace_step;
#endif
    }
    dq.finish(tid);
}

// ------------------------------------------------------------------------
// host: plan + launch
// ------------------------------------------------------------------------
static int lg2_exact(int v) {
    for (int i = 0; i < 31; ++i)
        if ((1 << i) == v) return i;
    return -1;
}
static int pow2_divisor(int v, int cap) { // largest power of two <= cap dividing v
    int d = 1;
    while (d * 2 <= cap && v % (d * 2) == 0) d *= 2;
    return d;
}

bool chain_plan(const ChainGeom *g, int n, ChainPair *pairs, ChainArgs &a, int lds_budget, int force_G, int force_dbuf) {
    if (n < 1 || n > CHAIN_MAX) return false;
    int maxCG = 1, KSC = 1;
    for (int i = 0; i < n; ++i) {
        const ChainGeom &s = g[i];
        ChainPair &c = pairs[i];
        if (s.C % 16 != 0 || s.C < 16 || s.C > 256 || s.N % 16 != 0 || s.N < 16 || s.N > 256) return false;
        if (s.S != 1 && s.S != 2) return false;
        if (s.OH != (s.H + s.S - 1) / s.S || s.OW != (s.W + s.S - 1) / s.S) return false;
        if (i + 1 < n && (g[i + 1].H != s.OH || g[i + 1].W != s.OW || g[i + 1].C != s.N)) return false;
        c.H = s.H, c.W = s.W, c.C = s.C, c.S = s.S, c.OH = s.OH, c.OW = s.OW, c.N = s.N;
        c.NQ = s.C / 16, c.lgNQ = lg2_exact(c.NQ), c.KS = (s.C + 63) / 64;
        KSC = std::max(KSC, c.KS <= 1 ? 1 : (c.KS == 2 ? 2 : 4));
        // tile: one halo pixel on every side; the 16-byte group index inside a pixel is XOR-ed with bits of the tile column so
        // that 16 consecutive columns hit 16 distinct 16-byte bank slots (NQ a power of two <= 16)
        c.swz_sh = 0, c.swz_mask = 0;
        if (c.lgNQ > 0) c.swz_sh = 4 - c.lgNQ, c.swz_mask = c.NQ - 1;
        // depthwise columns: divisors only (no overhang, no per-lane predicate)
        const int CX = pow2_divisor(s.OW, 16), CY = pow2_divisor(s.OH, 16 / CX), CG = 16 / (CX * CY);
        c.lgCX = lg2_exact(CX), c.lgCY = lg2_exact(CY);
        maxCG = std::max(maxCG, CG);
        // Row and image pitch: when the 16 columns of a unit span rows (CY > 1) or images (CG > 1), the pitches decide which
        // 16-byte bank slots the lanes of a ds_read_b128 service group hit.  Bank model (MI355X_MICROARCH.md, LDS): a wave's
        // b128 read is served in four groups of 16 lanes, one cycle per group plus one per extra distinct address on a busy
        // slot; the pads (multiples of 16 bytes) with the fewest modelled cycles over the channel groups win.
        {
            static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                           {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
            const int row0 = (s.W + 2) * s.C;
            int best_cost = 1 << 30, best_rp = 0, best_ip = 0;
            for (int rp = 0; rp < (CY > 1 ? 16 : 1); ++rp)
                for (int ip = 0; ip < (CG > 1 ? 16 : 1); ++ip) {
                    const int ROW = row0 + 16 * rp, TILE = (s.H + 2) * ROW + 16 * ip;
                    int cost = 0;
                    for (int q = 0; q < std::min(c.NQ, 4); ++q)
                        for (int gi = 0; gi < 4; ++gi) {
                            int addr[16], worst = 1;
                            for (int l = 0; l < 16; ++l) {
                                const int lane = grp[gi][l], col = lane & 15, g = lane >> 4, gg = g < 2 ? g : 2;
                                const int cx = col & (CX - 1), cy = (col >> c.lgCX) & (CY - 1), cg = col >> (c.lgCX + c.lgCY);
                                const int xin = cx * s.S + gg;
                                addr[l] = cg * TILE + cy * s.S * ROW + xin * s.C + 16 * (q ^ ((xin >> c.swz_sh) & c.swz_mask));
                            }
                            for (int slot = 0; slot < 16; ++slot) {
                                int distinct = 0;
                                for (int l = 0; l < 16; ++l) {
                                    if (((addr[l] >> 4) & 15) != slot) continue;
                                    bool seen = false;
                                    for (int m = 0; m < l; ++m) seen = seen || addr[m] == addr[l];
                                    distinct += !seen;
                                }
                                worst = std::max(worst, distinct);
                            }
                            cost += worst;
                        }
                    cost = cost * 64 + rp + ip; // (ties: the smaller pads)
                    if (cost < best_cost) best_cost = cost, best_rp = rp, best_ip = ip;
                }
            c.ROW = row0 + 16 * best_rp;
            c.TILE = (s.H + 2) * c.ROW + 16 * best_ip;
        }
    }
    // output tiles per wave block (the kernel's register budget: chain_tbm) and blocks: nt = TB * NBLK, NBLK a power of two <= 8
    const int TBM = KSC == 1 ? MF_CHAIN_TBM1 : 2;
    for (int i = 0; i < n; ++i) {
        ChainPair &c = pairs[i];
        const int nt = c.N / 16;
        c.NBLK = 0;
        for (int nb = 1; nb <= 8; nb *= 2)
            if (nt % nb == 0 && nt / nb <= TBM) {
                c.NBLK = nb, c.TB = nt / nb;
                break;
            }
        if (!c.NBLK) return false;
        if (i + 1 < n && c.TB == 3) return false; // a 12-byte piece would straddle the next tile's 16-byte groups
    }
    // ---- LDS plan for G images per step ----
    auto lds_for = [&](int G, bool fill, bool dbuf = false) {
        int off = 0, nfill = 0;
        for (int i = 0; i < n; ++i) {
            ChainPair &c = pairs[i];
            // pair i reads the tile pair i-1 wrote: the same region as pair i-1's OWN input if the geometry class (and zero point) agree
            bool share = i > 0 && pairs[i - 1].H == c.H && pairs[i - 1].W == c.W && pairs[i - 1].C == c.C && g[i - 1].izp4 == g[i].izp4 &&
                         pairs[i - 1].S == 1 && pairs[i - 1].ROW == c.ROW && pairs[i - 1].TILE == c.TILE && // (the pitches are per pair: bank model)
                         !(dbuf && i == 1); // (a double-buffered input tile is never a pointwise destination)
            if (share) c.tile_off = pairs[i - 1].tile_off;
            else {
                c.tile_off = off;
                const int copies = (dbuf && i == 0) ? 2 : 1, bytes = (G * c.TILE + 255) & ~255;
                if (fill) a.fill_off[nfill] = off, a.fill_bytes[nfill] = copies * bytes, a.fill_izp4[nfill] = g[i].izp4;
                if (fill && i == 0) a.dbuf = dbuf ? 1 : 0, a.dbuf_stride = dbuf ? bytes : 0;
                ++nfill;
                off += copies * bytes;
            }
        }
        if (fill) a.nfill = nfill;
        off += 512; // slack behind the last tile
        int mid = 0;
        for (int i = 0; i < n; ++i) {
            ChainPair &c = pairs[i];
            c.plimit_img = c.OH * c.OW;
            c.P = G * c.plimit_img, c.NCH = (c.P + 15) / 16, c.PLANE = c.NCH * 256 + 16;
            mid = std::max(mid, c.NQ * c.PLANE);
        }
        a.mid_off = off;
        off += (mid + 255) & ~255;
        for (int i = 0; i < n; ++i) {
            ChainPair &c = pairs[i];
            c.otab_off = -1, c.dst_off = -1;
            c.dROW = c.dTILE = c.dC = c.dswz_sh = c.dswz_mask = 0;
            if (i + 1 < n) {
                const ChainPair &d = pairs[i + 1];
                c.dst_off = d.tile_off, c.dROW = d.ROW, c.dTILE = d.TILE, c.dC = d.C, c.dswz_sh = d.swz_sh, c.dswz_mask = d.swz_mask;
                c.otab_off = off;
                off += (c.P * 4 + 15) & ~15;
            }
        }
        a.q_off = off;
        off += 16;
        return off;
    };
    // ---- cost model (microseconds per image and CU): every phase costs its requantised bytes over the rate a workgroup gets
    // (times the imbalance of its work list over the waves) but never less than a phase's latency (dependent chain + barrier);
    // the chain's HBM bytes are the other roof.  Calibrated on the generated models (profiles/r04/chain_*).
    const double opcost = switches().chain_opcost; // (tuning: weight of the operand-reload term)
    auto estimate = [&](int G, int lds, int &nwave_out) {
        // workgroups per CU the LDS admits (at most two: 8 waves each = four waves per SIMD); with four k steps the registers allow
        // two waves per SIMD, i.e. one workgroup.  (Four-wave workgroups, four per CU, for small tensors were measured in round 4:
        // 8x8x128 -> 128 0.27 ms against 0.22 with two eight-wave workgroups; not kept.)
        const int wgs = (KSC == 4 || lds > 80 * 1024) ? 1 : 2;
        const int nwave = (wgs == 1 && KSC == 1) ? 16 : 8;
        nwave_out = nwave;
        const double waves_per_simd = wgs * nwave / 4.0;
        const double r_cu = waves_per_simd >= 4.0 ? 14.0e3 : 8.0e3; // requantised bytes per microsecond and CU this kernel sustains
        const double r_wg = r_cu / wgs, t_lat = 0.9, t_step = 1.8; // (t_step: what a step costs beyond its phases -- queue draw, top barrier, the
                                                                 // first operand wait behind the staging DMAs; measured 1.5 - 2.5 us on the generated models)
        double step = t_step;
        for (int i = 0; i < n; ++i) {
            const ChainPair &c = pairs[i];
            const int CG = 16 >> (c.lgCX + c.lgCY);
            const int U = c.NQ * (G / CG) * (c.OH >> c.lgCY) * (c.OW >> c.lgCX);
            const int items = ((G * c.OH * c.OW + 15) / 16) * c.NBLK;
            const double imb_d = (double)((U + nwave - 1) / nwave * nwave) / U, imb_p = (double)((items + nwave - 1) / nwave * nwave) / items;
            step += std::max(t_lat, (double)G * c.OH * c.OW * c.C * imb_d / r_wg) + std::max(t_lat, (double)G * c.OH * c.OW * c.N * imb_p / r_wg);
            // a chain (n > 1) fetches the operands of every phase again in every step: ~4 KB per wave for a depthwise phase,
            // (TB KS + TB) KB for a pointwise one, through the CU's vector L1 at ~100 KB per microsecond
            // Measured (profiles/r04/chain_opcost.txt): at G >= 4 the fetches mostly hide under the phases (six 4x4x128 pairs in one
            // launch: 1.12 ms against 1.50 as six launches), at G = 1 - 2 they do not (64x64x16 s2 + 32x32x32: 1.77 against 1.31).
            if (n > 1) step += opcost * (G >= 4 ? 0.3 : 1.5) * nwave * (4.0 + c.TB * c.KS + c.TB) / 100.0 * wgs;
        }
        const double compute = step / (G * wgs);
        const double hbm = ((double)g[0].H * g[0].W * g[0].C + (double)g[n - 1].OH * g[n - 1].OW * g[n - 1].N) / 17.0e3;
        return std::max(compute, hbm);
    };
    // images per step: a multiple of every pair's CG; the cheapest of the doublings that fit
    int bestG = 0, best_nwave = 8;
    double best = 1e30;
    // double buffering of the input tile: where it does not cost a workgroup per CU
    auto want_dbuf = [&](int G) {
        const int l0 = lds_for(G, false, false), l1 = lds_for(G, false, true);
        if (force_dbuf >= 0) return force_dbuf != 0 && l1 <= lds_budget;
        auto cls = [](int l) { return l > 80 * 1024 ? 1 : 2; }; // workgroups per CU
        return l1 <= lds_budget && cls(l0) == cls(l1);
    };
    if (force_G > 0) { // the caller's choice (the measured search of fused_chain_partition): any multiple of the column grids' images
        if (force_G % maxCG != 0 || force_G > 128) return false;
        const int lds = lds_for(force_G, false, want_dbuf(force_G));
        if (lds > lds_budget) return false;
        int nw = 8;
        best = estimate(force_G, lds, nw), bestG = force_G, best_nwave = nw;
    }
    for (int G = maxCG; G <= 128 && force_G <= 0; G *= 2) {
        const bool db = want_dbuf(G);
        const int lds = lds_for(G, false, db);
        if (lds > lds_budget) break;
        int nw = 8;
        const double e = estimate(G, lds, nw);
        if (e < best * 0.97) best = e, bestG = G, best_nwave = nw; // (a larger step must pay for its LDS: 3 % at least)
    }
    if (!bestG) return false;
    const int G = bestG, NW = best_nwave;
    a.G = G, a.nwave = NW, a.est_us_per_image = best;
    a.dbuf = 0, a.dbuf_stride = 0;
    a.max_cg = maxCG;
    a.lds_bytes = lds_for(G, true, want_dbuf(G));
    a.stage_after = 0;
    for (int i = 0; i < n; ++i)
        if (pairs[i].tile_off == pairs[0].tile_off) a.stage_after = i;
    a.npairs = n, a.KSC = KSC;
    a.resident = 0; // set below: a single pair whose every wave stays inside one channel group
    a.hbm_bytes = (double)g[0].H * g[0].W * g[0].C + (double)g[n - 1].OH * g[n - 1].OW * g[n - 1].N;
    a.requant_bytes = 0;
    for (int i = 0; i < n; ++i) {
        ChainPair &c = pairs[i];
        a.requant_bytes += (double)c.OH * c.OW * (c.C + c.N);
        const int CG = 16 >> (c.lgCX + c.lgCY);
        c.UG = G / CG, c.UY = c.OH >> c.lgCY, c.UX = c.OW >> c.lgCX, c.NU = c.UG * c.UY * c.UX;
        // the unit list (channel group, then column, image group, row) cut into one contiguous range per wave
        const int U = c.NQ * c.NU;
        c.single_q = 1;
        for (int w = 0; w < 16; ++w) {
            const int u0 = w < NW ? (int)((long)w * U / NW) : U, u1 = w < NW ? (int)((long)(w + 1) * U / NW) : U;
            const int q = u0 < U ? u0 / c.NU : 0, r = u0 < U ? u0 % c.NU : 0;
            c.ustart[w][0] = q, c.ustart[w][1] = r / (c.UG * c.UY), c.ustart[w][2] = r % (c.UG * c.UY), c.ustart[w][3] = 0;
            c.ucount[w] = u1 - u0;
            if (u1 > u0 && (u1 - 1) / c.NU != q) c.single_q = 0;
        }
        c.pad_ = 0;
        c.rtab = nullptr;
    }
    const bool no_res = switches().chain_no_res; // A/B
    a.resident = (n == 1 && pairs[0].single_q && !no_res) ? 1 : 0;
    return true;
}

void chain_rtab(const ChainPair &c, std::vector<int> &out) {
    const int CG = 16 >> (c.lgCX + c.lgCY), CY = 1 << c.lgCY;
    out.assign((size_t)c.UG * c.UY * 2, 0);
    for (int ug = 0; ug < c.UG; ++ug)
        for (int uy = 0; uy < c.UY; ++uy) {
            out[(size_t)(ug * c.UY + uy) * 2] = ug * CG * c.TILE + uy * CY * c.S * c.ROW;
            out[(size_t)(ug * c.UY + uy) * 2 + 1] = ug * CG * c.plimit_img * 16 + uy * CY * c.OW * 16;
        }
}

double chain_unfused_us_per_image(const ChainGeom *g, int n) {
    // one launch per operator on the run-time-geometry kernels: depthwise ~3.8 TB/s, 1x1 convolution ~4.6 TB/s of algorithmic bytes
    // (bench.py runtime_geometry), small tensors less (a floor per launch is not modelled: batches are large)
    double us = 0;
    for (int i = 0; i < n; ++i) {
        const double in = (double)g[i].H * g[i].W * g[i].C, mid = (double)g[i].OH * g[i].OW * g[i].C, out = (double)g[i].OH * g[i].OW * g[i].N;
        us += (in + mid) / (3.8e6 / 256) + (mid + out) / (4.6e6 / 256);
    }
    return us;
}

template <int KSC, int NW, bool SOLO, bool RES, int MG, uint32_t XR4>
static void launch_chain_t(const int8_t *in, int8_t *out, const ChainArgs &a, int batch, hipStream_t s) {
    // occupancy per (device, LDS size in KiB).  The dynamic-LDS limit of the function is raised to the maximum every time a
    // slot is filled (a smaller later value would make a larger earlier plan unlaunchable).
    static std::atomic<int> cache[LaunchState::MAX_DEV][161];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LaunchState::MAX_DEV) dev = 0;
    std::atomic<int> &slot = cache[dev][(a.lds_bytes + 1023) / 1024];
    int per_cu = slot.load(std::memory_order_relaxed);
    if (per_cu <= 0) {
        (void)hipFuncSetAttribute((const void *)chain_rt<KSC, NW, SOLO, RES, MG, XR4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_rt<KSC, NW, SOLO, RES, MG, XR4>, NW * 64, (size_t)a.lds_bytes) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        slot.store(per_cu, std::memory_order_relaxed);
    }
    const int nsteps = (batch + a.G - 1) / a.G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    ChainArgs b = a;
    b.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * a.hbm_bytes, (double)batch * a.requant_bytes));
    // A handful of short steps per workgroup: static striding (the queue's first draw and its arrival count are a fixed cost per
    // launch, and there is nothing to balance over 8 - 16 steps; profiles/r04/chain_dq_ab.txt: 3x3x128 0.073 -> 0.066 ms, 6x6x64 s2 0.074 -> 0.072).
    {
        const bool forced = switches().dq_cfg_set || switches().dq_tune || switches().chain_dq_auto;
        const double t_step_us = a.est_us_per_image * a.G * per_cu;
        if (!forced && t_step_us < 8.0 && nsteps <= 20 * grid) b.qcfg = 0x100;
    }
    b.queue = dq_slot(b.queue, b.qlaunch);
    hipLaunchKernelGGL((chain_rt<KSC, NW, SOLO, RES, MG, XR4>), dim3(grid), dim3(NW * 64), a.lds_bytes, s, in, out, b, batch);
#if MF_CHAIN_DIAG
    {
        (void)hipStreamSynchronize(s);
        long long h[128];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chain_trace), sizeof(h));
        fprintf(stderr, "[chain trace] %d pairs G %d lds %d nwave %d grid %d per_cu %d dbuf %d qcfg 0x%x\n", a.npairs, a.G, a.lds_bytes, a.nwave, grid, per_cu, a.dbuf, b.qcfg);
        for (int w = 0; w < 2; ++w) {
            const long long *t = h + 64 * w;
            fprintf(stderr, "   wave %d: cycles since the step's top:", w ? 3 : 0);
            for (int i = 1; i < 3 + 4 * a.npairs && i < 60; ++i) fprintf(stderr, " %d:%lld", i, t[i] - t[0]);
            fprintf(stderr, " | after queue top %lld, after staging issue %lld, next step's top %lld\n", t[60] - t[0], t[61] - t[0], t[62] - t[0]);
        }
    }
#endif
}
void launch_chain(const int8_t *in, int8_t *out, const ChainArgs &a, int batch, hipStream_t s) {
#define MF_CHAIN_GO2(KSC, W, SO, RE)                                                         \
    do {                                                                                 \
        if (a.xr) {                                                                      \
            if (a.magic == 2) launch_chain_t<KSC, W, SO, RE, 2, 0x80808080u>(in, out, a, batch, s); \
            else launch_chain_t<KSC, W, SO, RE, 1, 0x80808080u>(in, out, a, batch, s);   \
        } else {                                                                         \
            if (a.magic == 2) launch_chain_t<KSC, W, SO, RE, 2, 0u>(in, out, a, batch, s); \
            else launch_chain_t<KSC, W, SO, RE, 1, 0u>(in, out, a, batch, s);            \
        }                                                                                \
    } while (0)
#define MF_CHAIN_GO(KSC, W)                                                              \
    do {                                                                                 \
        if (a.npairs == 1 && a.resident) MF_CHAIN_GO2(KSC, W, true, true);               \
        else if (a.npairs == 1) MF_CHAIN_GO2(KSC, W, true, false);                       \
        else MF_CHAIN_GO2(KSC, W, false, false);                                         \
    } while (0)
    if (a.KSC == 1 && a.nwave == 16) MF_CHAIN_GO(1, 16);
    else if (a.KSC == 1) MF_CHAIN_GO(1, 8);
    else if (a.KSC == 2) MF_CHAIN_GO(2, 8);
    else if (a.magic == 0) {
        // 256-deep products whose accumulators may leave (-2^22, 2^22) (full-range weights): the v_cvt form of the epilogue.  Only
        // here: below 256 input channels |acc| <= K * 127 * 255 < 2^22 always holds.
        if (a.xr) {
            if (a.npairs == 1 && a.resident) launch_chain_t<4, 8, true, true, 0, 0x80808080u>(in, out, a, batch, s);
            else if (a.npairs == 1) launch_chain_t<4, 8, true, false, 0, 0x80808080u>(in, out, a, batch, s);
            else launch_chain_t<4, 8, false, false, 0, 0x80808080u>(in, out, a, batch, s);
        } else {
            if (a.npairs == 1 && a.resident) launch_chain_t<4, 8, true, true, 0, 0u>(in, out, a, batch, s);
            else if (a.npairs == 1) launch_chain_t<4, 8, true, false, 0, 0u>(in, out, a, batch, s);
            else launch_chain_t<4, 8, false, false, 0, 0u>(in, out, a, batch, s);
        }
    } else MF_CHAIN_GO(4, 8);
#undef MF_CHAIN_GO
#undef MF_CHAIN_GO2
}

} // namespace k
} // namespace mf
