// k_stage.hip -- a run of identical DepthwiseConv2D 3x3 + Conv2D 1x1 pairs on a small tensor as ONE persistent
// kernel: person_detect ops 13..22, five pairs on 6x6x128.
// (src/ops/depthwise_conv_2d.rs:28-105, src/ops/conv_2d.rs:28-108.)
//
// Per inference these tensors are 4.6 KB: as separate pair kernels each layer pays a launch, an HBM round trip and
// a DMA wait per four-image step.  Here G images enter LDS once, run through all NREP pairs there, and the last
// pair's output leaves.  Every intermediate tensor is the reference's int8 tensor, requantised with the
// reference's arithmetic -- it just lives in LDS.
//
// A step (G = 4 images per workgroup, 8 waves) alternates two phases per pair:
//     depthwise (tile -> MID)   as in dwpw_mm: taps on the matrix pipe, unit = 16 columns x 16 channels, wave w owns
//                               channel group w; 9 units per wave
//     pointwise (MID -> tile)   wave w owns output channels 16w .. 16w+15 for ALL pixels (9 chunks of 16): its A
//                               operands and epilogue constants are 8 + 12 VGPRs, and its 4-byte results go straight
//                               into the NEXT depthwise's halo tile (same swizzle) -- the last pair writes a plain
//                               [pixel][128] tensor instead, which is copied to HBM with 16-byte stores
// Only the depthwise phase ends in a workgroup barrier: the pointwise phase needs every channel group of MID, but the
// next depthwise of wave w reads exactly the channels wave w has just written (LDS operations of a wave complete in
// order), and it writes the OTHER of two MID buffers, which no slower wave is still reading.
// Inside a phase the items run as a three-stage pipeline -- operands of item i+2 loaded, the MFMAs of item i+1 issued
// between the pieces of item i's requantisation (MFMAs are asynchronous; the wave itself executes in order) -- and
// the operands of the next phase (12 + 12 or 8 + 12 VGPRs, L2-resident) are fetched before the current one's work.
// The next step's images are DMA-staged as soon as the last depthwise has read the tile.
// LDS: halo tiles 34 KB + two MID buffers 18 KB each = 70 KB -> two workgroups per CU.
//
// What was tried and dropped (r02, measured): carrying on through the stride-2 pair, the 3x3x256 pair and the tail in
// the same kernel.  Those phases have 3 .. 6 items per wave and five different operand sets per wave, so they ran at a
// third of the 6x6 phases' efficiency (17.5 k of a step's 51 k cycles for 13 % of its work) -- slower than the three
// separate launches they replace.
#include "k_common.hpp"

#include <type_traits>

#include <cstdio>

#ifndef MF_STAGE_DIAG
#define MF_STAGE_DIAG 0 // 2: cycle stamps of block 0 / wave 0 at the phase boundaries of its 2nd step (never shipped)
#endif
#ifndef MF_STAGE_KO
#define MF_STAGE_KO 0 // knock-out timing experiments (WRONG results, never shipped): 1 no requantisation, 2 no copy-out to HBM, 4 no staging after the
                      // first step, 8 no barriers inside a step, 16 operand loads only in the first step
#endif
#ifndef MF_STAGE_LDS_KB
#define MF_STAGE_LDS_KB 70 // (tuning: 100 forces one workgroup per CU)
#endif

#ifndef MF_STAGE_PDIAG
#define MF_STAGE_PDIAG 0
#endif
#ifndef MF_STAGE_CNT_WAIT
#define MF_STAGE_CNT_WAIT 0 // (tuning) 1: the wait at the top of a step leaves the previous step's output stores in flight
#endif
// 1: the vector-memory issues that follow the depthwise barrier -- the next record, the next depthwise's operands and, in the last pair,
// the next step's staging DMAs -- are issued BEHIND the first MFMAs of the pointwise phase instead of in front of it.  vmcnt retires in
// order: the phase's first use of its own operands and record (loaded a phase earlier) is compiled into `s_waitcnt vmcnt(0)`, and with
// those issues already in flight that wait was a full L2 round trip (in the last pair: the HBM round trip of the DMAs) at the top of
// every pointwise phase (scripts/asm_dma_waits.py, the listing of round 5's kernel: `global_load rdw; s_waitcnt vmcnt(0);
// v_readfirstlane rpw`).
#ifndef MF_STAGE_LATE_LOADS
#define MF_STAGE_LATE_LOADS 1
#endif
#ifndef MF_STAGE_CNT_WAIT
#define MF_STAGE_CNT_WAIT 0 // (tuning) 1: the wait at the top of a step leaves the previous step's output stores in flight
#endif
#ifndef MF_STAGE_FENCE
#define MF_STAGE_FENCE 0 // (tuning) scheduling fences -- bit 0: inside a unit (the hand interleave of MFMAs and epilogue halves of rounds
                         // 3-4), bit 1: at the end of a unit.  With the two-instruction epilogue of round 5 the compiler's own schedule of
                         // the nine units of a phase is 1 % ahead of the fenced one (four same-box alternations), so none by default.
#endif
#define MF_SB_IN() do { if (MF_STAGE_FENCE & 1) __builtin_amdgcn_sched_barrier(0); } while (0)
#define MF_SB_END() do { if (MF_STAGE_FENCE & 2) __builtin_amdgcn_sched_barrier(0); } while (0)
namespace mf {
namespace k {

#if MF_STAGE_DIAG == 2
__device__ long long g_stage_trace[32];
__device__ long long g_stage_blk[1024][2]; // 100 MHz clock at entry / exit of every workgroup
#define MF_TR(k) do { if (blockIdx.x == 0 && wave == 0 && lane == 0 && trace_step == 1) g_stage_trace[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MF_TR(k) do { } while (0)
#endif

namespace {
struct DwW {        // depthwise operands of one 16-channel group
    v4i A[3];
    float4 a, s;
    int4 k;
};
struct PwW {        // pointwise operands of one 16-output-channel tile, K = 128
    v4i A[2];
    float4 a, s;
    int4 k;
};
struct Taps {
    v4i b[3];
};
} // namespace

template <int G, int NTHR, int MG, uint32_t XR4>
__global__ __launch_bounds__(NTHR, 4) void stage_6x6x128(const int8_t *__restrict__ in, int8_t *__restrict__ out, StageArgs p,
                                                         int batch) {
    static_assert(G == 4 && NTHR == 512, "the column grid below is written for 4 images and 8 waves");
    epi_enter<MG>();
    // any number of pairs >= 2: the MID buffers alternate so that the LAST pair uses region B and finds region A free for
    // its plain output
    const int NREP = p.nrep, par = (NREP & 1) ^ 1;
    constexpr int NWAVE = 8;
    // 6x6x128 halo tile.  Depthwise column grid: 2 rows x 2 x x 4 images (y fastest).  Row pitch +32, image pitch +64
    // and the 16-byte group index XOR (x & 1) make every tap read conflict-free and the other three access patterns
    // of a pair 1.5x / 2x / 1x their ideal LDS cycles (scripts/model/stage_banks.py: 270 cycles per wave and pair,
    // against 464 for the dwpw_mm<6,6,128> layout).
    constexpr int LP6 = 128, ROW6 = 128 + 768 + 128 + 32, TILE6 = 8 * ROW6 + 64, TS6 = 0x001;
    constexpr int PIX6 = 36, NP6 = G * PIX6;
    constexpr int PLANE6 = NP6 * 16;           // MID planes [16-channel group][pixel][16 B]
    constexpr int IMG6 = PIX6 * 128;
    constexpr int OFF_B = G * TILE6 + 512;     // MID of the even pairs
    constexpr int OFF_A = OFF_B + 8 * PLANE6;  // MID of the odd pairs; the last pair's plain [pixel][128] output
    constexpr int OFF_Q = OFF_A + 8 * PLANE6;  // the step queue's two ints
    static_assert(OFF_Q + 16 <= MF_STAGE_LDS_KB * 1024, "LDS budget");
    static_assert(NP6 * 128 == 8 * PLANE6, "the plain output fills region A exactly");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < OFF_B / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);

    // ---- lane constants ----
    const int col = lane & 15, g = lane >> 4;
    // depthwise: columns = (row parity cy, x parity cx, image cg), units = 3 row pairs x 3 x pairs
    const int a_cy = col & 1, a_cx = (col >> 1) & 1, a_cg = col >> 2;
    const int a_xl = a_cx + g - 1;
    const int tb6 = a_cg * TILE6 + a_cy * ROW6 + LP6 + a_xl * 128 + 16 * (wave ^ tile_swz<TS6>(a_xl));
    const int mb6 = wave * PLANE6 + (a_cg * PIX6 + a_cy * 6 + a_cx) * 16 + 4 * g;
    // pointwise: lane (pixel column pcol, pg) of chunk c handles pixel 16c + pcol, output channels 16 wave + 4 pg ..
    const int pcol = lane & 15, pg = lane >> 4;
    int o6[9]; // where chunk c's result goes in the halo tile (this wave's 16-channel group, swizzled)
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int pix = c * 16 + pcol, img = pix / PIX6, r = pix % PIX6, y = r / 6, x = r % 6;
        o6[c] = img * TILE6 + (y + 1) * ROW6 + LP6 + x * 128 + 16 * (wave ^ tile_swz<TS6>(x)) + 4 * pg;
    }
    // ... and in the last pair's plain [pixel][128] output (+ 2048 per chunk).  The 16 lanes of a column group write the
    // same 16-byte slot of 16 consecutive pixels, 128 bytes apart = one bank: the slot index is XOR-ed with the pixel's
    // low 3 bits (16-way -> 2-way conflict; the copy to HBM below undoes it).
    const int oplain = OFF_A + pcol * 128 + 16 * (wave ^ (pcol & 7)) + 4 * pg;

    // The pair table is read with SCALAR loads (constant address space): a phase's operand fetches then wait for a
    // scalar-cache hit, not for a vector load of their own pointers.  The pointer is re-laundered every step so that
    // operand addresses are formed where they are used instead of being hoisted out of the step loop into VGPRs.
    typedef __attribute__((address_space(4))) const StagePair c_pair;
    c_pair *pairs = (c_pair *)(uintptr_t)p.pairs;
    // (the pointers come out of a table in memory, so the compiler no longer knows they are global: say so, or every
    // fetch becomes a flat load with a 64-bit per-lane address)
    typedef __attribute__((address_space(1))) const v4i g_v4i;
    const uint32_t l16 = (uint32_t)lane * 16u, g16 = (uint32_t)g * 16u, pg16 = (uint32_t)pg * 16u;
    auto ld16 = [](const void *base, uint32_t off) { return *(g_v4i *)((uintptr_t)base + off); };
    auto ldf4 = [&](const void *base, uint32_t off) {
        const v4i v = ld16(base, off);
        return make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3]));
    };
    auto ldi4 = [&](const void *base, uint32_t off) {
        const v4i v = ld16(base, off);
        return make_int4(v[0], v[1], v[2], v[3]);
    };
    auto load_dw = [&](int pair) { // channel group `wave`
        c_pair &sp = pairs[pair];
        DwW w;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) w.A[ty] = ld16((const uint8_t *)sp.dw_wmm + (wave * 3 + ty) * 1024, l16);
        w.a = ldf4((const uint8_t *)sp.dwA + wave * 64, g16);
        w.s = ldf4((const uint8_t *)sp.dwS + wave * 64, g16);
        w.k = ldi4((const uint8_t *)sp.dwK + wave * 64, g16); // (the host added the bit-pattern offset: no dependent VALU here)
        return w;
    };
    auto load_pw = [&](int pair) { // output tile `wave`
        c_pair &sp = pairs[pair];
        PwW w;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w.A[ks] = ld16((const uint8_t *)sp.pw_w + (wave * 2 + ks) * 1024, l16);
        w.a = ldf4((const uint8_t *)sp.pwA + wave * 64, pg16);
        w.s = ldf4((const uint8_t *)sp.pwS + wave * 64, pg16);
        w.k = ldi4((const uint8_t *)sp.pwK + wave * 64, pg16);
        return w;
    };
    auto dw_load = [&](int taddr) {
        Taps t;
        t.b[0] = *(const v4i *)(lds + taddr);
        t.b[1] = *(const v4i *)(lds + taddr + ROW6);
        t.b[2] = *(const v4i *)(lds + taddr + 2 * ROW6);
        return t;
    };

    // mode 3: the first patch record of (phase, this wave) -- kernels.hpp EpiPatchRec -- fetched like the operands, one phase ahead and
    // with a VECTOR load (every lane the same 8 bytes): a scalar load at the top of a phase has its whole latency exposed there, and
    // its s_waitcnt lgkmcnt(0) drains the phase's LDS pipeline with it (0.474 -> 0.557 ms for the kernel, measured).
    typedef int v2i __attribute__((ext_vector_type(2)));
    auto ld_rec = [&](int phase) {
        typedef __attribute__((address_space(1))) const v2i g_v2i;
        if constexpr (MG == 3) return *(g_v2i *)((uintptr_t)p.patch_tab + (size_t)((phase * 8 + wave) * 2) * sizeof(EpiPatchRec));
        else return v2i{0, 0};
    };
    auto rec_of = [](const v2i &r) {
        EpiPatchRec rec{__builtin_amdgcn_readfirstlane(r[0]), __builtin_amdgcn_readfirstlane(r[1])};
#if MF_STAGE_PDIAG & 1 // (timing experiments, WRONG results: 1 only the unpatched copy exists; 2 every wave takes it, all copies compiled)
        asm volatile("" ::"s"(rec.P), "s"(rec.meta));
        rec.P = 0;
#elif MF_STAGE_PDIAG & 2
        int z = 0;
        asm volatile("" : "+s"(z));
        rec.P &= z;
#endif
        return rec;
    };

    auto stage = [&](int st) { // G images, 6 rows each, one 768-byte DMA per row, group index swizzled like TS6
        const int src_lane = lane ^ tile_swz<TS6>(lane >> 3);
#pragma unroll
        for (int k = 0; k < (G * 6 + NWAVE - 1) / NWAVE; ++k) {
            const int r = k * NWAVE + wave;
            const int gi = r / 6, y = r % 6;
            if (r < G * 6 && st * G + gi < batch && lane < 48)
                dma16(in + ((size_t)(st * G + gi) * IMG6 + y * 768 + src_lane * 16), lds + gi * TILE6 + (y + 1) * ROW6 + LP6);
        }
    };

    DynSteps dq;
    dq.init(lds + OFF_Q, p.queue, tid, p.qcfg);
    wg_sync(); // halo fill complete before any DMA lands
    const int nsteps = (batch + G - 1) / G;
    if (dq.step < nsteps) stage(dq.step);
    v2i rdw = ld_rec(0);
    DwW wd = load_dw(0);
    int ko_steps = 0; // (steps done: the knock-out switches 4 / 16 act from the second step on)
#if MF_STAGE_KO & 16
    PwW wpk = load_pw(0);
#endif

#if MF_STAGE_DIAG == 2
    int trace_step = 0;
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) g_stage_blk[blockIdx.x][0] = (long long)__builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && wave == 0 && lane == 0) {
        g_stage_trace[25] = (long long)__builtin_readcyclecounter();
        g_stage_trace[27] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
    for (; dq.step < nsteps; dq.advance(tid)) {
        const int step = dq.step;
        MF_TR(24);
#if MF_STAGE_CNT_WAIT
        // This step's images and the first depthwise operands must have landed; the previous step's copy-out stores -- two or three per
        // wave, issued after them, and vmcnt retires in order -- may stay in flight.  (The bare barrier: __syncthreads() carries a fence
        // that hipcc completes with vmcnt(0) while it believes an LDS-DMA may be outstanding.)
        if (ko_steps > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
#if MF_STAGE_CNT_WAIT
        // (tuning, measured +-0: this step's images and the first depthwise operands must have landed; the previous step's copy-out
        // stores -- two or three per wave, issued after them, and vmcnt retires in order -- stay in flight; the bare barrier because
        // __syncthreads() carries a fence that hipcc completes with vmcnt(0) while it believes an LDS-DMA may be outstanding)
        if (ko_steps > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync(); // this step's images are in the tile; the previous step's output has been copied out of region A
#endif
#endif
        dq.top(tid);
        asm volatile("" : "+s"(pairs));
        const int gvalid = min(G, batch - step * G);
        MF_TR(0);

        for (int rep = 0; rep < NREP; ++rep) {
            const int mid = ((rep + par) & 1) ? OFF_A : OFF_B; // this pair's MID buffer
#if MF_STAGE_KO & 16
            if (ko_steps == 0 || rep == 0) wpk = load_pw(ko_steps == 0 ? rep : 0);
            const PwW wp = wpk;
#else
            const v2i rpw = ld_rec(2 * rep + 1);
            const PwW wp = load_pw(rep);                // lands during the depthwise phase
#endif
            // ---------------- depthwise: tile -> MID ----------------
            // PR: where this wave's patched channel sits (mode 3, kernels.hpp EpiPatchRec) -- -1: none (almost every wave and
            // operator); 0 .. 3: that accumulator of a lane; 4: apply the wave's (up to two) records exactly (epi_patch_apply).
            // A compile-time constant per copy of the loop: the hand-interleaved MFMA / epilogue schedule below does not survive a
            // branch inside the loop (0.47 -> 0.56 ms with a never-taken one).  A patched accumulator is ONE value out of millions,
            // and every wave of the phase waits at the barrier behind it for the slowest: so the copies 0 .. 3 only DETECT it -- one
            // compare into a scalar mask per unit, the unpatched arithmetic otherwise -- and the phase is redone by copy 4 in the rare
            // step that saw one (it reads a buffer nobody writes during the phase and rewrites this wave's own outputs: idempotent).
            // Returns the lanes that saw their patched accumulator.
            auto dw_phase = [&](auto pr_tag, const EpiPatchRec &dpr, const EpiPatchRec &dpr2) -> unsigned long long {
                constexpr int PR = decltype(pr_tag)::value;
                const float lo = pairs[rep].dw_lo, hi = pairs[rep].dw_hi;
                const int Pl = g == ((dpr.meta >> 2) & 3) ? dpr.P : 0x7fffffff; // (no biased accumulator is 0x7fffffff)
                unsigned long long hit = 0;
                const int mb = mid + mb6;
                auto toff = [](int u) { return (u / 3) * 2 * ROW6 + (u % 3) * 2 * 128; };
                auto moff = [](int u) { return ((u / 3) * 12 + (u % 3) * 2) * 16; };
                Taps t1 = dw_load(tb6 + toff(0)), t2 = dw_load(tb6 + toff(1));
                v4i acc = {wd.k.x, wd.k.y, wd.k.z, wd.k.w};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[0], t1.b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[1], t1.b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[2], t1.b[2], acc, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const bool more = u + 1 < 9;
                    Taps t3 = t2;
                    if (u + 2 < 9) t3 = dw_load(tb6 + toff(u + 2));
                    v4i nxt = {wd.k.x, wd.k.y, wd.k.z, wd.k.w};
                    MF_SB_IN();
                    if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[0], t2.b[0], nxt, 0, 0, 0);
                    // (behind the next unit's first MFMA, like the epilogue itself: acc's own MFMAs have had their latency by now)
                    if constexpr (PR == 4) epi_patch_apply(acc, dpr, g), epi_patch_apply(acc, dpr2, g);
                    const float r0 = epi_value<MG>(acc[0], wd.a.x, wd.s.x, lo, hi);
                    const float r1 = epi_value<MG>(acc[1], wd.a.y, wd.s.y, lo, hi);
                    if constexpr (PR >= 0 && PR < 4) {
                        hit |= __builtin_amdgcn_ballot_w64(acc[PR] == Pl);
                        asm volatile("" : "+s"(hit)); // one running mask (left alone, the compiler keeps nine and spills scalars)
                    }
                    MF_SB_IN();
                    if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[1], t2.b[1], nxt, 0, 0, 0);
                    const float r2 = epi_value<MG>(acc[2], wd.a.z, wd.s.z, lo, hi);
                    const float r3 = epi_value<MG>(acc[3], wd.a.w, wd.s.w, lo, hi);
                    MF_SB_IN();
                    if (more && !(MF_STAGE_KO & 32)) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd.A[2], t2.b[2], nxt, 0, 0, 0); // (knock-out 32: two of three)
#if MF_STAGE_KO & 1
                    *(uint32_t *)(lds + mb + moff(u)) = (uint32_t)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
#else
                    *(uint32_t *)(lds + mb + moff(u)) = epi_pack4<MG, XR4>(r0, r1, r2, r3);
#endif
                    MF_SB_END();
                    acc = nxt, t2 = t3;
                }
                return hit;
            };
            if constexpr (MG == 3) { // the copy of the loop is chosen by a chain of scalar branches
                const EpiPatchRec dpr = rec_of(rdw);
                const int sel = dpr.P == 0 ? -1 : ((dpr.meta & 32) ? 4 : (dpr.meta & 3));
                unsigned long long redo = 0;
                if (sel < 0) dw_phase(std::integral_constant<int, -1>{}, dpr, dpr);
                else if (sel == 0) redo = dw_phase(std::integral_constant<int, 0>{}, dpr, dpr);
                else if (sel == 1) redo = dw_phase(std::integral_constant<int, 1>{}, dpr, dpr);
                else if (sel == 2) redo = dw_phase(std::integral_constant<int, 2>{}, dpr, dpr);
                else if (sel == 3) redo = dw_phase(std::integral_constant<int, 3>{}, dpr, dpr);
                else redo = ~0ull; // two patched channels in this wave's 16: straight to the exact copy
                if (__builtin_expect(redo != 0, 0))
                    dw_phase(std::integral_constant<int, 4>{}, dpr, epi_patch_load(p.patch_tab, ((2 * rep) * 8 + wave) * 2 + 1));
            } else {
                (void)dw_phase(std::integral_constant<int, -1>{}, EpiPatchRec{0, 0}, EpiPatchRec{0, 0});
            }
            MF_TR(1 + 3 * rep);
#if MF_STAGE_KO & 8
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            wg_sync(); // MID complete (every channel group); every wave is done reading the tile
#endif
            MF_TR(2 + 3 * rep);
            const bool last = rep == NREP - 1;
            auto next_issues = [&]() {
                if (last) {
                    const int next = dq.nxt;
                    if (next < nsteps && !((MF_STAGE_KO & 4) && ko_steps > 0)) stage(next); // the tile is dead: the next step's images fly under the last pointwise phase
                }
                rdw = ld_rec(last ? 0 : 2 * rep + 2);
                if (!((MF_STAGE_KO & 16) && ko_steps > 0)) wd = load_dw(last ? 0 : rep + 1); // the next depthwise's operands land during the pointwise phase
            };
            if constexpr (!MF_STAGE_LATE_LOADS) next_issues();
            // ---------------- pointwise: MID -> tile (last pair: -> plain output in region A) ----------------
            // ISSUE: this call also issues next_issues() (MF_STAGE_LATE_LOADS), behind its first MFMAs
            auto pw_phase = [&](auto pr_tag, auto issue_tag, const EpiPatchRec &ppr, const EpiPatchRec &ppr2) -> unsigned long long { // (PR as in dw_phase)
                constexpr int PR = decltype(pr_tag)::value;
                constexpr bool ISSUE = decltype(issue_tag)::value && MF_STAGE_LATE_LOADS != 0;
                const float lo = pairs[rep].pw_lo, hi = pairs[rep].pw_hi;
                const int Pl = pg == ((ppr.meta >> 2) & 3) ? ppr.P : 0x7fffffff;
                unsigned long long hit = 0;
                const int rb = mid + pg * PLANE6 + pcol * 16;
                v4i c0 = *(const v4i *)(lds + rb), c1 = *(const v4i *)(lds + rb + 4 * PLANE6);
                v4i acc = {wp.k.x, wp.k.y, wp.k.z, wp.k.w};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[0], c0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[1], c1, acc, 0, 0, 0);
                c0 = *(const v4i *)(lds + rb + 256), c1 = *(const v4i *)(lds + rb + 4 * PLANE6 + 256);
                if constexpr (ISSUE) { // (the fences keep the scheduler from hoisting the loads back in front of the MFMAs' wait)
                    __builtin_amdgcn_sched_barrier(0);
                    next_issues();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    const bool more = c + 1 < 9;
                    v4i e0 = c0, e1 = c1;
                    if (c + 2 < 9) {
                        e0 = *(const v4i *)(lds + rb + (c + 2) * 256);
                        e1 = *(const v4i *)(lds + rb + 4 * PLANE6 + (c + 2) * 256);
                    }
                    v4i nxt = {wp.k.x, wp.k.y, wp.k.z, wp.k.w};
                    MF_SB_IN();
                    if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[0], c0, nxt, 0, 0, 0);
                    if constexpr (PR == 4) epi_patch_apply(acc, ppr, pg), epi_patch_apply(acc, ppr2, pg);
                    const float r0 = epi_value<MG>(acc[0], wp.a.x, wp.s.x, lo, hi);
                    const float r1 = epi_value<MG>(acc[1], wp.a.y, wp.s.y, lo, hi);
                    if constexpr (PR >= 0 && PR < 4) {
                        hit |= __builtin_amdgcn_ballot_w64(acc[PR] == Pl);
                        asm volatile("" : "+s"(hit)); // one running mask (left alone, the compiler keeps nine and spills scalars)
                    }
                    MF_SB_IN();
                    if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[1], c1, nxt, 0, 0, 0);
                    const float r2 = epi_value<MG>(acc[2], wp.a.z, wp.s.z, lo, hi);
                    const float r3 = epi_value<MG>(acc[3], wp.a.w, wp.s.w, lo, hi);
#if MF_STAGE_KO & 1
                    const uint32_t d = (uint32_t)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
#else
                    const uint32_t d = epi_pack4<MG, XR4>(r0, r1, r2, r3);
#endif
                    if (last) *(uint32_t *)(lds + oplain + c * 2048) = d;
                    else *(uint32_t *)(lds + o6[c]) = d;
                    MF_SB_END();
                    acc = nxt, c0 = e0, c1 = e1;
                }
                return hit;
            };
            if constexpr (MG == 3) {
                const EpiPatchRec ppr = rec_of(rpw);
                const int sel = ppr.P == 0 ? -1 : ((ppr.meta & 32) ? 4 : (ppr.meta & 3));
                unsigned long long redo = 0;
                using yes = std::true_type;
                if (sel < 0) pw_phase(std::integral_constant<int, -1>{}, yes{}, ppr, ppr);
                else if (sel == 0) redo = pw_phase(std::integral_constant<int, 0>{}, yes{}, ppr, ppr);
                else if (sel == 1) redo = pw_phase(std::integral_constant<int, 1>{}, yes{}, ppr, ppr);
                else if (sel == 2) redo = pw_phase(std::integral_constant<int, 2>{}, yes{}, ppr, ppr);
                else if (sel == 3) redo = pw_phase(std::integral_constant<int, 3>{}, yes{}, ppr, ppr);
                else {
                    redo = ~0ull; // (two patched channels: straight to the exact copy, which never issues -- a redo must not issue twice)
                    if constexpr (MF_STAGE_LATE_LOADS != 0) next_issues();
                }
                if (__builtin_expect(redo != 0, 0))
                    pw_phase(std::integral_constant<int, 4>{}, std::false_type{}, ppr, epi_patch_load(p.patch_tab, ((2 * rep + 1) * 8 + wave) * 2 + 1));
            } else {
                (void)pw_phase(std::integral_constant<int, -1>{}, std::true_type{}, EpiPatchRec{0, 0}, EpiPatchRec{0, 0});
            }
            MF_TR(3 + 3 * rep);
            // NO barrier here (see the header): the next depthwise of this wave reads only what this wave has written
            asm volatile("" ::: "memory");
        }

        // ---------------- the last pair's output leaves the chip ----------------
#if MF_STAGE_KO & 8
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        wg_sync(); // every channel group of the output is in region A
#endif
        MF_TR(16);
        {
            const int nbytes = gvalid * PIX6 * 128;
            int8_t *dst = out + (size_t)step * G * PIX6 * 128;
            for (int i = tid * 16; i < nbytes; i += NTHR * 16) {
                const int pix = i >> 7, slot = (i >> 4) & 7;
                const uint4 vv = *(const uint4 *)(lds + OFF_A + pix * 128 + 16 * (slot ^ (pix & 7)));
                if (!(MF_STAGE_KO & 2) || vv.x == 0x12345678u) st_out(dst + i, vv);
            }
        }
        MF_TR(17);
        ++ko_steps;
#if MF_STAGE_DIAG == 2
        ++trace_step;
#endif
    }
    dq.finish(tid);
#if MF_STAGE_DIAG == 2
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) g_stage_blk[blockIdx.x][1] = (long long)__builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && wave == 0 && lane == 0) {
        g_stage_trace[26] = (long long)__builtin_readcyclecounter();
        g_stage_trace[28] = (long long)__builtin_amdgcn_s_memrealtime();
        g_stage_trace[29] = trace_step;
    }
#endif
}

// ---- launcher ----
const char *stage_name(int H, int W, int C, int npairs) {
    static const char *names[17] = {nullptr, nullptr, "stage_6x6x128<4,512,2>", "stage_6x6x128<4,512,3>", "stage_6x6x128<4,512,4>",
                                    "stage_6x6x128<4,512,5>", "stage_6x6x128<4,512,6>", "stage_6x6x128<4,512,7>", "stage_6x6x128<4,512,8>",
                                    "stage_6x6x128<4,512,9>", "stage_6x6x128<4,512,10>", "stage_6x6x128<4,512,11>", "stage_6x6x128<4,512,12>",
                                    "stage_6x6x128<4,512,13>", "stage_6x6x128<4,512,14>", "stage_6x6x128<4,512,15>", "stage_6x6x128<4,512,16>"};
    return (H == 6 && W == 6 && C == 128 && npairs >= 2 && npairs <= 16) ? names[npairs] : nullptr; // (the run length is a kernel argument)
}
bool launch_stage(int H, int W, int C, int npairs, const int8_t *in, int8_t *out, const StageArgs &a_in, int batch, hipStream_t s) {
    if (!stage_name(H, W, C, npairs)) return false;
    constexpr int G = 4, NTHR = 512;
    StageArgs a = a_in;
    a.nrep = npairs;
    a.qcfg = dq_config((batch + G - 1) / G, 512, dq_est_us((double)batch * 2 * H * W * C, (double)batch * 2 * npairs * H * W * C));
    a.queue = dq_slot(a.queue, a.qlaunch);
    constexpr int lds = MF_STAGE_LDS_KB * 1024;
    const int nsteps = (batch + G - 1) / G;
    int per_cu = 0, grid = 0;
    // epilogue mode (k_common.hpp): 2 when every clamp of the run is the element type's whole range, else 1
#define MF_STAGE_GO(MG, XR)                                                                                        \
    do {                                                                                                           \
        static LaunchState st_;                                                                                    \
        per_cu = prepared(st_, stage_6x6x128<G, NTHR, MG, XR>, NTHR, lds);                                   \
        grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;                                                      \
        hipLaunchKernelGGL((stage_6x6x128<G, NTHR, MG, XR>), dim3(grid), dim3(NTHR), lds, s, in, out, a, batch); \
    } while (0)
    if (a.mode == 3) { // the single-fma form for every operator of the run (its code does not depend on the element type)
        MF_STAGE_GO(3, 0u);
    } else if (a.xr4) { // u8 element type: the stored byte is the value ^ 0x80 (XR4, see kernels.hpp)
        if (a.mode == 2) MF_STAGE_GO(2, 0x80808080u); else MF_STAGE_GO(1, 0x80808080u);
    } else {
        if (a.mode == 2) MF_STAGE_GO(2, 0u); else MF_STAGE_GO(1, 0u);
    }
#undef MF_STAGE_GO
#if MF_STAGE_DIAG == 2
    {
        static int calls = 0;
        if (++calls == 3) {
            (void)hipStreamSynchronize(s);
            long long h[32];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage_trace), sizeof(h));
            fprintf(stderr, "[stage trace] grid %d per_cu %d; cycles since step start:", grid, per_cu);
            for (int i = 0; i < 18; ++i) fprintf(stderr, " %d:%lld", i, h[i] - h[24]);
            fprintf(stderr, " | kernel: %lld shader cycles, %lld ticks of the 100 MHz clock, %lld steps; prologue %lld cycles\n", h[26] - h[25], h[28] - h[27], h[29],
                    h[24] - h[25]);
            static long long hb[1024][2];
            (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(g_stage_blk), sizeof(hb));
            long long t0 = hb[0][0];
            for (int i = 0; i < grid && i < 1024; ++i) t0 = hb[i][0] < t0 ? hb[i][0] : t0;
            fprintf(stderr, "[stage blocks] start/end in us since the first start:");
            for (int i = 0; i < grid && i < 1024; i += 1) fprintf(stderr, " %d:%.1f-%.1f", i, (hb[i][0] - t0) * 0.01, (hb[i][1] - t0) * 0.01);
            fprintf(stderr, "\n");
        }
    }
#endif
    return true;
}

} // namespace k
} // namespace mf
