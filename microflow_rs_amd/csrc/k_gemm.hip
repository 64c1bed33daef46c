// k_gemm.hip -- FullyConnected (src/ops/fully_connected.rs:24-82): row-wave kernels for few outputs and the
// dense int8 MFMA GEMM.
//
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
#include "k_common.hpp"

namespace mf {
namespace k {

// FullyConnected with few outputs and a long reduction (speech: K=4000, N=4):
// one wavefront per input row, 16-byte coalesced loads, DPP/shuffle reduction.
// Memory-bound: every input byte is read exactly once.
template <int N>
__global__ __launch_bounds__(256) void fc_rowwave(const int8_t *__restrict__ in,
                                                  int8_t *__restrict__ out, FcArgs p, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int K16 = p.K >> 4; // K % 16 == 0 is a routing precondition
    for (size_t row = wave; row < rows; row += nwaves) {
        const uint4 *x = (const uint4 *)(in + row * (size_t)p.K);
        int dot[N], rs = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) dot[j] = 0;
        for (int k = lane; k < K16; k += 64) {
            const uint4 v = x[k];
            rs = sdot4(v.x, 0x01010101u, rs);
            rs = sdot4(v.y, 0x01010101u, rs);
            rs = sdot4(v.z, 0x01010101u, rs);
            rs = sdot4(v.w, 0x01010101u, rs);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const uint4 w = ((const uint4 *)(p.w + (size_t)j * p.K))[k];
                dot[j] = sdot4(v.x, w.x, dot[j]);
                dot[j] = sdot4(v.y, w.y, dot[j]);
                dot[j] = sdot4(v.z, w.z, dot[j]);
                dot[j] = sdot4(v.w, w.w, dot[j]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rs += __shfl_xor(rs, off, 64);
#pragma unroll
            for (int j = 0; j < N; ++j) dot[j] += __shfl_xor(dot[j], off, 64);
        }
        if (lane < N) {
            int d = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) d = (lane == j) ? dot[j] : d;
            const int acc = d - p.wzp * rs + p.Kc[lane];
            out[row * N + lane] = (int8_t)(requant(acc, p.A[lane], p.S, p.lo_f, p.hi_f) ^ p.xr);
        }
    }
}

// fc_rowwave followed by the softmax over its N outputs (speech.tflite: FullyConnected 4000 -> 4,
// Softmax) in one launch: the FC result of a row IS the whole [1][N] softmax tensor, so lanes
// 0..N-1 exchange their exp-table entries by shuffles and every one of them forms the sum in the
// reference's order (src/ops/softmax.rs:20-21) before quantising its own probability.
template <int N>
__global__ __launch_bounds__(256) void fc_rowwave_softmax(const int8_t *__restrict__ in, int8_t *__restrict__ out,
                                                          FcArgs p, SoftmaxArgs sm, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int K16 = p.K >> 4;
    for (size_t row = wave; row < rows; row += nwaves) {
        const uint4 *x = (const uint4 *)(in + row * (size_t)p.K);
        int dot[N], rs = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) dot[j] = 0;
        for (int k = lane; k < K16; k += 64) {
            const uint4 v = x[k];
            rs = sdot4(v.x, 0x01010101u, rs);
            rs = sdot4(v.y, 0x01010101u, rs);
            rs = sdot4(v.z, 0x01010101u, rs);
            rs = sdot4(v.w, 0x01010101u, rs);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const uint4 w = ((const uint4 *)(p.w + (size_t)j * p.K))[k];
                dot[j] = sdot4(v.x, w.x, dot[j]);
                dot[j] = sdot4(v.y, w.y, dot[j]);
                dot[j] = sdot4(v.z, w.z, dot[j]);
                dot[j] = sdot4(v.w, w.w, dot[j]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rs += __shfl_xor(rs, off, 64);
#pragma unroll
            for (int j = 0; j < N; ++j) dot[j] += __shfl_xor(dot[j], off, 64);
        }
        int d = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) d = (lane == j) ? dot[j] : d;
        const int ch = lane < N ? lane : 0;
        const int acc = d - p.wzp * rs + p.Kc[ch];
        // the FullyConnected output byte as it would be stored (i8 domain), then softmax's table index
        const int y = (int)(int8_t)(requant(acc, p.A[ch], p.S, p.lo_f, p.hi_f) ^ p.xr);
        const float e = sm.exp_table[y + 128];
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) sum = __fadd_rn(sum, __shfl(e, j, 64));
        const float prob = __fdiv_rn(e, sum);
        const float q = __fadd_rn(__fdiv_rn(prob, sm.oscale), sm.ozp_f);
        const float r = __fadd_rn(q, __builtin_copysignf(0x1.fffffep-2f, q));
        const int qi = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, sm.sat_lo, sm.sat_hi);
        if (lane < N) out[row * N + lane] = (int8_t)(qi ^ sm.xr);
    }
}

// ------------------------------------------------------------------------
// FAST PATH 4 -- FullyConnected as a dense int8 MFMA GEMM (BASELINE config 5).
// (src/ops/fully_connected.rs:24-82; rows of all inferences form one [M][K] matrix)
//
//   Y[m][n] = requant( sum_k X[m][k] * W[n][k]  - wzp * rowsum(X[m])  + (c3 - c2[n]) )
//
// Both operands are K-contiguous ("NT" GEMM), the natural layout for
// v_mfma_i32_32x32x32_i8 whose lanes each hold 16 consecutive k-bytes of one row.
//   tile     : 256 x 256 per workgroup, 8 waves as 2 (m) x 4 (n), each wave 128 x 64 = 4 x 2
//              MFMA tiles (128 accumulator VGPRs); 128 x 128 with 4 waves when the problem has
//              too few 256^2 tiles to fill the chip.
//   staging  : BK = 128 bytes per step; X and W tiles go HBM/L2 -> LDS by LDS-DMA
//              (global_load_lds_dwordx4), double buffered.
//   schedule : 256^2 tile (one workgroup per CU, two waves per SIMD): the two wave rows (wm = 0 / 1,
//              one wave of each per SIMD) run ONE BARRIER apart, so one row's MFMA section
//              coincides with the other row's ds_read section instead of both stalling on the
//              LDS at once; the DMAs of the next tile are issued between the MFMAs.  Details
//              and the hazard argument are at the loop.  Measured on 4096^3 (scripts/ubench/
//              gemm_i8.hip): +10 % with random operands, +18 % with zero operands over the
//              lockstep loop (vmcnt(0) + barrier per step), which the 128^2 tile keeps.
//              The gap between zero and random operands (3.0 vs 2.2 POP/s) is the chip's power
//              management, not the schedule: a bare MFMA loop issues at 4.5 POP/s.
//   LDS image: [row][128 B], 16-byte slot index XOR ((row >> 1) & 7) -- with that key the 16
//              lanes of every ds_read_b128 service group ({0-3,12-15,20-27}, ...) hit 16
//              distinct 16-byte bank slots (row & 7 would be 2-way).  A DMA writes LDS linearly
//              (base + lane*16), so the swizzle is applied to the GLOBAL source chunk each
//              lane fetches and again on the fragment reads (guide rule 21: both sides).
//   operands : MFMA "A" = W rows (n), MFMA "B" = X rows (m), so that D[n][m] leaves every
//              lane with 16 results of ONE output row m.  The lane -> W-row map is permuted
//              (rho -> 16*((rho>>2)&1) + 4*(rho>>3) + (rho&3)) so those 16 results are 16
//              CONSECUTIVE n: one packed 16-byte store per lane per tile, no transposition.
//   grid     : XCD-aware remap so the 8 tiles that share panels sit behind the same L2.
// The f32 epilogue is the reference's, fused; |acc| can exceed 2^24 at K = 4096, where
// f32(acc) rounds to nearest even exactly like Rust's `as f32`.
// ------------------------------------------------------------------------
typedef int v16i __attribute__((ext_vector_type(16)));

// RS: the weight zero point term x1 = wzp * sum_k x[m][k] (fully_connected.rs:60-64) is formed INSIDE the GEMM: the X fragments
// a wave feeds the matrix pipe are the rows whose sums are needed, so wave (wm, wn) adds up ITS share of them (row block
// mt == wn: the WN waves of a wave row hold the same X fragments) with v_dot4 against ones, issued between the MFMAs
// (the VALU is idle there); the sums meet in LDS after the last k step.  No row-sum pre-pass, no extra launch.
// RSP: the same term from row sums formed in this launch's PROLOGUE instead of a pre-pass launch (the pre-pass costs 6 - 8 % of a
// 4096^3 step for a 16 MB read: a kernel boundary and a launch that cannot fill the chip).  The tiles_n workgroups of a tile row each
// sum 256 / tiles_n of its rows while their first k tile's DMAs fly, publish them with system-scope stores and count themselves in
// rs_sync[2 tm]; ~60 us later the epilogue polls that counter -- bounded: a workgroup that does not see it complete (the grid was not
// co-resident: nothing else guarantees that the producers have even started) sums its tile's rows itself -- and the last reader of
// a tile row zeroes the pair of counters for the next launch.  The operand X is read-only for the launch, so the fallback needs no
// synchronisation at all.
#ifndef MF_FC_RSP_POLLS
#define MF_FC_RSP_POLLS 256
#endif
template <int BM, int BN, int WM, int WN, bool STAGGER, bool RS, bool RSP = false>
__global__ __launch_bounds__(64 * WM * WN) void fc_mfma(const int8_t *__restrict__ X,
                                                        int8_t *__restrict__ Y, FcGemmArgs p) {
    static_assert(!(RS && RSP), "one way of forming the row sums");
    constexpr int BK = 128;
    constexpr int NW = WM * WN;                              // waves per workgroup
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;      // 32x32 MFMA tiles per wave
    constexpr int XT = BM * BK, WT = BN * BK, BUF = XT + WT; // one staging buffer (two allocated)
    constexpr int XP = XT / 1024 / NW, WP = WT / 1024 / NW;  // 1 KiB DMA pieces per wave
    static_assert(!RS || MT == WN, "row sums: wave (wm, wn) owns row block mt == wn");
    static_assert(XT % (1024 * NW) == 0 && WT % (1024 * NW) == 0, "DMA pieces must divide over the waves");
    // swizzle key: with (row >> 1) & 7 the 16 lanes of every ds_read_b128 service group
    // ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte bank slots (row & 7 would be 2-way)
    auto key = [](int row) { return (row >> 1) & 7; };
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware 2-D tile order.  Dispatch puts block b on XCD b % 8 and each XCD runs its
    // blocks in order, one residency-full at a time.  Those PM*PN co-resident tiles are
    // mapped to a PM x PN patch, which needs only PM X-panels + PN W-panels per k-step
    // through that XCD's L2 instead of ~1 + PM*PN for a row-major order.
    constexpr int PM = (BM == 128) ? 8 : 4, PN = 8;           // 64 resident 128^2 tiles, 32 256^2 tiles
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN; // a ragged last row tile re-reads row M-1 and stores nothing for it
    int tm, tn;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;   // j-th block of this XCD
        const int patches_n = tiles_n / PN, npatch = (tiles_m / PM) * patches_n;
        if (tiles_m % PM == 0 && tiles_n % PN == 0 && (npatch & 7) == 0) {
            const int patch = xcd * (npatch >> 3) + j / (PM * PN), t = j % (PM * PN);
            tm = (patch / patches_n) * PM + t / PN;
            tn = (patch % patches_n) * PN + t % PN;
        } else {
            tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
        }
    }
    const int K = p.K;
    const int8_t *Wt = p.w + (size_t)tn * BN * K;
    // row `row` of this workgroup's X tile: rows past the matrix (ragged M) alias the last row -- their products are
    // computed and dropped
    auto xrow = [&](int row) {
        const int g = tm * BM + row;
        return X + (size_t)(g < p.M ? g : p.M - 1) * K;
    };

    // DMA piece i (1 KiB) of a tile = rows 8i .. 8i+7; lane -> (row, swizzled 16-byte slot)
    auto stage = [&](int kt, int buf) {
        const int r8 = lane >> 3, s8 = lane & 7;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int i = wave * XP + j, row = 8 * i + r8;
            dma16(xrow(row) + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + i * 1024);
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int i = wave * WP + j, row = 8 * i + r8;
            dma16(Wt + (size_t)row * K + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + XT + i * 1024);
        }
    };

    v16i acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

    // fragment rows of this lane
    const int rho = lane & 31, half = lane >> 5;
    const int nloc = 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    int xoff[MT], woff[NT], xkey[MT], wkey[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int row = wm * (BM / WM) + t * 32 + rho;
        xoff[t] = row * BK, xkey[t] = key(row);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int row = wn * (BN / WN) + t * 32 + nloc;
        woff[t] = XT + row * BK, wkey[t] = key(row);
    }

    auto load_frags = [&](const uint8_t *lb, int ks, v4i (&a)[NT], v4i (&b)[MT]) {
#pragma unroll
        for (int t = 0; t < MT; ++t) b[t] = *(const v4i *)(lb + xoff[t] + (((ks * 2 + half) ^ xkey[t]) << 4));
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *(const v4i *)(lb + woff[t] + (((ks * 2 + half) ^ wkey[t]) << 4));
    };

    const int nk = K / BK;
    // RSP: sum_k x[m][k] of `nrows` rows from `row0` on (rows past the matrix: skipped), one wave per row, into dst (device memory
    // with system-scope stores, or LDS)
    auto rowsums = [&](int row0, int nrows, int *dst, bool to_lds) {
        for (int r = wave; r < nrows; r += NW) {
            const int m = row0 + r;
            if (m >= p.M) continue;
            const uint4 *x = (const uint4 *)(X + (size_t)m * K);
            int acc = 0;
            for (int k = lane; k < (K >> 4); k += 64) {
                const uint4 v = x[k];
                acc = sdot4(v.x, 0x01010101u, acc), acc = sdot4(v.y, 0x01010101u, acc);
                acc = sdot4(v.z, 0x01010101u, acc), acc = sdot4(v.w, 0x01010101u, acc);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) {
                if (to_lds) dst[m - row0] = acc;
                else __hip_atomic_store(dst + m, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    // The prologue form of it, split so that nothing of it is on the critical path: the loads are issued in FRONT of the first k
    // tile's DMAs (rs_issue: up to 8 x 16 bytes per lane in registers) and return under the same wait the first tile needs anyway;
    // the sums are formed and stored behind that wait (rs_store); the count that publishes them is taken one k tile later, behind
    // that tile's own vmcnt(0) + barrier, when every wave's stores have been acknowledged (rs_count).  Shapes with more than 8
    // loads per lane, or a single k tile, do all three at once up front (rs_sync_all).
    constexpr int RSV = 8;
    const int rpt = RSP ? BM / tiles_n : 0;             // rows of this tile row that this workgroup sums (launcher: tiles_n divides BM)
    const int rs_rows = (rpt + NW - 1) / NW, rs_chunks = (K / 16 + 63) / 64;
    const bool rs_split = RSP && rs_rows * rs_chunks <= RSV && nk > 1;
    uint4 rsv[RSV];
    auto rs_issue = [&]() {
        if constexpr (RSP) {
            if (!rs_split) return;
#pragma unroll
            for (int i = 0; i < RSV; ++i) {
                const int r = wave + NW * (i / rs_chunks), c = (i % rs_chunks) * 64 + lane, m = tm * BM + tn * rpt + r;
                rsv[i] = make_uint4(0u, 0u, 0u, 0u);
                if (i < rs_rows * rs_chunks && r < rpt && m < p.M && c < (K >> 4)) rsv[i] = ((const uint4 *)(X + (size_t)m * K))[c];
            }
        }
    };
    auto rs_store = [&]() {
        if constexpr (RSP) {
            if (!rs_split) return;
            for (int j = 0; j < rs_rows; ++j) {
                int acc = 0;
#pragma unroll
                for (int i = 0; i < RSV; ++i)
                    if (i / rs_chunks == j) {
                        acc = sdot4(rsv[i].x, 0x01010101u, acc), acc = sdot4(rsv[i].y, 0x01010101u, acc);
                        acc = sdot4(rsv[i].z, 0x01010101u, acc), acc = sdot4(rsv[i].w, 0x01010101u, acc);
                    }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                const int r = wave + NW * j, m = tm * BM + tn * rpt + r;
                if (lane == 0 && r < rpt && m < p.M) __hip_atomic_store(p.rs_sums + m, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    auto rs_count = [&]() {
        if constexpr (RSP) {
            if (tid == 0) __hip_atomic_fetch_add(p.rs_sync + 2 * tm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    auto rs_sync_all = [&]() { // (after stage(0, ..): the first k tile's DMAs fly under the reads)
        if constexpr (RSP) {
            if (rs_split) return;
            rowsums(tm * BM + tn * rpt, rpt, p.rs_sums, false);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's sums have left
            wg_sync();
            rs_count();
        }
    };
    int rs = 0; // RS: this lane's share of sum_k x[row][k], row = row block wn of wave row wm, fragment row rho
    if constexpr (!STAGGER) {
        // lockstep loop: the DMAs of step t+1 fly during the MFMAs of step t; one vmcnt(0) +
        // barrier per step
        int cur = 0;
        rs_issue();
        stage(0, 0);
        rs_sync_all();
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (RSP && rs_split && kt == 0) rs_store();
            wg_sync();
            if (RSP && rs_split && kt == 1) rs_count(); // (every wave has waited for its stores: the vmcnt(0) above)
            if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
            const uint8_t *lb = lds + cur * BUF;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                v4i a[NT], b[MT];
                load_frags(lb, ks, a, b);
                v4i bs = b[0];
                if constexpr (RS) {
#pragma unroll
                    for (int mt = 1; mt < MT; ++mt)
                        if (wn == mt) bs = b[mt];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
                        if constexpr (RS) {
                            if (nt * MT + mt < 4) rs = sdot4((uint32_t)bs[nt * MT + mt], 0x01010101u, rs);
                        }
                    }
            }
        }
        if constexpr (RS) wg_sync(); // every wave is done with the staging buffers: the row sums go there
    } else {
        // Staggered wave rows.  Per tile kt every wave runs, in program order,
        //     L0: ds_read the fragments of k-substeps 0,1 of buffer cur
        //     P0  (barrier) ; lgkmcnt(0)
        //     M0: 16 MFMAs, with the 8 DMA pieces of tile kt+1 (-> buffer cur^1) issued between them
        //     P1  (barrier)
        //     L1: ds_read the fragments of k-substeps 2,3 of buffer cur ; vmcnt(0)
        //     P2  (barrier) ; lgkmcnt(0)
        //     M1: 16 MFMAs
        //     P3  (barrier)
        // and wave row 1 executes one extra barrier first, so it is always ONE physical barrier
        // behind row 0: row 0's M sections line up with row 1's L sections and vice versa.
        // Let n be the physical index of row 0's P0(kt); row 1's P0(kt) is n+1.
        //   WAR (DMA of tile kt+1 overwrites tile kt-1's buffer): tile kt-1 is last read in
        //     L1(kt-1) and those reads retire at the lgkmcnt(0) after P2(kt-1) -- physical n-2
        //     for row 0, n-1 for row 1.  DMAs are issued after P0(kt), i.e. after physical n
        //     (row 0) / n+1 (row 1); a wave passes barrier n only once every wave has ARRIVED
        //     at n, and row 1 executes its lgkmcnt(0) between n-1 and its arrival at n.
        //   RAW (tile kt+1 is first read in L0(kt+1), after P3(kt) = physical n+3 / n+4): every
        //     wave waits vmcnt(0) -- its own DMAs have landed in LDS -- before P2(kt), which is
        //     physical n+2 (row 0) / n+3 (row 1); so by the time any wave passes n+3 all
        //     waves' DMAs of tile kt+1 have landed.
        // sched_barrier(0) pins the compiler's instruction order around the barriers.
        constexpr int PIECES = XP + WP, NMF = 2 * NT * MT, GAP = NMF / (PIECES + 1) > 0 ? NMF / (PIECES + 1) : 1;
        static_assert(BK == 128 && WM == 2, "phase plan: 4 k-substeps per tile, two wave rows");
        auto stage_piece = [&](int kt, int buf, int j) {
            const int r8 = lane >> 3, s8 = lane & 7;
            if (j < XP) {
                const int i = wave * XP + j, row = 8 * i + r8;
                dma16(xrow(row) + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + i * 1024);
            } else {
                const int i = wave * WP + (j - XP), row = 8 * i + r8;
                dma16(Wt + (size_t)row * K + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + XT + i * 1024);
            }
        };
        rs_issue();
        stage(0, 0);
        rs_sync_all();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rs_store();
        wg_sync();
        if (wm == 1) __builtin_amdgcn_s_barrier(); // the stagger
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            const uint8_t *lb = lds + cur * BUF;
            const bool more = kt + 1 < nk;
            int piece = 0;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                v4i a[2][NT], b[2][MT];
                load_frags(lb, 2 * ph, a[0], b[0]);
                load_frags(lb, 2 * ph + 1, a[1], b[1]);
                if (more && ph == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // (RSP: behind tile 0's second barrier every wave -- either wave row: they run one barrier apart, and this is row 1's
                // turn one barrier later -- has waited for its row-sum stores with the vmcnt(0) above: wave row 1's lane 0 counts)
                if (RSP && rs_split && kt == 0 && ph == 1 && wave == NW - 1 && lane == 0)
                    __hip_atomic_fetch_add(p.rs_sync + 2 * tm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_sched_barrier(0);
                v4i bs[2] = {b[0][0], b[1][0]};
                if constexpr (RS) {
#pragma unroll
                    for (int mt = 1; mt < MT; ++mt)
                        if (wn == mt) bs[0] = b[0][mt], bs[1] = b[1][mt];
                }
                __builtin_amdgcn_s_setprio(1);
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u][nt], b[u][mt], acc[nt][mt], 0, 0, 0);
                            if constexpr (RS) { // one v_dot4 behind each of the first eight MFMAs of the section
                                if (cnt < 8) rs = sdot4((uint32_t)bs[cnt >> 2][cnt & 3], 0x01010101u, rs);
                            }
                            ++cnt;
                            if (ph == 0 && cnt % GAP == 0 && piece < PIECES) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (more) stage_piece(kt + 1, cur ^ 1, piece);
                                ++piece;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier(); // row 0 absorbs row 1's extra barrier
    }

    // RS: lanes rho and rho + 32 hold the two halves of every k step's 32 bytes; the finished sums meet in LDS (the
    // staging buffers are dead: every wave has passed its last fragment read)
    if constexpr (RS) {
        rs += __shfl_xor(rs, 32, 64);
        if (half == 0) ((int *)lds)[wm * (BM / WM) + wn * 32 + rho] = rs;
        wg_sync();
    }
    if constexpr (RSP) {
        // the tile row's sums: finished when all tiles_n workgroups of the row have counted themselves (they did so ~ a GEMM ago)
        wg_sync(); // every wave has passed its last fragment read: the staging buffers are free for the 256 sums
        int *ls = (int *)lds, *flag = (int *)lds + BM;
        if (tid == 0) {
            int ok = 0;
            for (int poll = 0; poll < MF_FC_RSP_POLLS && !ok; ++poll) {
                ok = __hip_atomic_load(p.rs_sync + 2 * tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= tiles_n;
                if (!ok) __builtin_amdgcn_s_sleep(32);
            }
            *flag = ok;
        }
        wg_sync();
        if (*flag) {
            for (int r = tid; r < BM; r += 64 * NW) {
                const int m = tm * BM + r;
                ls[r] = m < p.M ? __hip_atomic_load(p.rs_sums + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
            }
        } else {
            rowsums(tm * BM, BM, ls, true); // (the producers never showed up: not co-resident -- sum the tile's rows here)
        }
        wg_sync();
        // the last reader of this tile row leaves the counters zero for the next launch (every workgroup of the row has produced by
        // the time it reads: its own prologue precedes its epilogue)
        if (tid == 0 && __hip_atomic_fetch_add(p.rs_sync + 2 * tm + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == tiles_n - 1) {
            __hip_atomic_store(p.rs_sync + 2 * tm, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(p.rs_sync + 2 * tm + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // epilogue: lane (m column = lane & 31, half) holds n = tile + 16*half + r, r = 0..15
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = tn * BN + wn * (BN / WN) + nt * 32 + 16 * half;
        float cA[16];
        int cK[16];
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            const float4 fa = *(const float4 *)(p.A + n0 + r);
            const int4 ik = *(const int4 *)(p.Kc + n0 + r);
            cA[r] = fa.x, cA[r + 1] = fa.y, cA[r + 2] = fa.z, cA[r + 3] = fa.w;
            cK[r] = ik.x, cK[r + 1] = ik.y, cK[r + 2] = ik.z, cK[r + 3] = ik.w;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = tm * BM + wm * (BM / WM) + mt * 32 + rho;
            if (m >= p.M) continue; // ragged last row tile
            int corr = 0; // x1 = wzp * row-sum of the input
            if constexpr (RS || RSP) corr = p.wzp * ((const int *)lds)[wm * (BM / WM) + mt * 32 + rho];
            else if (p.rowsum) corr = p.wzp * p.rowsum[m];
            uint32_t d[4];
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int q0 = requant(acc[nt][mt][r] + cK[r] - corr, cA[r], p.S, p.lo_f, p.hi_f);
                const int q1 = requant(acc[nt][mt][r + 1] + cK[r + 1] - corr, cA[r + 1], p.S, p.lo_f, p.hi_f);
                const int q2 = requant(acc[nt][mt][r + 2] + cK[r + 2] - corr, cA[r + 2], p.S, p.lo_f, p.hi_f);
                const int q3 = requant(acc[nt][mt][r + 3] + cK[r + 3] - corr, cA[r + 3], p.S, p.lo_f, p.hi_f);
                d[r >> 2] = pack4(q0, q1, q2, q3) ^ p.xr4;
            }
            *(uint4 *)(Y + (size_t)m * p.N + n0) = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
}

// sum_k x[row][k] for the weight-zero-point term of FullyConnected (fully_connected.rs:60-64);
// one wave per row, 16-byte loads.  Only launched when wzp != 0.
__global__ __launch_bounds__(256) void fc_rowsum(const int8_t *__restrict__ in, int *__restrict__ rowsum,
                                                 size_t rows, int K) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    for (size_t row = wave; row < rows; row += nwaves) {
        const uint4 *x = (const uint4 *)(in + row * (size_t)K);
        int rs = 0;
        for (int k = lane; k < (K >> 4); k += 64) {
            const uint4 v = x[k];
            rs = sdot4(v.x, 0x01010101u, rs);
            rs = sdot4(v.y, 0x01010101u, rs);
            rs = sdot4(v.z, 0x01010101u, rs);
            rs = sdot4(v.w, 0x01010101u, rs);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
        if (lane == 0) rowsum[row] = rs;
    }
}

// ---- launchers ----
bool launch_fc_rowwave(const int8_t *in, int8_t *out, const FcArgs &a, size_t rows, hipStream_t s) {
    const int grid = grid_for(rows, 4);
    switch (a.N) {
    case 1: hipLaunchKernelGGL(fc_rowwave<1>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    case 2: hipLaunchKernelGGL(fc_rowwave<2>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    case 4: hipLaunchKernelGGL(fc_rowwave<4>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    case 8: hipLaunchKernelGGL(fc_rowwave<8>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    default: return false;
    }
}
bool fc_mfma_supported(size_t rows, int N, int K) {
    // any row count from half a tile up (a ragged last tile is masked); N and K in whole 128s
    return rows >= 64 && N % 128 == 0 && K % 128 == 0 && (rows + 127) / 128 * (size_t)(N / 128) < (1u << 30);
}
void launch_fc_rowsum(const int8_t *in, int *rowsum, size_t rows, int K, hipStream_t s) {
    hipLaunchKernelGGL(fc_rowsum, dim3(grid_for(rows, 4)), dim3(256), 0, s, in, rowsum, rows, K);
}
template <int BM, int BN, int WM, int WN, bool STAGGER, bool RS, bool RSP = false>
static void launch_fc_mfma_t(const int8_t *in, int8_t *out, const FcGemmArgs &a, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 128;
    static LaunchState st;
    (void)prepared(st, fc_mfma<BM, BN, WM, WN, STAGGER, RS, RSP>, 64 * WM * WN, lds);
    const int grid = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL((fc_mfma<BM, BN, WM, WN, STAGGER, RS, RSP>), dim3(grid), dim3(64 * WM * WN), lds, s, in, out, a);
}
// the in-launch row sums: 256 x 256 tiles (the instance that fills the chip), a tile row's rows dealt evenly over its tiles
bool fc_mfma_rowsum_prologue(size_t rows, int N) {
    if (switches().fc_rowsum_fold || switches().fc_tile == 128 || switches().fc_rowsum_prepass) return false;
    const int tiles_n = N / 256;
    return N % 256 == 0 && tiles_n >= 1 && 256 % tiles_n == 0 && (size_t)((rows + 255) / 256) * (size_t)tiles_n >= 192;
}
// The weight zero point term needs sum_k x[m][k].  Default: the separate fc_rowsum launch in front of the GEMM.  MF_FC_ROWSUM_FOLD=1:
// the RS instance forms the sums inside the GEMM (no extra launch, no row-sum buffer) -- measured 3-4 % SLOWER than the pre-pass
// (profiles/r04/fc_rowsum_ab.txt: 75.0-75.7 us against 71.8-73.1 us per 4096^3 step; weight zero point 0: 67.8), so it is the switch,
// not the default.
bool fc_mfma_rowsum_prepass() {
    const bool fold = switches().fc_rowsum_fold;
    return !fold;
}
void launch_fc_mfma(const int8_t *in, int8_t *out, const FcGemmArgs &a, hipStream_t s) {
    const int force = switches().fc_tile;
    // 256 x 256 tiles halve the L2 -> LDS traffic per MAC; they need >= 256 tiles to fill the chip
    const bool big = a.N % 256 == 0 && (size_t)((a.M + 255) / 256) * (a.N / 256) >= 192;
    const bool rs = a.wzp != 0 && !a.rowsum; // the weight zero point term from in-kernel row sums
    if (a.rs_sums && a.rs_sync && a.wzp != 0) { // (the caller asked fc_mfma_rowsum_prologue first)
        launch_fc_mfma_t<256, 256, 2, 4, true, false, true>(in, out, a, s);
        return;
    }
    if ((big && force != 128) || (force == 256 && a.N % 256 == 0)) {
        if (rs) launch_fc_mfma_t<256, 256, 2, 4, true, true>(in, out, a, s);
        else launch_fc_mfma_t<256, 256, 2, 4, true, false>(in, out, a, s);
    } else {
        if (rs) launch_fc_mfma_t<128, 128, 2, 2, false, true>(in, out, a, s);
        else launch_fc_mfma_t<128, 128, 2, 2, false, false>(in, out, a, s);
    }
}
bool launch_fc_rowwave_softmax(const int8_t *in, int8_t *out, const FcArgs &a, const SoftmaxArgs &sm, size_t rows,
                               hipStream_t s) {
    const int grid = grid_for(rows, 4);
    switch (a.N) {
    case 2: hipLaunchKernelGGL(fc_rowwave_softmax<2>, dim3(grid), dim3(256), 0, s, in, out, a, sm, rows); return true;
    case 4: hipLaunchKernelGGL(fc_rowwave_softmax<4>, dim3(grid), dim3(256), 0, s, in, out, a, sm, rows); return true;
    case 8: hipLaunchKernelGGL(fc_rowwave_softmax<8>, dim3(grid), dim3(256), 0, s, in, out, a, sm, rows); return true;
    default: return false;
    }
}

} // namespace k
} // namespace mf
