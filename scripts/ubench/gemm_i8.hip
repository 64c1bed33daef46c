// gemm_i8.hip -- staging-pipeline experiments for the FullyConnected int8 MFMA GEMM
// (microflow_rs_amd/csrc/k_gemm.hip: fc_mfma).  Same math as the product kernel
// (NT GEMM, i32 accumulate, f32 requantize epilogue, packed 16-byte stores); what varies is
// the K-step, the number of LDS buffers and the synchronisation of the LDS-DMA pipeline.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 \
//         scripts/ubench/gemm_i8.hip -o scripts/ubench/gemm_i8 && scripts/ubench/gemm_i8 [M N K]
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

struct Args {
    const int8_t *w;
    const float *A;
    const int *Kc;
    float S, lo_f, hi_f;
    int M, N, K;
};

__device__ __forceinline__ void dma16(const int8_t *src_lane, uint8_t *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void_t *)src_lane, (lds_void_t *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int requant(int acc, float A, float S, float lo_f, float hi_f) {
    const float x = __fadd_rn(A, __fmul_rn(S, (float)acc));
    float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    r = __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
    return (int)r;
}
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0: vmcnt(0) + __syncthreads per step, stage(kt+1) after the barrier (2 buffers).
// MODE 1: raw s_barrier, counted vmcnt: NBUF-1 tiles staged ahead, NBUF-2 still in flight
//         across the barrier.
template <int BM, int BN, int WM, int WN, int BK, int NBUF, int MODE>
__global__ __launch_bounds__(64 * WM * WN) void gemm(const int8_t *__restrict__ X, int8_t *__restrict__ Y, Args p) {
    constexpr int NW = WM * WN;
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    constexpr int XT = BM * BK, WT = BN * BK, BUF = XT + WT;
    constexpr int XP = XT / 1024 / NW, WP = WT / 1024 / NW;
    constexpr int SLOTS = BK / 16, RPP = 1024 / BK; // 16-byte slots per row, rows per 1 KiB piece
    static_assert(XT % (1024 * NW) == 0 && WT % (1024 * NW) == 0, "pieces");
    auto key = [](int row) { return BK == 128 ? (row >> 1) & 7 : (row >> 2) & 3; };
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    constexpr int PM = (BM == 128) ? 8 : 4, PN = 8;
    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    int tm, tn;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
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
    const int8_t *Xt = X + (size_t)tm * BM * K;
    const int8_t *Wt = p.w + (size_t)tn * BN * K;

    auto stage = [&](int kt, int buf) {
        const int rr = lane / SLOTS, ss = lane % SLOTS;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int i = wave * XP + j, row = RPP * i + rr;
            dma16(Xt + (size_t)row * K + (size_t)kt * BK + ((ss ^ key(row)) << 4), lds + buf * BUF + i * 1024);
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int i = wave * WP + j, row = RPP * i + rr;
            dma16(Wt + (size_t)row * K + (size_t)kt * BK + ((ss ^ key(row)) << 4), lds + buf * BUF + XT + i * 1024);
        }
    };

    v16i acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

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

    auto compute = [&](const uint8_t *lb) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            v4i a[NT], b[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) b[t] = *(const v4i *)(lb + xoff[t] + (((ks * 2 + half) ^ xkey[t]) << 4));
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = *(const v4i *)(lb + woff[t] + (((ks * 2 + half) ^ wkey[t]) << 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
        }
    };

    // one DMA piece (X for j < XP, W after) of tile kt into buffer buf
    auto stage_piece = [&](int kt, int buf, int j) {
        const int rr = lane / SLOTS, ss = lane % SLOTS;
        if (j < XP) {
            const int i = wave * XP + j, row = RPP * i + rr;
            dma16(Xt + (size_t)row * K + (size_t)kt * BK + ((ss ^ key(row)) << 4), lds + buf * BUF + i * 1024);
        } else {
            const int i = wave * WP + (j - XP), row = RPP * i + rr;
            dma16(Wt + (size_t)row * K + (size_t)kt * BK + ((ss ^ key(row)) << 4), lds + buf * BUF + XT + i * 1024);
        }
    };
    // compute tile in `lb` while issuing the DMA pieces of tile kt_next between MFMA groups
    auto compute_staging = [&](const uint8_t *lb, int kt_next, int buf_next, bool do_stage) {
        constexpr int PIECES = XP + WP, KS = BK / 32;
        constexpr int PER = (PIECES + KS * NT - 1) / (KS * NT); // pieces after each nt-group of MFMAs
        int piece = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            v4i a[NT], b[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) b[t] = *(const v4i *)(lb + xoff[t] + (((ks * 2 + half) ^ xkey[t]) << 4));
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = *(const v4i *)(lb + woff[t] + (((ks * 2 + half) ^ wkey[t]) << 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (do_stage) {
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        if (piece + q < PIECES) stage_piece(kt_next, buf_next, piece + q);
                }
                piece += PER;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // MODE 3: fragments double-buffered in registers: all ds_reads of k-substep ks+1 are issued
    // BEFORE the MFMAs of ks (pinned with sched_barrier), so only the first substep of a tile
    // exposes LDS latency.
    auto load_frags = [&](const uint8_t *lb, int ks, v4i (&a)[NT], v4i (&b)[MT]) {
#pragma unroll
        for (int t = 0; t < MT; ++t) b[t] = *(const v4i *)(lb + xoff[t] + (((ks * 2 + half) ^ xkey[t]) << 4));
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *(const v4i *)(lb + woff[t] + (((ks * 2 + half) ^ wkey[t]) << 4));
    };
    auto mfmas = [&](const v4i (&a)[NT], const v4i (&b)[MT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
    };
    auto compute_db = [&](const uint8_t *lb) {
        constexpr int KS = BK / 32;
        v4i a0[NT], b0[MT], a1[NT], b1[MT];
        load_frags(lb, 0, a0, b0);
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            if (ks + 1 < KS) load_frags(lb, ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < KS) load_frags(lb, ks + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KS) mfmas(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nk = K / BK;
    if constexpr (MODE == 7) {
        // FOUR waves (one per SIMD), each a 128 x 128 register tile (256 accumulator VGPRs): no second wave to hide
        // behind, so everything else is issued in the shadow of the wave's own MFMAs, in program order: between two
        // MFMAs of sub-step ks sits one ds_read of sub-step ks+1 (fragments double-buffered in registers, across
        // tile boundaries too) or one LDS-DMA piece of a later tile.  One barrier per K-tile, B(kt), at the start of
        // the tile's LAST sub-step: by then every wave has read all of tile kt (its last fragments were loaded during
        // the previous sub-step), so (a) buffer kt & 1 is free for the DMAs of tile kt+2 and (b) with vmcnt(0) in front
        // of it tile kt+1 -- issued one tile time earlier -- is complete for everyone, and its first fragments can be
        // loaded during this last sub-step.
        constexpr int KS = BK / 32, PIECES = XP + WP, HALF = PIECES / 2;
        static_assert(KS == 4 && NBUF == 2 && MT * NT == 16 && PIECES <= 16, "MODE 7 plan");
        v4i fa[2][NT], fb[2][MT];
        // DMA == 0: no pieces; 1 / 2: first / second half of tile kt_dma's pieces.  Nothing in a slot is conditional
        // at run time (a tile index past the end is clamped: the re-staged buffer is never read again).
        auto slot = [&](const v4i (&a)[NT], const v4i (&b)[MT], v4i (&an)[NT], v4i (&bn)[MT], const uint8_t *lbn, int ksn,
                        int kt_dma, int buf_dma, auto dma_tag) {
            constexpr int DMA = decltype(dma_tag)::value;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = nt * MT + mt;
                    acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < MT) {
                        bn[m] = *(const v4i *)(lbn + xoff[m] + (((ksn * 2 + half) ^ xkey[m]) << 4));
                    } else if (m < MT + NT) {
                        an[m - MT] = *(const v4i *)(lbn + woff[m - MT] + (((ksn * 2 + half) ^ wkey[m - MT]) << 4));
                    } else if (DMA != 0 && m - MT - NT < HALF) {
                        stage_piece(kt_dma, buf_dma, (DMA - 1) * HALF + m - MT - NT);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        using std::integral_constant;
        const int last = nk - 1;
        stage(0, 0);
#pragma unroll
        for (int q = 0; q < HALF; ++q) stage_piece(1 < nk ? 1 : last, 1, q);
        wait_vm<HALF>();
        __builtin_amdgcn_s_barrier();
        load_frags(lds, 0, fa[0], fb[0]);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const uint8_t *lb = lds + cur * BUF, *lo = lds + (cur ^ 1) * BUF;
            const int k1 = kt + 1 < nk ? kt + 1 : last, k2 = kt + 2 < nk ? kt + 2 : last;
            // tile kt+1: second half of its pieces (the first half went out in the previous tile's last sub-step)
            slot(fa[0], fb[0], fa[1], fb[1], lb, 1, k1, cur ^ 1, integral_constant<int, 2>{});
            slot(fa[1], fb[1], fa[0], fb[0], lb, 2, 0, 0, integral_constant<int, 0>{});
            slot(fa[0], fb[0], fa[1], fb[1], lb, 3, 0, 0, integral_constant<int, 0>{});
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // B(kt)
            __builtin_amdgcn_sched_barrier(0);
            slot(fa[1], fb[1], fa[0], fb[0], lo, 0, k2, cur, integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 5 || MODE == 6) {
        constexpr int KS = BK / 32, PIECES = XP + WP;
        constexpr int SUB = MODE == 5 ? 1 : 2;       // k-substeps per phase
        constexpr int PH = KS / SUB;                 // phases per tile
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if ((wave >> 2) == 1) __builtin_amdgcn_s_barrier(); // stagger
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            const uint8_t *lb = lds + cur * BUF;
            const bool more = kt + 1 < nk;
            int piece = 0;
#pragma unroll
            for (int ph = 0; ph < PH; ++ph) {
                v4i a[SUB][NT], b[SUB][MT];
#pragma unroll
                for (int u = 0; u < SUB; ++u) load_frags(lb, ph * SUB + u, a[u], b[u]);
                if (more && ph == PH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                // DMA pieces of tile kt+1 spread between the MFMAs of every phase but the last
                constexpr int NMF = SUB * NT * MT;
                constexpr int PER_PHASE = (PIECES + PH - 2) / (PH - 1);
                constexpr int GAP = NMF / (PER_PHASE + 1) > 0 ? NMF / (PER_PHASE + 1) : 1;
                int issued = 0, cnt = 0;
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u][nt], b[u][mt], acc[nt][mt], 0, 0, 0);
                            ++cnt;
                            if (ph < PH - 1 && cnt % GAP == 0 && issued < PER_PHASE) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (more && piece < PIECES) stage_piece(kt + 1, cur ^ 1, piece);
                                ++piece, ++issued;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if ((wave >> 2) == 0) __builtin_amdgcn_s_barrier(); // balance the stagger
    } else if constexpr (MODE == 4) {
        // Two wave groups (wm = 0 / 1: one wave of each per SIMD) run ONE BARRIER apart, so that
        // one group's MFMA section coincides with the other group's LDS-read / DMA-issue section.
        // Per tile: KS phases, each = [ds_read frags(ks) (+ DMA pieces of tile kt+1)] b1 [MFMAs] b2.
        // Hazards (P(kt,j) = j-th barrier of tile kt in program order; the groups' program
        // barriers are one physical barrier apart):
        //   WAR  DMAs of tile kt+1 go into the buffer tile kt-1 was read from; that tile's last
        //        reads retire at the lgkmcnt(0) after P(kt-1,6), which both groups have passed
        //        once ANY wave is past P(kt,0)  ->  issue DMAs after P(kt,0) only.
        //   RAW  every wave waits vmcnt(0) before P(kt,6); the lagging group's P(kt,6) is the
        //        leading group's P(kt,7), and tile kt+1 is first read after P(kt,7).
        constexpr int KS = BK / 32, PIECES = XP + WP;
        static_assert(KS == 4, "phase plan written for 4 sub-steps per tile");
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wm == 1) __builtin_amdgcn_s_barrier(); // stagger
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            const uint8_t *lb = lds + cur * BUF;
            const bool more = kt + 1 < nk;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                v4i a[NT], b[MT];
                load_frags(lb, ks, a, b);
                if (more) {
                    // pieces: 3 after phase 0's b1 (below), 3 in phase 1's, 2 in phase 2's load section
                    if (ks == 1) {
#pragma unroll
                        for (int q = 3; q < 6 && q < PIECES; ++q) stage_piece(kt + 1, cur ^ 1, q);
                    } else if (ks == 2) {
#pragma unroll
                        for (int q = 6; q < PIECES; ++q) stage_piece(kt + 1, cur ^ 1, q);
                    } else if (ks == 3) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (more && ks == 0) {
#pragma unroll
                    for (int q = 0; q < 3 && q < PIECES; ++q) stage_piece(kt + 1, cur ^ 1, q);
                }
                __builtin_amdgcn_s_setprio(1);
                mfmas(a, b);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier(); // balance the stagger
    } else if constexpr (MODE == 3) {
        int cur = 0;
        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
            compute_db(lds + cur * BUF);
        }
    } else if constexpr (MODE == 2) {
        int cur = 0;
        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute_staging(lds + cur * BUF, kt + 1, cur ^ 1, kt + 1 < nk);
        }
    } else if constexpr (MODE == 0) {
        int cur = 0;
        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
            compute(lds + cur * BUF);
        }
    } else {
        constexpr int PIECES = XP + WP; // DMA instructions per tile per wave
        constexpr int AHEAD = NBUF - 1; // tiles staged ahead of the one being computed
#pragma unroll
        for (int t = 0; t < AHEAD; ++t)
            if (t < nk) stage(t, t);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt must have landed; the AHEAD-1 younger tiles may stay in flight
            if (kt + AHEAD - 1 < nk) wait_vm<PIECES *(AHEAD - 1)>();
            else wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // buffer (kt-1) % NBUF was read in the previous iteration, which every wave has left
            int nxt = cur + AHEAD;
            if (nxt >= NBUF) nxt -= NBUF;
            if (kt + AHEAD < nk) stage(kt + AHEAD, nxt);
            compute(lds + cur * BUF);
            if (++cur == NBUF) cur = 0;
        }
    }

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
            uint32_t d[4];
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int q0 = requant(acc[nt][mt][r] + cK[r], cA[r], p.S, p.lo_f, p.hi_f);
                const int q1 = requant(acc[nt][mt][r + 1] + cK[r + 1], cA[r + 1], p.S, p.lo_f, p.hi_f);
                const int q2 = requant(acc[nt][mt][r + 2] + cK[r + 2], cA[r + 2], p.S, p.lo_f, p.hi_f);
                const int q3 = requant(acc[nt][mt][r + 3] + cK[r + 3], cA[r + 3], p.S, p.lo_f, p.hi_f);
                d[r >> 2] = pack4(q0, q1, q2, q3);
            }
            *(uint4 *)(Y + (size_t)m * p.N + n0) = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
}

// bare MFMA issue rate: no memory traffic, 4 independent accumulators per wave
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_rate(int iters, int *sink) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, (int)threadIdx.x, 7, 9};
    v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
    v4i d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int i = 0; i < iters; ++i) {
        if constexpr (SHAPE == 32) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d3, 0, 0, 0);
        }
    }
    int r = 0;
    for (int k = 0; k < 16; ++k) r += c0[k] + c1[k] + c2[k] + c3[k];
    for (int k = 0; k < 4; ++k) r += d0[k] + d1[k] + d2[k] + d3[k];
    if (r == 0x7fffffff) sink[0] = r;
}

template <int SHAPE>
static void rate(int waves_per_simd, int *sink) {
    const int iters = 20000, blocks = 256 * waves_per_simd; // 256-thread blocks = 4 waves = one per SIMD
    hipLaunchKernelGGL(mfma_rate<SHAPE>, dim3(blocks), dim3(256), 0, 0, 100, sink);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_rate<SHAPE>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 4 * iters * 4 * (SHAPE == 32 ? 65536.0 : 32768.0);
    printf("bare v_mfma_i32_%s_i8, %d wave(s)/SIMD on 256 CUs: %.1f TOP/s (%.3f ms)\n",
           SHAPE == 32 ? "32x32x32" : "16x16x64", waves_per_simd, ops / (ms * 1e-3) / 1e12, ms);
}

__global__ void ref_gemm(const int8_t *X, int8_t *Y, Args p, int row0, int nrows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = row0 + blockIdx.y;
    if (n >= p.N || blockIdx.y >= (unsigned)nrows) return;
    int acc = 0;
    for (int k = 0; k < p.K; ++k) acc += (int)X[(size_t)m * p.K + k] * (int)p.w[(size_t)n * p.K + k];
    Y[(size_t)blockIdx.y * p.N + n] = (int8_t)requant(acc + p.Kc[n], p.A[n], p.S, p.lo_f, p.hi_f);
}

template <int BM, int BN, int WM, int WN, int BK, int NBUF, int MODE>
static float run(const char *name, const int8_t *X, int8_t *Y, const Args &a, const int8_t *ref_rows, int row0,
                 int nrows, int iters) {
    constexpr int lds = NBUF * (BM + BN) * BK;
    auto kern = gemm<BM, BN, WM, WN, BK, NBUF, MODE>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = (a.M / BM) * (a.N / BN);
    CK(hipMemset(Y, 0, (size_t)a.M * a.N));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, 0, X, Y, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, 0, X, Y, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    std::vector<int8_t> got((size_t)nrows * a.N);
    CK(hipMemcpy(got.data(), Y + (size_t)row0 * a.N, got.size(), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref_rows[i];
    const double tops = 2.0 * a.M * a.N * a.K / (ms * 1e-3) / 1e12;
    printf("%-44s lds %3d KB  %8.4f ms  %7.1f TOP/s  %5.1f%%  %s\n", name, lds / 1024, ms, tops, tops / 5033.0 * 100,
           bad ? "MISMATCH" : "ok");
    fflush(stdout);
    return ms;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const int iters = 200;
    std::vector<int8_t> hx((size_t)M * K), hw((size_t)N * K);
    uint64_t s = 0x1234567;
    auto rnd = [&]() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (int8_t)(s >> 56);
    };
    const int fill = argc > 4 ? atoi(argv[4]) : 0; // 0 random, 1 zeros, 2 small values (-2..1)
    for (auto &v : hx) v = fill == 1 ? 0 : fill == 2 ? (int8_t)(rnd() >> 6) : rnd();
    for (auto &v : hw) v = fill == 1 ? 0 : fill == 2 ? (int8_t)(rnd() >> 6) : rnd();
    printf("operand fill mode %d\n", fill);
    std::vector<float> hA(N);
    std::vector<int> hK(N);
    for (int i = 0; i < N; ++i) hA[i] = 3.0f + (float)(i % 7) * 0.37f, hK[i] = (i * 977) % 4001 - 2000;
    int8_t *dX, *dW, *dY, *dR;
    float *dA;
    int *dK;
    CK(hipMalloc(&dX, hx.size()));
    CK(hipMalloc(&dW, hw.size()));
    CK(hipMalloc(&dY, (size_t)M * N));
    const int row0 = M / 2 - 64, nrows = 256 < M ? 256 : M;
    CK(hipMalloc(&dR, (size_t)nrows * N));
    CK(hipMalloc(&dA, N * 4));
    CK(hipMalloc(&dK, N * 4));
    CK(hipMemcpy(dX, hx.data(), hx.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hw.data(), hw.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, hA.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dK, hK.data(), N * 4, hipMemcpyHostToDevice));
    Args a{dW, dA, dK, 127.0f / (74.0f * 74.0f * 3.0f * sqrtf((float)K)), -128.0f, 127.0f, M, N, K};
    hipLaunchKernelGGL(ref_gemm, dim3((N + 255) / 256, nrows), dim3(256), 0, 0, dX, dR, a, row0 >= 0 ? row0 : 0, nrows);
    std::vector<int8_t> ref((size_t)nrows * N);
    CK(hipMemcpy(ref.data(), dR, ref.size(), hipMemcpyDeviceToHost));
    const int r0 = row0 >= 0 ? row0 : 0;
    {
        int *sink;
        CK(hipMalloc(&sink, 4));
        rate<32>(1, sink), rate<32>(2, sink), rate<16>(1, sink), rate<16>(2, sink);
    }
    printf("int8 GEMM M=%d N=%d K=%d (NT), random operands, fused requantize epilogue\n", M, N, K);
    for (int rep = 0; rep < 1; ++rep) {
    run<256, 256, 2, 2, 128, 2, 7>("256x256 4 waves x 128x128, frags 1 substep ahead", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 2, 128, 2, 0>("256x256 4 waves x 128x128, syncthreads", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 2, 128, 2, 3>("256x256 4 waves x 128x128, frag db within tile", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 0>("256x256 BK128 2buf syncthreads (product)", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 4>("256x256 BK128 2buf staggered wave groups", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 5>("256x256 staggered, DMA inside MFMA sections", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 6>("256x256 staggered, 2 phases x 16 MFMA", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 4, 2, 128, 2, 6>("256x256 staggered 2x16, waves 4x2", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 6>("256x256 staggered, 2 phases x 16 MFMA (again)", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 3>("256x256 BK128 2buf frag double-buffer", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 4, 2, 128, 2, 3>("256x256 BK128 2buf frag db, waves 4x2", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 2>("256x256 BK128 2buf DMA interleaved w/ MFMA", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 4, 2, 128, 2, 2>("256x256 BK128 2buf interleaved, waves 4x2", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 64, 2, 0>("256x256 BK64  2buf syncthreads", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 128, 2, 1>("256x256 BK128 2buf raw barrier", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 64, 3, 1>("256x256 BK64  3buf raw barrier vmcnt(N)", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 2, 4, 64, 4, 1>("256x256 BK64  4buf raw barrier vmcnt(2N)", dX, dY, a, ref.data(), r0, nrows, iters);
    run<256, 256, 4, 2, 64, 4, 1>("256x256 BK64  4buf, waves 4x2", dX, dY, a, ref.data(), r0, nrows, iters);
    run<128, 128, 2, 2, 128, 2, 0>("128x128 BK128 2buf syncthreads", dX, dY, a, ref.data(), r0, nrows, iters);
    run<128, 128, 2, 2, 64, 4, 1>("128x128 BK64  4buf raw barrier", dX, dY, a, ref.data(), r0, nrows, iters);
    run<128, 256, 2, 4, 64, 4, 1>("128x256 BK64  4buf raw barrier (8 waves)", dX, dY, a, ref.data(), r0, nrows, iters);
    }
    return 0;
}
