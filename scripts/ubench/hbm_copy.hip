// Microbenchmark: what a plain streaming kernel reaches on this chip, at the tensor sizes of the layer-wise operators
// (0.3 .. 3.6 GB per launch) and at their read : write ratios (1:1 pointwise K = N, 2:1 / 4:1 stride-2 depthwise,
// 1:2 pointwise N = 2K).  16 bytes per lane, whole contiguous KiB per wave instruction, U loads in flight per lane --
// i.e. the access pattern of pw_mfma / dw3x3_* without any arithmetic.  The best rate over the grid sweep is the
// practical ceiling the layer-wise kernels' GB/s should be read against (the 8 TB/s pin-rate peak is not reachable by
// any kernel).   Build: hipcc --offload-arch=gfx950 -O3 hbm_copy.hip -o hbm_copy
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

// RD 16-byte reads per WR 16-byte writes (per lane and iteration); NT = non-temporal stores
template <int RD, int WR, int U, bool NT>
__global__ __launch_bounds__(256) void stream(const v4i *__restrict__ in, v4i *__restrict__ out, long long nvec_out) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < nvec_out / WR; i0 += stride * U) {
        v4i r[U][RD];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = std::min(i0 + u * stride, nvec_out / WR - 1);
#pragma unroll
            for (int k = 0; k < RD; ++k) r[u][k] = in[(i / 64 * RD + k) * 64 + (i & 63)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i < nvec_out / WR) {
                v4i s = r[u][0];
#pragma unroll
                for (int k = 1; k < RD; ++k) s ^= r[u][k];
#pragma unroll
                for (int k = 0; k < WR; ++k) {
                    v4i *dst = &out[(i / 64 * WR + k) * 64 + (i & 63)];
                    if (NT) __builtin_nontemporal_store(s + k, dst);
                    else *dst = s + k;
                }
            }
        }
    }
}

template <int RD, int WR, int U, bool NT>
static double best(const v4i *in, v4i *out, double out_bytes, int *best_grid) {
    const long long nvec_out = (long long)(out_bytes / 16);
    double best_gbs = 0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int mult : {2, 4, 8, 16, 32, 64}) {
        const int grid = 256 * mult;
        std::vector<float> t;
        for (int r = 0; r < 7; ++r) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((stream<RD, WR, U, NT>), dim3(grid), dim3(256), 0, 0, in, out, nvec_out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double bytes = out_bytes * (double)(RD + WR) / WR;
        const double gbs = bytes / (t[t.size() / 2] * 1e-3) / 1e9;
        if (gbs > best_gbs) best_gbs = gbs, *best_grid = grid;
    }
    return best_gbs;
}

int main() {
    const size_t cap = (size_t)4 << 30;
    v4i *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, cap) != hipSuccess || hipMalloc(&out, cap) != hipSuccess) return 1;
    (void)hipMemset(in, 1, cap);
    (void)hipMemset(out, 0, cap);
    std::printf("%-26s %10s %10s %10s %10s %10s   (GB/s moved, median of 5, best grid)\n", "total bytes per launch ->", "0.3 GB", "0.6 GB", "1.2 GB",
                "2.4 GB", "3.6 GB");
    const double totals[5] = {0.3e9, 0.6e9, 1.2e9, 2.4e9, 3.6e9};
#define ROW(name, RD, WR, U, NT)                                                            \
    do {                                                                                    \
        std::printf("%-26s", name);                                                         \
        for (double tot : totals) {                                                         \
            int g = 0;                                                                      \
            const double ob = tot * WR / (RD + WR);                                         \
            std::printf(" %6.0f@%-4d", best<RD, WR, U, NT>(in, out, ob, &g), g / 256);      \
        }                                                                                   \
        std::printf("\n");                                                                  \
    } while (0)
    ROW("copy 1:1  U=4", 1, 1, 4, false);
    ROW("copy 1:1  U=8", 1, 1, 8, false);
    ROW("copy 1:1  U=4 nt-store", 1, 1, 4, true);
    ROW("read 2 : write 1  U=4", 2, 1, 4, false);
    ROW("read 4 : write 1  U=2", 4, 1, 2, false);
    ROW("read 1 : write 2  U=4", 1, 2, 4, false);
    ROW("read 1 : write 2  nt", 1, 2, 4, true);
    ROW("read 8 : write 1  U=2", 8, 1, 2, false);
    return 0;
}
