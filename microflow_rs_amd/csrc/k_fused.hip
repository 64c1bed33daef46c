// k_fused.hip -- fused operator groups: the routing of the DepthwiseConv2D 3x3 + Conv2D 1x1 pairs and the network tail.
//
// (src/ops/depthwise_conv_2d.rs:28-105, src/ops/conv_2d.rs:28-108, src/ops/average_pool_2d.rs:29-66,
// src/ops/softmax.rs:15-27)
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
#include "k_common.hpp"
#include "k_tail.hpp"

namespace mf {
namespace k {

// ------------------------------------------------------------------------
// FAST PATH 3c -- fused network tail: AveragePool2D whose output is 1x1
// (src/ops/average_pool_2d.rs:29-66) -> Conv2D 1x1 with N <= 8 outputs
// (src/ops/conv_2d.rs:28-108) -> [Reshape] -> Softmax over the N values
// (src/ops/softmax.rs:15-27).  person_detect ops 27..30: 2304 bytes in, 2 bytes out.
// One wavefront per inference: lane l owns channels 4l..4l+3 (+256 per extra pass), sums its
// taps with byte-masked sdot4, requantises the pool (an int8 tensor, like the reference's),
// takes its share of the N dot products, butterfly-reduces them across the wave, and lanes
// 0..N-1 finish the head epilogue and the table softmax.  Every intermediate tensor keeps the
// reference's exact arithmetic; they just stay in registers.
// ------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void tail_pool_head_softmax(const int8_t *__restrict__ in,
                                                              int8_t *__restrict__ out, TailArgs p,
                                                              size_t batch) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const size_t img_elems = (size_t)p.H * p.W * p.C;
    for (size_t b = wave; b < batch; b += nwaves) tail_one<N>(in + b * img_elems, out + b * N, p, lane);
}

// ---- launchers ----
// The DepthwiseConv2D 3x3 + Conv2D 1x1 pair kernels live in k_fused_mm.hip (depthwise taps on the matrix pipe): dwpw_rr keeps the
// intermediate tensor in registers (C <= 32), dwpw_mm in LDS (every table shape).  MF_DWPW_IMPL=mm takes dwpw_mm for every pair.
// (Round 1's dwpw3x3 -- taps on the VALU -- was retired in round 5: every table shape has had a matrix-pipe kernel since round 2.)
int dwpw_impl() {
    return switches().dwpw_mm_only ? 1 : 2;
}
const char *dwpw_name(int H, int W, int C, int S, int N) {
    if (dwpw_impl() == 2 && dwpw_rr_name(H, W, C, S, N)) return dwpw_rr_name(H, W, C, S, N);
    return dwpw_mm_name(H, W, C, S, N);
}
bool launch_dwpw(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a,
                 int batch, hipStream_t s) {
    if (dwpw_impl() == 2 && launch_dwpw_rr(H, W, C, S, N, in, out, a, batch, s)) return true;
    return launch_dwpw_mm(H, W, C, S, N, in, out, a, batch, s);
}

bool tail_supported(int C, int N, int ntaps) {
    return C % 4 == 0 && C >= 4 && ntaps >= 1 && ntaps <= 64 && (N == 1 || N == 2 || N == 4 || N == 8);
}
void launch_tail(const int8_t *in, int8_t *out, const TailArgs &a, size_t batch, hipStream_t s) {
    const int grid = grid_for(batch, 4);
    switch (a.N) {
    case 1: hipLaunchKernelGGL(tail_pool_head_softmax<1>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    case 2: hipLaunchKernelGGL(tail_pool_head_softmax<2>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    case 4: hipLaunchKernelGGL(tail_pool_head_softmax<4>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    default: hipLaunchKernelGGL(tail_pool_head_softmax<8>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    }
}

} // namespace k
} // namespace mf
