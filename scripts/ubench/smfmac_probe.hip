// What v_smfmac_i32_16x16x128_i8 computes on gfx950, found by experiment: for a single non-zero stored element of the sparse operand A
// (lane group g, stored byte s, index register value X) and a dense B whose element (k, n) is bit n of k, the result row tells which
// dense K position that stored element multiplies.  Prints k_eff for every (g, s) under a set of index values, and checks the
// hypothesis   k_eff = 32 g + 4 (s / 2) + ((X >> (4 (s / 2) + 2 (s & 1))) & 3).
//     hipcc --offload-arch=gfx950 -O2 smfmac_probe.hip -o smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

// one wave per experiment e = (g, s, xi): out[e][n] = D[row 0][n]; out[e][16 + n] = D[row 5][n] (a second row, must be zero)
__global__ void probe(const uint32_t *xs, int nx, int *out) {
    const int e = blockIdx.x, lane = threadIdx.x;
    const int xi = e % nx, s = (e / nx) % 16, g = e / nx / 16;
    const int col = lane & 15, lg = lane >> 4;
    // A: stored byte s of lane (row 0, group g) = 1
    uint8_t a[16] = {};
    if (col == 0 && lg == g) a[s] = 1;
    v4i A;
    __builtin_memcpy(&A, a, 16);
    // B: lane (col n, group lg) holds dense K = 32 lg + b, b = 0..31: value = bit n of k (n < 7), else 0
    uint8_t b[32];
    for (int i = 0; i < 32; ++i) {
        const int k = 32 * lg + i;
        b[i] = col < 7 ? (uint8_t)((k >> col) & 1) : (col == 7 ? 1 : 0); // column 7: all ones (tells that exactly one product happened)
    }
    v8i B;
    __builtin_memcpy(&B, b, 32);
    v4i C = {0, 0, 0, 0};
    C = __builtin_amdgcn_smfmac_i32_16x16x128_i8(A, B, C, (int)xs[xi], 0, 0);
    // D layout: lane (col n, group lg) holds rows 4 lg .. 4 lg + 3
    if (lg == 0) out[e * 32 + col] = C[0];      // row 0
    if (lg == 1) out[e * 32 + 16 + col] = C[1]; // row 5
}
int main() {
    std::vector<uint32_t> xs = {0x00000000u, 0x55555555u, 0xAAAAAAAAu, 0xFFFFFFFFu, 0x44444444u /* (0,1) per group */, 0xE4E4E4E4u /* fields 0,1,2,3 */,
                                0x1B1B1B1Bu /* fields 3,2,1,0 */, 0x88888888u /* (0,2) */, 0xCCCCCCCCu /* (0,3) */, 0x99999999u /* (1,2) */, 0x11111111u /* (1,0) */};
    // plus: only one 2-bit field set to 3
    for (int f = 0; f < 16; ++f) xs.push_back(3u << (2 * f));
    const int nx = (int)xs.size(), ne = 4 * 16 * nx;
    uint32_t *dx;
    int *dout;
    hipMalloc(&dx, nx * 4), hipMalloc(&dout, ne * 32 * 4);
    hipMemcpy(dx, xs.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0, ne * 32 * 4);
    hipLaunchKernelGGL(probe, dim3(ne), dim3(64), 0, 0, dx, nx, dout);
    std::vector<int> h(ne * 32);
    hipMemcpy(h.data(), dout, ne * 32 * 4, hipMemcpyDeviceToHost);
    int bad = 0, odd = 0;
    for (int g = 0; g < 4; ++g)
        for (int s = 0; s < 16; ++s) {
            printf("g %d s %2d:", g, s);
            for (int xi = 0; xi < nx; ++xi) {
                const int *r = &h[(((g * 16 + s) * nx) + xi) * 32];
                int k = 0;
                for (int n = 0; n < 7; ++n) k |= (r[n] & 1) << n;
                const int count = r[7];
                int other = 0;
                for (int n = 0; n < 16; ++n) other |= r[16 + n];
                const int want = 32 * g + 4 * (s / 2) + ((xs[xi] >> (4 * (s / 2) + 2 * (s & 1))) & 3);
                if (count != 1 || other) ++odd;
                if (k != want) ++bad;
                if (xi < 11) printf(" %3d%s", k, count == 1 ? "" : (count == 0 ? "(0)" : "(n)"));
            }
            printf("\n");
        }
    printf("index values:");
    for (int xi = 0; xi < 11; ++xi) printf(" %08x", xs[xi]);
    printf("\nhypothesis k = 32 g + 4 (s/2) + field(4 (s/2) + 2 (s&1)): %d of %d experiments differ; %d with a product count != 1 or a stray row\n", bad, ne, odd);
    return 0;
}
