// v_smfmac_i32_16x16x128_i8 against a host model of the mapping smfmac_probe.hip found, with RANDOM stored values, per-lane random
// index registers and random B, all rows -- the single-element probe leaves open whether a row's indices come from its own lane.
//     hipcc --offload-arch=gfx950 -O2 smfmac_check.hip -o smfmac_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
__global__ void k(const v4i *A, const v8i *B, const int *idx, const v4i *C, v4i *D) {
    const int l = threadIdx.x + 64 * blockIdx.x;
    D[l] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(A[l], B[l], C[l], idx[l], 0, 0);
}
int main() {
    const int NT = 64; // trials
    std::vector<int8_t> A(NT * 64 * 16), B(NT * 64 * 32);
    std::vector<uint32_t> idx(NT * 64);
    std::vector<int> C(NT * 64 * 4), D(NT * 64 * 4);
    srand(7);
    for (auto &v : A) v = (int8_t)(rand() % 256 - 128);
    for (auto &v : B) v = (int8_t)(rand() % 256 - 128);
    for (auto &v : C) v = rand() % 2001 - 1000;
    for (int t = 0; t < NT; ++t)
        for (int l = 0; l < 64; ++l) {
            uint32_t x = 0;
            for (int grp = 0; grp < 8; ++grp) { // two DISTINCT positions per group; trials >= 32: any order, below: ascending
                int p0 = rand() % 4, p1 = rand() % 4;
                while (p1 == p0) p1 = rand() % 4;
                if (t < 32 && p0 > p1) std::swap(p0, p1);
                x |= (uint32_t)(p0 | (p1 << 2)) << (4 * grp);
            }
            idx[t * 64 + l] = x;
        }
    void *dA, *dB, *dI, *dC, *dD;
    hipMalloc(&dA, A.size()), hipMalloc(&dB, B.size()), hipMalloc(&dI, idx.size() * 4), hipMalloc(&dC, C.size() * 4), hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dI, idx.data(), idx.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NT), dim3(64), 0, 0, (const v4i *)dA, (const v8i *)dB, (const int *)dI, (const v4i *)dC, (v4i *)dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad_trials = 0;
    for (int t = 0; t < NT; ++t) {
        long ref[16][16];
        // C / D layout: lane (col n = l & 15, group lg) register i = row 4 lg + i
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) ref[m][n] = C[(t * 64 + (m / 4) * 16 + n) * 4 + (m & 3)];
        for (int la = 0; la < 64; ++la) {
            const int m = la & 15, ga = la >> 4;
            for (int s = 0; s < 16; ++s) {
                const int v = A[(t * 64 + la) * 16 + s];
                const int ha = s >> 3, grp = (s & 7) >> 1, fld = (idx[t * 64 + la] >> (2 * s)) & 3;
                const int lb = 2 * (ga & 1) + ha, half = ga >> 1, byte = 16 * half + 4 * grp + fld;
                for (int n = 0; n < 16; ++n) ref[m][n] += (long)v * B[(t * 64 + lb * 16 + n) * 32 + byte];
            }
        }
        int bad = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) bad += ref[m][n] != D[(t * 64 + (m / 4) * 16 + n) * 4 + (m & 3)];
        if (bad) {
            if (bad_trials < 4) printf("trial %d (%s indices): %d of 256 results differ, e.g. D[0][0] = %d, model %ld\n", t, t < 32 ? "ascending" : "any-order", bad, D[t * 64 * 4], ref[0][0]);
            ++bad_trials;
        }
    }
    printf("%d of %d trials differ from the model (trials 0..31 ascending index pairs, 32..63 any order)\n", bad_trials, NT);
    return 0;
}
