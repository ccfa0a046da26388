// Probe: v_mfma_f32_4x4x1_16b_f32 as an exact, ORDERED outer-product accumulator.
//   D[blk][i][j] += A[blk][i] * B[blk][j]   (16 blocks of 4x4, K = 1)
// Checks (1) the operand / result layout assumed by the shared-W block reduction: A and B element of lane l belong to block l/4,
// index l%4; D register r of lane l is D[l/4][r][l%4]; (2) that a chain of 64 such instructions with A in {0, 1} equals the
// sequential fp32 chain acc = fma(a, b, acc) bit for bit (including tiny and huge magnitudes).
//   build: hipcc --offload-arch=gfx950 -O2 -o mfma_4x4x1 mfma_4x4x1.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* __restrict__ A /*[64 k][64 lanes]*/, const float* __restrict__ B, float* __restrict__ D /*[4][64]*/, int K) {
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < K; ++kk) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[kk * 64 + l], B[kk * 64 + l], acc, 0, 0, 0);
    D[0 * 64 + l] = acc.x; D[1 * 64 + l] = acc.y; D[2 * 64 + l] = acc.z; D[3 * 64 + l] = acc.w;
}

int main() {
    const int K = 64;
    std::vector<float> A(K * 64), B(K * 64), D(4 * 64), R(4 * 64, 0.f);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int kk = 0; kk < K; ++kk)
        for (int l = 0; l < 64; ++l) {
            A[kk * 64 + l] = (rnd() >> 30) == (uint32_t)(l % 4) ? 1.0f : 0.0f;            // one-hot over i = l%4 (varies per block too: fine)
            const int e = (int)(rnd() % 60) - 40;                                          // magnitudes 2^-40 .. 2^19
            B[kk * 64 + l] = std::ldexp(((int)(rnd() >> 8) - (1 << 23)) / 8388608.0f, e);
        }
    // reference: D[blk][i][j] = chain over k of fmaf(A[k][4 blk + i], B[k][4 blk + j], acc); stored as R[i][4 blk + j]
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
                for (int kk = 0; kk < K; ++kk) acc = std::fmaf(A[kk * 64 + 4 * blk + i], B[kk * 64 + 4 * blk + j], acc);
                R[i * 64 + 4 * blk + j] = acc;
            }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t q = 0; q < D.size(); ++q) { uint32_t a, b; memcpy(&a, &D[q], 4); memcpy(&b, &R[q], 4); if (a != b) { if (bad < 5) printf("mismatch at reg %zu lane %zu: %a vs %a\n", q / 64, q % 64, D[q], R[q]); ++bad; } }
    printf("v_mfma_f32_4x4x1_16b_f32: layout + ordered exact accumulation: %s (%d of %zu differ)\n", bad ? "MISMATCH" : "OK", bad, D.size());
    return bad != 0;
}
