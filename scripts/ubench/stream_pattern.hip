// Microbenchmark: what the memory system delivers for the ACCESS PATTERN of the single-step streaming kernels (k_step_reg_lm /
// k_step_reg_q4), with no arithmetic at all -- per learner 432 contiguous bytes read (3 columns x 36 floats), then either nothing,
// the 144 bytes of one column, or all 432 bytes written back in place.  Same wave shape as k_step_reg_q4: 16 learners per wave
// (a 6.9 KB contiguous image, 7 coalesced 16-byte loads per lane), 4 waves per block.  And a plain copy for reference.
//   build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int AF = 108, F = 36, LPW = 16, IMG4 = LPW * AF / 4, NLD = (IMG4 + 63) / 64;

// MODE 0: read only; 1: read + write one column (144 B of every 432); 2: read + write everything
template <int MODE>
__global__ __launch_bounds__(256) void k_pattern(float* __restrict__ W, int64_t n, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wbase = ((int64_t)blockIdx.x * 4 + wv) * LPW;
    if (wbase >= n) return;
    f4* __restrict__ img = reinterpret_cast<f4*>(W + wbase * AF);
    f4 ld[NLD];
#pragma unroll
    for (int m = 0; m < NLD; ++m) ld[m] = (lane + 64 * m < IMG4) ? img[lane + 64 * m] : f4{0, 0, 0, 0};
    float acc = 0.0f;
#pragma unroll
    for (int m = 0; m < NLD; ++m) acc += ld[m].x + ld[m].y + ld[m].z + ld[m].w;
    if (MODE == 1) {
        const int q = lane >> 2, b = lane & 3;
        const int a = (int)((wbase + q) % 3);
        if (b == a) {
            f4* col = reinterpret_cast<f4*>(W + (wbase + q) * AF + a * F);
#pragma unroll
            for (int k = 0; k < F / 4; ++k) col[k] = f4{acc, acc, acc, (float)k};
        }
    } else if (MODE == 3 || MODE == 4) {
        // the column, widened to whole 64-byte (MODE 3) / 128-byte (MODE 4) sectors: the quad writes the aligned range in 16-byte pieces
        constexpr int64_t G = MODE == 3 ? 64 : 128;
        const int q = lane >> 2, b = lane & 3;
        const int a = (int)((wbase + q) % 3);
        const int64_t s0 = ((wbase + q) * AF + a * F) * 4, e0 = s0 + F * 4;
        const int64_t s1 = s0 & ~(G - 1), e1 = (e0 + G - 1) & ~(G - 1);
        char* base = reinterpret_cast<char*>(W);
        for (int64_t o = s1 + 16 * b; o < e1 && o < n * AF * 4; o += 64) *reinterpret_cast<f4*>(base + o) = f4{acc, acc, acc, (float)b};
    } else if (MODE == 2) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) if (lane + 64 * m < IMG4) img[lane + 64 * m] = ld[m] + f4{1, 1, 1, 1};
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ src, f4* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const int64_t sizes[] = {65536, 131072, 262144, 1048576, 4194304};
    float* sink; CHECK(hipMalloc(&sink, 64));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int64_t n : sizes) {
        float *W, *W2; const size_t bytes = (size_t)n * AF * 4;
        CHECK(hipMalloc(&W, bytes)); CHECK(hipMalloc(&W2, bytes)); CHECK(hipMemset(W, 0, bytes)); CHECK(hipMemset(W2, 0, bytes));
        const unsigned grid = (unsigned)((n + 63) / 64);
        const int reps = n <= 262144 ? 200 : 40;
        auto timeit = [&](int mode, double& us) -> int {
            for (int r = 0; r < 5; ++r) {
                if (mode == 0) hipLaunchKernelGGL(k_pattern<0>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_pattern<1>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 2) hipLaunchKernelGGL(k_pattern<2>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 4) hipLaunchKernelGGL(k_pattern<3>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 5) hipLaunchKernelGGL(k_pattern<4>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const f4*)W, (f4*)W2, (int64_t)(bytes / 16));
            }
            CHECK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) hipLaunchKernelGGL(k_pattern<0>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_pattern<1>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 2) hipLaunchKernelGGL(k_pattern<2>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 4) hipLaunchKernelGGL(k_pattern<3>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else if (mode == 5) hipLaunchKernelGGL(k_pattern<4>, dim3(grid), dim3(256), 0, 0, W, n, sink);
                else hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const f4*)W, (f4*)W2, (int64_t)(bytes / 16));
            }
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            us = ms * 1e3 / reps;
            return 0;
        };
        const char* names[] = {"read 432 B/learner", "read 432 + write 144 (one column)", "read 432 + write 432 (in place)", "copy (read + write 432 to another buffer)",
                               "read 432 + write the column as whole 64-B sectors", "read 432 + write the column as whole 128-B lines"};
        const double moved[] = {432.0, 576.0, 864.0, 864.0, 576.0, 576.0};       // sectors: counted as the 144 useful bytes
        for (int mode = 0; mode < 6; ++mode) {
            double us; if (timeit(mode, us)) return 1;
            printf("learners %8lld (W %7.1f MB)  %-44s %9.2f us per launch  %6.2f TB/s\n", (long long)n, bytes / 1e6, names[mode], us, moved[mode] * n / us / 1e6);
        }
        CHECK(hipFree(W)); CHECK(hipFree(W2));
    }
    return 0;
}
