// Microbenchmark: VALU issue rate of gfx950 as a function of waves per SIMD.
// Each wave runs ITER iterations of an unrolled body of N_OPS instructions of one kind.
//   build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;

template <int KIND>
__global__ void k(float* out, int iters, int half_exec) {
    if (half_exec && (threadIdx.x & 32)) return;      // only lanes 0..31 of every wave stay active
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)threadIdx.x * 1e-3f + i;
    float b = 1.0001f, c = 0.5f;
    uint32_t u[4] = {threadIdx.x, 2u, 3u, 4u};
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    f2 pb = f2{b, b}, pc = f2{c, c};
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {            // 16 independent fma chains
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], b, c);
        } else if constexpr (KIND == 1) {     // 1 dependent fma chain
#pragma unroll
            for (int r = 0; r < 64; ++r) a[0] = __builtin_fmaf(a[0], b, c);
        } else if constexpr (KIND == 2) {     // 8 independent packed fma chains (2 fma each)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pb, pc);
        } else if constexpr (KIND == 3) {     // 64-bit multiply-add chains (Philox core op), 4 independent
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) { uint64_t m = (uint64_t)u[i] * 0xD2511F53u; u[i] = (uint32_t)(m >> 32) ^ (uint32_t)m; }
        } else if constexpr (KIND == 4) {     // 3 independent fma chains (like 3 actions' dot products)
#pragma unroll
            for (int r = 0; r < 21; ++r)
#pragma unroll
                for (int i = 0; i < 3; ++i) a[i] = __builtin_fmaf(a[i], b, c);
        } else if constexpr (KIND == 5) {     // v_cndmask selects, independent
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = (a[i] > c) ? a[(i + 1) & 15] : b;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(u[0] ^ u[1] ^ u[2] ^ u[3]);
}

template <int KIND>
int run(const char* name, int ops_per_iter, float* d_out, int half_exec = 0) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;     // 256 threads = 4 waves per block = one wave per SIMD per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 10, half_exec);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, ITER, half_exec);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_wave = (double)ITER * ops_per_iter;
        const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
        printf("%-28s waves/SIMD %d: %8.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms,
               ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    }
    return 0;
}

int main() {
    float* d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    run<0>("fma x16 independent", 64, d_out);
    run<4>("fma x3 chains", 63, d_out);
    run<1>("fma dependent chain", 64, d_out);
    run<2>("pk_fma x8 independent", 64, d_out);
    run<3>("mad_u64_u32 x4 (+xor)", 64 * 2, d_out);
    run<5>("cndmask x16", 64 * 2, d_out);
    // does a wave with half of its lanes masked off issue faster?  (it does not: see profiles/r01_ubench_valu_issue.txt)
    run<0>("fma x16 indep, 32 lanes", 64, d_out, 1);
    run<2>("pk_fma x8 indep, 32 lanes", 64, d_out, 1);
    return 0;
}
