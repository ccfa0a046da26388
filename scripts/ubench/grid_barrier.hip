// Microbenchmark: cost of one device-wide step of a persistent kernel on gfx950 (8 XCDs, one L2 each):
//   every block writes a 108-float row -> release -> arrive on a counter -> bounded spin -> acquire -> every block sums all rows.
// Variants: how the rows are made visible (agent-scope fences around plain accesses vs sc1 "atomic" accesses) and whether the
// fold is done at all.   build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int AF = 108, BLOCK = 512;

__device__ __forceinline__ bool wait_count(unsigned* ctr, unsigned target, unsigned* err) {
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 100000000ull) { *err = 1; return false; }     // 1 s at 100 MHz: never hang the box
    }
    return true;
}

// MODE 0: barrier only; 1: fences + plain row accesses + fold; 2: sc1 (agent-scope relaxed atomic) row accesses + fold
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k(float* rows /*[2][AF][nb]*/, unsigned* ctr, unsigned* err, float* out, int iters) {
    const int nb = gridDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float sh_w[AF];
    __shared__ int ok;
    if (threadIdx.x < AF) sh_w[threadIdx.x] = 0.0f;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        float* r = rows + (size_t)(it & 1) * AF * nb;
        if (threadIdx.x < AF) {
            const float v = 1e-3f * (float)(blockIdx.x + it + threadIdx.x) + sh_w[threadIdx.x] * 1e-6f;
            if (MODE == 2) __hip_atomic_store(&r[(size_t)threadIdx.x * nb + blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else r[(size_t)threadIdx.x * nb + blockIdx.x] = v;
        }
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            ok = wait_count(ctr, (unsigned)nb * (unsigned)(it + 1), err) ? 1 : 0;
        }
        __syncthreads();
        if (!ok) return;
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (MODE >= 1) {
            // wave w owns outputs j = w, w+8, ...: lane l sums rows l, l+64, ... then a wave reduction
            for (int j = wave; j < AF; j += BLOCK / 64) {
                float acc = 0.0f;
                for (int r0 = lane; r0 < nb; r0 += 64) {
                    const float* p = &r[(size_t)j * nb + r0];
                    acc += (MODE == 2) ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
                }
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                if (lane == 0) sh_w[j] += acc;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x < AF && blockIdx.x == 0) out[threadIdx.x] = sh_w[threadIdx.x];
}

// The guide's "barrier-xcd": blocks arrive on the counter of their group (g = blockIdx % 8: the XCD under the observed placement -- speed only, the
// protocol does not depend on it), the group's LAST arriver goes to the top counter, waits for all groups and releases its group's generation word.
// Counters are monotonic (no reset); one 128-byte line per word.
struct XcdBar { unsigned w[(8 + 1 + 8) * 32]; };
__device__ __forceinline__ unsigned* xb_cnt(XcdBar* b, int g) { return &b->w[g * 32]; }
__device__ __forceinline__ unsigned* xb_top(XcdBar* b) { return &b->w[8 * 32]; }
__device__ __forceinline__ unsigned* xb_gen(XcdBar* b, int g) { return &b->w[(9 + g) * 32]; }
__global__ __launch_bounds__(BLOCK) void k_xcd(XcdBar* bar, unsigned* err, int iters) {
    const int nb = gridDim.x, g = blockIdx.x & 7;
    const unsigned members = (unsigned)((nb - g + 7) / 8), groups = (unsigned)(nb < 8 ? nb : 8);
    __shared__ int ok;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            bool fine = true;
            const unsigned old = __hip_atomic_fetch_add(xb_cnt(bar, g), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == members * (unsigned)(it + 1)) {
                __hip_atomic_fetch_add(xb_top(bar), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                fine = wait_count(xb_top(bar), groups * (unsigned)(it + 1), err);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(xb_gen(bar, g), (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                fine = wait_count(xb_gen(bar, g), (unsigned)(it + 1), err);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            ok = fine ? 1 : 0;
        }
        __syncthreads();
        if (!ok) return;
    }
}
int run_xcd(int nb) {
    XcdBar* bar; unsigned* err;
    CHECK(hipMalloc(&bar, sizeof(XcdBar))); CHECK(hipMalloc(&err, 4));
    for (int iters : {100, 2000}) {
        CHECK(hipMemset(bar, 0, sizeof(XcdBar))); CHECK(hipMemset(err, 0, 4));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_xcd, dim3(nb), dim3(BLOCK), 0, 0, bar, err, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h_err; CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        printf("%-34s blocks %3d iters %4d: %8.3f ms  = %6.2f us per step   err %u\n", "barrier only, XCD-hierarchical", nb, iters, ms, ms * 1e3 / iters, h_err);
    }
    hipFree(bar); hipFree(err);
    return 0;
}

template <int MODE>
int run(const char* name, int nb) {
    float *rows, *out; unsigned *ctr, *err;
    CHECK(hipMalloc(&rows, sizeof(float) * 2 * AF * nb)); CHECK(hipMalloc(&out, sizeof(float) * AF));
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&err, 4));
    for (int iters : {100, 2000}) {
        CHECK(hipMemset(ctr, 0, 4)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(rows, 0, sizeof(float) * 2 * AF * nb));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(BLOCK), 0, 0, rows, ctr, err, out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h_err; float h_out[AF];
        CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost));
        // expected W[j] after `iters` folds (MODE >= 1, ignoring the 1e-6 feedback): sum over it, b of 1e-3 (b + it + j)
        printf("%-34s blocks %3d iters %4d: %8.3f ms  = %6.2f us per step   err %u   W[0] %.4f W[107] %.4f\n", name, nb, iters, ms, ms * 1e3 / iters, h_err,
               h_out[0], h_out[107]);
    }
    hipFree(rows); hipFree(out); hipFree(ctr); hipFree(err);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    if (argc > 1) {      // `grid_barrier 512 ...`: the barrier alone at the given block counts (512 blocks x 512 threads = C3's 262 144 resident learners)
        for (int a = 1; a < argc; ++a) if (run<0>("barrier only", atoi(argv[a])) || run_xcd(atoi(argv[a]))) return 1;
        return 0;
    }
    for (int nb : {64, 256}) {
        if (run<0>("barrier only", nb)) return 1;
        if (run<1>("fences + plain rows + fold", nb)) return 1;
        if (run<2>("sc1 rows + fold", nb)) return 1;
    }
    return 0;
}
