// Microbenchmark: cost of LDS float atomics (ds_add_f32, no return) per wave-instruction as a function of the address pattern
// and of the number of active lanes.  16 waves per CU (1024-thread blocks, one per CU), ITER dependent-free atomics per wave.
//   build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int active, int spread) {
    __shared__ float sl[8192];
    for (int j = threadIdx.x; j < 8192; j += 1024) sl[j] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // address pattern: `spread` distinct addresses per wave (1 = all lanes the same), waves use disjoint regions
    const int idx = wave * 256 + (lane % spread) * (KIND == 2 ? 33 : 1);
    if (lane < active) {
        for (int it = 0; it < ITER; ++it) {
            if (KIND == 3) atomicAdd(reinterpret_cast<unsigned long long*>(&sl[2 * idx]), 1ull);
            else if (KIND == 1) atomicAdd(reinterpret_cast<unsigned*>(&sl[idx]), 1u);
            else atomicAdd(&sl[idx], 1.0f);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sl[0] + sl[256];
}
template <int KIND>
int run(const char* name, int active, int spread) {
    float* out; CHECK(hipMalloc(&out, 256 * sizeof(float)));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(1024), 0, 0, out, active, spread);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(1024), 0, 0, out, active, spread);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double cyc = ms * 1e-3 * 2.4e9 / (16.0 * ITER);       // cycles per wave-instruction per CU (16 waves issue ITER each)
    printf("%-22s active lanes %2d, %2d distinct addresses per wave: %7.3f ms  %7.1f cycles per wave-instruction (CU-serial)\n", name, active, spread, ms, cyc);
    CHECK(hipFree(out));
    return 0;
}
int main() {
    const int act[] = {64, 32, 8, 1};
    const int spr[] = {1, 2, 8, 64};
    for (int a : act) for (int s : spr) if (s <= a) run<0>("ds_add_f32", a, s);
    for (int s : spr) run<1>("ds_add_u32", 64, s);
    run<2>("ds_add_f32 stride 33", 64, 64);
    for (int a : act) for (int s : spr) if (s <= a) run<3>("ds_add_u64", a, s);
    return 0;
}
