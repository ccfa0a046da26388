// measure.hip -- measurement hooks that need no ctx (bench.py).
//
// rsrl_hip_measure_copy: the device's own copy bandwidth, so that an HBM fraction can be quoted against what THIS box's memory system delivers next to the
// published 8 TB/s (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).  A float4 grid-stride copy kernel, timed with HIP events.
#include <hip/hip_runtime.h>

#include "../../include/rsrl_hip.h"

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
// three spellings of the same copy (the best one is reported: what the memory system delivers, not what one spelling reaches)
//   0: grid-stride, one float4 per iteration        1: four independent float4 loads in flight per thread, then the four stores
//   2: as 1 with non-temporal stores
template <int V>
__global__ __launch_bounds__(256) void k_copy_f4(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (V == 0) {
        for (; i < n; i += stride) dst[i] = src[i];
    } else {
        for (; i + 3 * stride < n; i += 4 * stride) {
            const f4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
            if constexpr (V == 2) {
                __builtin_nontemporal_store(a, &dst[i]); __builtin_nontemporal_store(b, &dst[i + stride]);
                __builtin_nontemporal_store(c, &dst[i + 2 * stride]); __builtin_nontemporal_store(d, &dst[i + 3 * stride]);
            } else {
                dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
            }
        }
        for (; i < n; i += stride) dst[i] = src[i];
    }
}
template <int V>
float time_copy(hipStream_t st, hipEvent_t a, hipEvent_t b, const f4* src, f4* dst, size_t n, int reps, unsigned grid) {
    float ms = 0.0f;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_copy_f4<V>, dim3(grid), dim3(256), 0, st, src, dst, n);
    (void)hipEventRecord(a, st);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy_f4<V>, dim3(grid), dim3(256), 0, st, src, dst, n);
    (void)hipEventRecord(b, st);
    if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0f;
    return ms;
}
}  // namespace

extern "C" int rsrl_hip_measure_copy(int device, size_t bytes, int reps, double* gbps_out) {
    if (!gbps_out || bytes < 16 || reps < 1) return RSRL_HIP_EINVAL;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return RSRL_HIP_EHIP; }
    const size_t n = bytes / 16;
    f4 *src = nullptr, *dst = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st = nullptr;
    int rc = RSRL_HIP_OK;
    if (hipMalloc((void**)&src, n * 16) != hipSuccess || hipMalloc((void**)&dst, n * 16) != hipSuccess) { rc = RSRL_HIP_ENOMEM; goto out; }
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { rc = RSRL_HIP_EHIP; goto out; }
    if (hipMemsetAsync(src, 0x3c, n * 16, st) != hipSuccess) { rc = RSRL_HIP_EHIP; goto out; }
    {
        double best = 0.0;
        for (unsigned per_cu : {8u, 16u, 32u}) {                        // blocks per CU of the grid-stride loops
            const unsigned cap = 256u * per_cu;
            const unsigned grid = (unsigned)((n + 255) / 256 < cap ? (n + 255) / 256 : cap);
            const float ms[3] = {time_copy<0>(st, a, b, src, dst, n, reps, grid), time_copy<1>(st, a, b, src, dst, n, reps, grid),
                                 time_copy<2>(st, a, b, src, dst, n, reps, grid)};
            for (float m : ms)
                if (m > 0.0f) { const double gbps = 2.0 * (double)(n * 16) * reps / ((double)m * 1e-3) / 1e9; best = gbps > best ? gbps : best; }
        }
        if (!(best > 0.0)) { rc = RSRL_HIP_EHIP; goto out; }
        *gbps_out = best;                                               // read + write
    }
out:
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (st) (void)hipStreamDestroy(st);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    (void)hipGetLastError();
    return rc;
}
