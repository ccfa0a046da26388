// Microbenchmark: one device-wide all-reduce of a 108-entry 64-bit fixed-point vector per step of a PERSISTENT kernel on gfx950
// (256 blocks of 512 threads, one per CU) -- what the shared-W MountainCar step needs between two batch-steps.  Three exchanges:
//
//   granule   reduce-scatter + all-gather through data-tagged 8-byte granules {value32, tag32} written by ONE relaxed agent-scope
//             (sc1) store each and polled with relaxed agent-scope loads: no counter, no fence (MI355X_MICROARCH.md price list,
//             "granules for latency"; cdna_hip_programming.md Guideline 16 form R2).  Hop 1: block b stores the two halves of its 108
//             partial sums into A[entry][half][b]; the owner block of an entry (block e) polls that entry's 2 x nb granules, one
//             per thread, and adds them up (exact integers).  Hop 2: the owner stores the 64-bit total as two granules B[parity][e][2];
//             thread e of EVERY block polls them.  Two fabric hops per step, placement-independent.
//   xcd       the XCD-hierarchical barrier of the guide (barrier-xcd): per-group counter (group = block % 8), group leader
//             release fence -> top counter -> acquire fence -> per-group generation; the partial sums travel as device atomics
//             into a table before the barrier and are read back after it.
//   flat      round 2's flat counter barrier (profiles/r02_ubench_grid_barrier.txt) with the same atomics.
//
// Every step every block checks every total against the closed form: a wrong or stale value shows up as `bad`.
//   build: hipcc --offload-arch=gfx950 -O3 -o granule_allreduce granule_allreduce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int AF = 108, BLOCK = 512;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ long long partial(int b, int e, int it) {         // block b's fixed-point partial sum of entry e at step `it`
    return (long long)(b + 1) * 1000003ll * (e + 1) + (long long)it * 7919ll + ((long long)b << 33) - (1ll << 40);     // both signs occur
}
__device__ __forceinline__ long long expected_total(int nb, int e, int it) {   // sum over b of partial(b, e, it), closed form
    const long long n = nb;
    return n * (n + 1) / 2 * 1000003ll * (e + 1) + n * (long long)it * 7919ll + ((n * (n - 1) / 2) << 33) - n * (1ll << 40);
}
__device__ __forceinline__ unsigned long long granule(unsigned value, unsigned tag) { return ((unsigned long long)tag << 32) | value; }

// poll one granule until its tag matches; bounded (sets *tmo and gives up)
__device__ __forceinline__ unsigned poll(gu64* g, unsigned tag, gu32* tmo) {
    unsigned long long x = __hip_atomic_load(g, RLX_AGENT);
    unsigned spins = 0;
    while ((unsigned)(x >> 32) != tag) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, RLX_AGENT); break; }
        x = __hip_atomic_load(g, RLX_AGENT);
    }
    return (unsigned)x;
}

__global__ __launch_bounds__(BLOCK) void k_granule(gu64* A /*[AF][2][nb]*/, gu64* B /*[2][AF][2]*/, gu32* tmo, unsigned* bad, int iters, unsigned tag0) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned long long red[2];
    __shared__ long long tot[AF];
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned tag = tag0 + (unsigned)it + 1u;
        if (tid < 2) red[tid] = 0;
        // hop 1, publish: thread e stores the two halves of this block's partial sum of entry e
        if (tid < AF) {
            const long long p = partial(b, tid, it);
            __hip_atomic_store(&A[((size_t)tid * 2 + 0) * nb + b], granule((unsigned)p, tag), RLX_AGENT);
            __hip_atomic_store(&A[((size_t)tid * 2 + 1) * nb + b], granule((unsigned)((unsigned long long)p >> 32), tag), RLX_AGENT);
        }
        __syncthreads();
        // hop 1, reduce: block e owns entry e (entries e, e + nb, ... when there are fewer blocks than entries)
        for (int e = b; e < AF; e += nb) {
            for (int j = tid; j < 2 * nb; j += BLOCK) {
                const int half = j / nb;
                const unsigned v = poll(&A[(size_t)e * 2 * nb + j], tag, tmo);
                // lo halves add up as unsigned, hi halves as signed 32-bit numbers: total = (sum hi << 32) + sum lo, exact
                const unsigned long long add = half ? (unsigned long long)(long long)(int)v : (unsigned long long)v;
                unsigned long long w = add;
                // the 64 lanes of a wave hold one half (nb is a multiple of 64 here) -> wave sum, then one LDS atomic per wave
                for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
                if ((tid & 63) == 0) atomicAdd(&red[half], w);
            }
            __syncthreads();
            if (tid == 0) {
                const unsigned long long total = (red[1] << 32) + red[0];
                gu64* dst = &B[((size_t)(it & 1) * AF + e) * 2];
                __hip_atomic_store(dst + 0, granule((unsigned)total, tag), RLX_AGENT);
                __hip_atomic_store(dst + 1, granule((unsigned)(total >> 32), tag), RLX_AGENT);
                red[0] = 0; red[1] = 0;
            }
            __syncthreads();
        }
        // hop 2, gather: thread e of every block polls entry e's total
        if (tid < AF) {
            gu64* src = &B[((size_t)(it & 1) * AF + tid) * 2];
            const unsigned lo = poll(src + 0, tag, tmo), hi = poll(src + 1, tag, tmo);
            tot[tid] = (long long)(((unsigned long long)hi << 32) | lo);
            if (tot[tid] != expected_total(nb, tid, it)) ++nbad;
        }
        __syncthreads();
        if (__hip_atomic_load(tmo, RLX_AGENT)) break;
    }
    if (nbad) atomicAdd(bad, nbad);
}


// ---- variant 2: ONE granule per entry -- {tag12 | 52-bit two's-complement value}.  A block's partial sum is clamped to +-2^42 and a
// rank's total over <= 256 blocks to +-2^50, so both fit 52 bits; the 12-bit tag changes every step and a slot is rewritten every
// step, so a stale granule can never carry the wanted tag.  PAIR = 1: hop 1 stores two neighbouring entries with one 16-byte
// store (each 8-byte half validates itself: no 16-byte atomicity is assumed).
__device__ __forceinline__ unsigned long long g52(long long v, unsigned tag) { return ((unsigned long long)(tag & 0xfffu) << 52) | ((unsigned long long)v & 0xfffffffffffffull); }
__device__ __forceinline__ long long g52_value(unsigned long long x) { return (long long)(x << 12) >> 12; }
// DEPTH loads of the same granule in flight, a few dozen cycles apart: the first one that carries the tag ends the wait, so the
// wait is quantised to a fraction of a memory round trip instead of a whole one
template <int DEPTH, int GAP>
__device__ __forceinline__ long long poll52p(gu64* g, unsigned tag, gu32* tmo) {
    unsigned spins = 0;
    for (;;) {
        unsigned long long x[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { x[d] = __hip_atomic_load(g, RLX_AGENT); if (d + 1 < DEPTH) __builtin_amdgcn_s_sleep(GAP); }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) if ((unsigned)(x[d] >> 52) == (tag & 0xfffu)) return g52_value(x[d]);
        if (++spins > (1u << 20)) { __hip_atomic_store(tmo, 1u, RLX_AGENT); return 0; }
    }
}
__device__ __forceinline__ long long poll52(gu64* g, unsigned tag, gu32* tmo) {
    unsigned long long x = __hip_atomic_load(g, RLX_AGENT);
    unsigned spins = 0;
    while ((unsigned)(x >> 52) != (tag & 0xfffu)) {
        if (++spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, RLX_AGENT); break; }
        x = __hip_atomic_load(g, RLX_AGENT);
    }
    return g52_value(x);
}
template <int PAIR, int DEPTH = 1, int GAP = 1>
__global__ __launch_bounds__(BLOCK) void k_granule52(gu64* A /*[AF][nb] or [AF/2][nb][2]*/, gu64* B /*[2][AF]*/, gu32* tmo, unsigned* bad, int iters, unsigned tag0) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned long long red[2];
    __shared__ long long tot[AF];
    unsigned nbad = 0;
    if (tid < 2) red[tid] = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const unsigned tag = tag0 + (unsigned)it + 1u;
        if (PAIR) {
            if (tid < AF / 2) {
                typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
                const u2 v = {g52(partial(b, 2 * tid, it), tag), g52(partial(b, 2 * tid + 1, it), tag)};
                typedef int i4 __attribute__((ext_vector_type(4)));
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, (int)((((size_t)tid * nb + b) * 2) * 8), 0, 16 /* sc1 */);
            }
        } else if (tid < AF) {
            __hip_atomic_store(&A[(size_t)tid * nb + b], g52(partial(b, tid, it), tag), RLX_AGENT);
        }
        // owners: PAIR = 0: block e owns entry e, thread k < nb polls source block k.  PAIR = 1: block c < AF/2 owns entries 2c, 2c+1,
        // thread j < 2 nb polls half j & 1 of source block j >> 1
        const int n_own = PAIR ? AF / 2 : AF, n_poll = PAIR ? 2 * nb : nb;
        for (int c = b; c < n_own; c += nb) {
            for (int j = tid; j < n_poll; j += BLOCK) {
                long long w = DEPTH > 1 ? poll52p<DEPTH, GAP>(&A[(size_t)c * n_poll + j], tag, tmo) : poll52(&A[(size_t)c * n_poll + j], tag, tmo);
                if (PAIR) {                                           // even lanes hold entry 2c, odd lanes entry 2c+1
                    for (int o = 32; o > 1; o >>= 1) w += __shfl_xor(w, o);
                    if ((tid & 63) < 2) atomicAdd(&red[tid & 1], (unsigned long long)w);
                } else {
                    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
                    if ((tid & 63) == 0) atomicAdd(&red[0], (unsigned long long)w);
                }
            }
            __syncthreads();
            if (tid < (PAIR ? 2 : 1)) {
                const int e = PAIR ? 2 * c + tid : c;
                __hip_atomic_store(&B[(size_t)(it & 1) * AF + e], g52((long long)red[tid], tag), RLX_AGENT);
                red[tid] = 0;
            }
            __syncthreads();
        }
        if (tid < AF) {
            tot[tid] = DEPTH > 1 ? poll52p<DEPTH, GAP>(&B[(size_t)(it & 1) * AF + tid], tag, tmo) : poll52(&B[(size_t)(it & 1) * AF + tid], tag, tmo);
            if (tot[tid] != expected_total(nb, tid, it)) ++nbad;
        }
        __syncthreads();
        if (__hip_atomic_load(tmo, RLX_AGENT)) break;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// ---- barrier variants: partial sums as device atomics into a table (3 rotating sets), barrier, read back ------------------
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, gu32* tmo) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, RLX_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, RLX_AGENT); return false; }
    }
    return true;
}
// XCD-hierarchical barrier: arrivals on the group's counter; the group's last arriver goes to the top counter; the top's last
// arriver bumps the generation words of all groups; everybody else polls its own group's generation
__device__ __forceinline__ void barrier_xcd(unsigned* grp_cnt /*[8*32]*/, unsigned* top_cnt, unsigned* gen /*[8*32]*/, unsigned epoch, int n_groups,
                                            gu32* tmo) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x % n_groups;
        const unsigned members = (gridDim.x - g + n_groups - 1) / n_groups;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned arrived = __hip_atomic_fetch_add(&grp_cnt[g * 32], 1u, RLX_AGENT) + 1u;
        if (arrived == members * epoch) {                                   // last of the group
            const unsigned top = __hip_atomic_fetch_add(top_cnt, 1u, RLX_AGENT) + 1u;
            if (top == (unsigned)n_groups * epoch)
                for (int q = 0; q < n_groups; ++q) __hip_atomic_store(&gen[q * 32], epoch, RLX_AGENT);
        }
        wait_ge(&gen[g * 32], epoch, tmo);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__device__ __forceinline__ void barrier_flat(unsigned* cnt, unsigned epoch, gu32* tmo) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, RLX_AGENT);
        wait_ge(cnt, gridDim.x * epoch, tmo);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
template <int KIND>   // 0 = xcd, 1 = flat
__global__ __launch_bounds__(BLOCK) void k_barrier(unsigned long long* tab /*[3][16][AF]*/, unsigned* sync_words, gu32* tmo, unsigned* bad, int iters) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long* out = tab + (size_t)(it % 3) * 16 * AF;
        unsigned long long* zero = tab + (size_t)((it + 1) % 3) * 16 * AF;
        if (tid < AF) {
            atomicAdd(&out[(b % 16) * AF + tid], (unsigned long long)partial(b, tid, it));
            if (b < 16) __hip_atomic_store(&zero[b * AF + tid], 0ull, RLX_AGENT);
        }
        if (KIND == 0) barrier_xcd(sync_words, sync_words + 8 * 32, sync_words + 9 * 32, (unsigned)it + 1u, nb < 8 ? nb : 8, tmo);
        else barrier_flat(sync_words, (unsigned)it + 1u, tmo);
        if (tid < AF) {
            long long s = 0;
            for (int r = 0; r < 16; ++r) s += (long long)__hip_atomic_load(&out[r * AF + tid], RLX_AGENT);
            if (s != expected_total(nb, tid, it)) ++nbad;
        }
        if (__hip_atomic_load(tmo, RLX_AGENT)) break;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    for (int nb : {64, 128, 256}) {
        if (nb > prop.multiProcessorCount) continue;
        unsigned long long *A, *B, *tab; unsigned *tmo, *bad, *sw;
        CHECK(hipMalloc(&A, sizeof(unsigned long long) * AF * 2 * nb)); CHECK(hipMalloc(&B, sizeof(unsigned long long) * 2 * AF * 2));
        CHECK(hipMalloc(&tab, sizeof(unsigned long long) * 3 * 16 * AF)); CHECK(hipMalloc(&sw, 4 * 32 * 20));
        CHECK(hipMalloc(&tmo, 4)); CHECK(hipMalloc(&bad, 4));
        for (int kind = 0; kind < 9; ++kind) {
            for (int iters : {200, 4000}) {
                CHECK(hipMemset(A, 0, sizeof(unsigned long long) * AF * 2 * nb)); CHECK(hipMemset(B, 0, sizeof(unsigned long long) * 2 * AF * 2));
                CHECK(hipMemset(tab, 0, sizeof(unsigned long long) * 3 * 16 * AF)); CHECK(hipMemset(sw, 0, 4 * 32 * 20));
                CHECK(hipMemset(tmo, 0, 4)); CHECK(hipMemset(bad, 0, 4));
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k_granule, dim3(nb), dim3(BLOCK), 0, 0, (gu64*)A, (gu64*)B, (gu32*)tmo, bad, iters, 0u);
                else if (kind == 1) hipLaunchKernelGGL(k_barrier<0>, dim3(nb), dim3(BLOCK), 0, 0, tab, sw, (gu32*)tmo, bad, iters);
                else if (kind == 2) hipLaunchKernelGGL(k_barrier<1>, dim3(nb), dim3(BLOCK), 0, 0, tab, sw, (gu32*)tmo, bad, iters);
                else if (kind == 3) hipLaunchKernelGGL(k_granule52<0>, dim3(nb), dim3(BLOCK), 0, 0, (gu64*)A, (gu64*)B, (gu32*)tmo, bad, iters, 0u);
                else if (kind == 4) hipLaunchKernelGGL((k_granule52<1>), dim3(nb), dim3(BLOCK), 0, 0, (gu64*)A, (gu64*)B, (gu32*)tmo, bad, iters, 0u);
                else if (kind == 5) hipLaunchKernelGGL((k_granule52<1, 2, 2>), dim3(nb), dim3(BLOCK), 0, 0, (gu64*)A, (gu64*)B, (gu32*)tmo, bad, iters, 0u);
                else if (kind == 6) hipLaunchKernelGGL((k_granule52<1, 3, 2>), dim3(nb), dim3(BLOCK), 0, 0, (gu64*)A, (gu64*)B, (gu32*)tmo, bad, iters, 0u);
                else if (kind == 7) hipLaunchKernelGGL((k_granule52<1, 4, 1>), dim3(nb), dim3(BLOCK), 0, 0, (gu64*)A, (gu64*)B, (gu32*)tmo, bad, iters, 0u);
                else hipLaunchKernelGGL((k_granule52<1, 4, 4>), dim3(nb), dim3(BLOCK), 0, 0, (gu64*)A, (gu64*)B, (gu32*)tmo, bad, iters, 0u);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned h_tmo, h_bad;
                CHECK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost));
                printf("%-34s blocks %3d iters %4d: %8.3f ms = %6.2f us per step   timeout %u   bad totals %u\n",
                       kind == 0 ? "granule32 x2 reduce-scatter+gather" : (kind == 1 ? "atomics + XCD-hierarchical barrier" : (kind == 2 ? "atomics + flat counter barrier" : (kind == 3 ? "granule52 reduce-scatter+gather" : (kind == 4 ? "granule52, 16-B paired stores" : (kind == 5 ? "  + 2 polls in flight, gap 2" : (kind == 6 ? "  + 3 polls in flight, gap 2" : (kind == 7 ? "  + 4 polls in flight, gap 1" : "  + 4 polls in flight, gap 4"))))))), nb, iters, ms,
                       ms * 1e3 / iters, h_tmo, h_bad);
            }
        }
        CHECK(hipFree(A)); CHECK(hipFree(B)); CHECK(hipFree(tab)); CHECK(hipFree(sw)); CHECK(hipFree(tmo)); CHECK(hipFree(bad));
    }
    return 0;
}
