// kernels_util.hip -- the small kernels more than one unit of the C ABI launches (table apply / scatter, the peer exchange, weight accessors, checksums ...): one
// definition, one copy of the machine code (ctx.hpp declares them; a kernel's host stub is an ordinary function).
#include "ctx.hpp"

RSRL_DEFINE_FX_READER(fx_saturations_util)

// tile coding, shared W: sum the n_rep copies of the FIXED-POINT delta table (and clear them) -- exact 64-bit integer sums,
// converted once: single rank W += fl(sum * lsb), otherwise that float goes to dW for the exchange.  n is a multiple of 2.
__global__ __launch_bounds__(256) void k_apply_rep(float* __restrict__ W, float* __restrict__ dW, long long* __restrict__ rep, int n_rep, int n, float lsb) {
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (j >= n) return;
    if (j + 1 >= n || (n & 1)) {                      // odd table sizes (copies not 16-byte aligned): one entry at a time
        for (int e = j; e < n && e < j + 2; ++e) {
            long long a = 0;
            for (int r = 0; r < n_rep; ++r) { long long* p = rep + (int64_t)r * n + e; const long long v = *p; if (v != 0) { a += v; *p = 0; } }
            const float d = (float)a * lsb;
            if (W) W[e] += d; else dW[e] = d;
        }
        return;
    }
    // all copies' loads go out together (one memory round trip instead of n_rep dependent ones), then the touched ones are cleared
    constexpr int kMaxRep = 16;
    longlong2 v[kMaxRep];
#pragma unroll
    for (int r = 0; r < kMaxRep; ++r)
        v[r] = r < n_rep ? *reinterpret_cast<const longlong2*>(rep + (int64_t)r * n + j) : make_longlong2(0, 0);
    long long a0 = 0, a1 = 0;
#pragma unroll
    for (int r = 0; r < kMaxRep; ++r) {
        if (r < n_rep && (v[r].x != 0 || v[r].y != 0)) {
            a0 += v[r].x; a1 += v[r].y;
            *reinterpret_cast<longlong2*>(rep + (int64_t)r * n + j) = make_longlong2(0, 0);
        }
    }
    const float d0 = (float)a0 * lsb, d1 = (float)a1 * lsb;
    if (W) { W[j] += d0; W[j + 1] += d1; }
    else { dW[j] = d0; dW[j + 1] = d1; }
}
// Shared tile coding, the scatter as a kernel of its own.  Block (chunk c, tiling t) takes the terms of `per_block` consecutive
// learners for ONE tiling: LDS slice of that tiling (64-bit fixed point), one LDS atomic per learner, ONE sweep, one device atomic
// per touched entry into copy c % n_rep of the table.  Against scattering inside the step kernel (1 024 learners x 8 tilings per
// block: a sweep per tiling per 1 024 learners, ~300 touched entries each) a block here covers 8x the learners per sweep and
// per flush: an eighth of the sweeps, a quarter of the device atomics.  The sums are integers: the same table whatever the
// grouping -- bit-identical to the fused scatter and to the oracle.
__global__ __launch_bounds__(1024) void k_tile_scatter(const uint16_t* __restrict__ keys, const float* __restrict__ terms, int64_t N, int S,
                                                       int per_block, long long* __restrict__ dW64, int n_rep, int64_t rep_stride, float inv_lsb) {
    extern __shared__ long long scatter_slice[];
    const int t = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * per_block;
    const int64_t i1 = i0 + per_block < N ? i0 + per_block : N;
    const uint16_t* __restrict__ kt = keys + (int64_t)t * N;
    float sc[8]; uint16_t kk[8];
    auto fetch = [&](int64_t ib) {                                       // eight learners per thread, their loads in flight together
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t i = ib + e * (int64_t)blockDim.x + threadIdx.x;
            sc[e] = i < i1 ? terms[i] : 0.0f;
            kk[e] = i < i1 ? kt[i] : (uint16_t)0;
        }
    };
    fetch(i0);                                                           // the first (usually the only) batch is on its way while the slice is cleared
    for (int j = threadIdx.x; j < S; j += blockDim.x) scatter_slice[j] = 0;
    __syncthreads();
    for (int64_t ib = i0; ib < i1; ib += 8 * (int64_t)blockDim.x) {
        if (ib != i0) fetch(ib);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned long long q = fx_quantise(sc[e], inv_lsb);      // the learner's term as ONE integer, the same for all its tilings
            if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&scatter_slice[kk[e]]), q);
        }
    }
    __syncthreads();
    long long* __restrict__ dst = dW64 + (int64_t)(blockIdx.x % (unsigned)n_rep) * rep_stride + (int64_t)t * S;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const long long v = scatter_slice[j];
        if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&dst[j]), (unsigned long long)v);
    }
}

// rsrl_hip_handle on shared weights: the mini-batch's fixed-point delta table -> float delta, table cleared
__global__ __launch_bounds__(256) void k_fx_finalize(long long* __restrict__ fx, float* __restrict__ dW, int n, float lsb) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long v = fx[j];
    dW[j] = (float)v * lsb;
    if (v != 0) fx[j] = 0;
}
// actions index weight columns: whatever a caller stored through a DEVICE pointer is brought into [0, A)
__global__ void k_clamp_actions(int32_t* __restrict__ a, int64_t n, int A) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = a[i]; a[i] = v < 0 ? 0 : (v > A - 1 ? A - 1 : v); }
}
// ---- RSRL_EXCHANGE_PEER: one-hop peer-write exchange of the shared-W delta ------------------------------
// Every rank stores its delta into slot [parity][rank] of EVERY rank's receive buffer (hipIpc-mapped: xGMI stores across
// GPUs), as naturally aligned 8-byte granules {value bits, tag = low 32 bits of (batch-step + 1)} written by one
// system-scope store each -- the tag travels with the value, so there is no separate flag, no fence and no second hop
// (MI355X_MICROARCH.md, hand-off price list: "granules for latency").  Each rank then sums the world slots in RANK order:
// every replica adds the same numbers in the same order => the replicas of W stay bit-identical, whatever the arrival
// order.  Two parities: a rank can be at most one exchange ahead of the slowest one (it cannot pass exchange t+1 before
// every peer has pushed t+1, i.e. finished reading t).
__global__ __launch_bounds__(256) void k_peer_push(const float* __restrict__ dW, int n, uint2* const* __restrict__ peers, int world, int rank,
                                                   uint64_t t, const uint64_t* __restrict__ t_dev, int64_t xdelta) {
    if (t_dev) t += *t_dev;
    const uint64_t xs = t + (uint64_t)xdelta;          // exchange sequence number: parity and tag (see Common::xdelta)
    // (a grid-stride loop: the grid is capped where several ranks share one device, peer_grid() below)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint64_t g = (uint64_t)__float_as_uint(dW[j]) | ((uint64_t)(uint32_t)(xs + 1) << 32);
        const size_t slot = ((size_t)(xs & 1) * world + rank) * (size_t)n + j;
        for (int r = 0; r < world; ++r)
            __hip_atomic_store(reinterpret_cast<uint64_t*>(peers[r] + slot), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// sum over ranks (ascending) of slot [parity][r][j], each polled until its tag says "exchange xs".  The spin is bounded by the
// wall clock (100 MHz; `timeout` ticks): a missing peer sets *err instead of hanging the GPU, and the sum is POISONED (NaN) --
// a partial sum is never applied silently.
__device__ __forceinline__ float peer_sum(const uint2* __restrict__ recv, int n, int world, int j, uint64_t xs, uint32_t* __restrict__ err, uint64_t timeout) {
    const uint32_t want = (uint32_t)(xs + 1);
    const uint64_t t_start = wall_clock64();
    bool failed = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // an earlier exchange timed out: fail fast, do not wait again
    float acc = 0.0f;
    for (int r = 0; r < world; ++r) {
        const uint64_t* p = reinterpret_cast<const uint64_t*>(recv + ((size_t)(xs & 1) * world + r) * (size_t)n + j);
        uint64_t g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while ((uint32_t)(g >> 32) != want) {
            if (failed || wall_clock64() - t_start > timeout) { atomicOr(err, 1u); failed = true; break; }
            __builtin_amdgcn_s_sleep(8);
            g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        acc += __uint_as_float((uint32_t)g);
    }
    return failed ? __builtin_nanf("") : acc;
}
__global__ __launch_bounds__(256) void k_peer_reduce(float* __restrict__ dW, int n, const uint2* __restrict__ recv, int world, uint64_t t,
                                                     const uint64_t* __restrict__ t_dev, int64_t xdelta, uint32_t* __restrict__ err, uint64_t timeout) {
    if (t_dev) t += *t_dev;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        dW[j] = peer_sum(recv, n, world, j, t + (uint64_t)xdelta, err, timeout);
}
// multi-rank mode: the fold as a kernel of its own (the copies of batch-step t's fixed-point delta table -> one float per
// output), feeding the exchange
__device__ __forceinline__ float tab_total(const long long* __restrict__ tab, int n, int j, float lr, uint64_t t) {
    DeltaTab dt(const_cast<long long*>(tab), n, lr, t);
    long long s = 0;
#pragma unroll
    for (int r = 0; r < kTabRep; ++r) s += dt.out[r * n + j];
    return (float)s * dt.lsb;
}
__global__ __launch_bounds__(kBlock) void k_tab_finalize(const long long* __restrict__ tab, int n, float lr, float* __restrict__ dW, uint64_t t,
                                                         const uint64_t* __restrict__ t_dev) {
    if (t_dev) t += *t_dev;
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j < n) dW[j] = tab_total(tab, n, j, lr, t);
}

// dense peer path, ONE launch instead of four: the delta of batch-step t goes from the fixed-point table straight into every
// rank's receive slot, and the same thread then sums the ranks' slots (rank order, bounded wait as k_peer_reduce) into W.
// Every rank pushes before it waits, so the ranks cannot wait for each other's pushes in a cycle.
__global__ __launch_bounds__(256) void k_tab_exchange_apply(const long long* __restrict__ tab, int n, float lr, uint2* const* __restrict__ peers,
                                                            const uint2* __restrict__ recv, float* __restrict__ W, int world, int rank, uint64_t t,
                                                            const uint64_t* __restrict__ t_dev, int64_t xdelta, uint32_t* __restrict__ err, uint64_t timeout) {
    if (t_dev) t += *t_dev;
    const uint64_t xs = t + (uint64_t)xdelta;
    // every element is pushed BEFORE the first wait (two grid-stride loops: the grid may be capped, peer_grid() below)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float tot = tab_total(tab, n, j, lr, t);
        const uint64_t mine = (uint64_t)__float_as_uint(tot) | ((uint64_t)(uint32_t)(xs + 1) << 32);
        const size_t slot = ((size_t)(xs & 1) * world + rank) * (size_t)n + j;
        for (int r = 0; r < world; ++r)
            __hip_atomic_store(reinterpret_cast<uint64_t*>(peers[r] + slot), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        W[j] += peer_sum(recv, n, world, j, xs, err, timeout);
}
__global__ void k_fill_f32(float* __restrict__ p, int64_t n, float v) { const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_set_dyn(DynParams* __restrict__ d, DynParams v) { *d = v; }
__global__ void k_set_t(uint64_t* __restrict__ t_dev, uint64_t v) { *t_dev = v; }
__global__ void k_advance_t(uint64_t* __restrict__ t_dev, uint64_t d) { *t_dev += d; }
__global__ void k_apply_dw(float* __restrict__ W, float* __restrict__ dW, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) { W[j] += dW[j]; dW[j] = 0.0f; }
}

// get/set of one learner's weights as row-major f32[F][A] (ndarray (F, A))   params/mod.rs:116-134
// device layouts: Fourier W[A][F][Nw] (learner fastest); tile coding W[Nw][F][A]
__device__ __forceinline__ int64_t w_index(bool tile, int64_t stride, int64_t wi, int F, int A, int f, int b) {
    return tile ? (wi * (int64_t)F + f) * A + b : ((int64_t)(b * F + f)) * stride + wi;
}
__global__ void k_weights_get(const float* __restrict__ W, bool tile, int64_t stride, int64_t wi, int F, int A, float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F * A) return;
    out[j] = W[w_index(tile, stride, wi, F, A, j / A, j % A)];
}
__global__ void k_weights_set(float* __restrict__ W, bool tile, int64_t stride, int64_t wi, int F, int A, const float* __restrict__ in) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F * A) return;
    W[w_index(tile, stride, wi, F, A, j / A, j % A)] = in[j];
}
// grid.x covers the learners, grid.y strides over the F*A weights
__global__ void k_weights_set_all(float* __restrict__ W, bool tile, int64_t N, int64_t stride, int64_t ls, int F, int A, const float* __restrict__ in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    for (int j = blockIdx.y; j < F * A; j += gridDim.y) W[w_index(tile, stride, i * ls, F, A, j / A, j % A)] = in[j];
}
// the same sum for a learner-major W[N][AF], with every word weighted by the index it has in the feature-major layout
// ((row)*N + learner): the checksum of the weights does not depend on the layout the ctx chose
__global__ void k_checksum_lm(const uint32_t* __restrict__ p, int64_t N, int AF, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    const size_t n = (size_t)N * AF;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const size_t learner = j / AF, row = j % AF;
        acc += (unsigned long long)p[j] * (2ull * (row * (size_t)N + learner) + 1ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
// order-independent checksum: sum over words of bits * (2*index + 1)  (mod 2^64)
__global__ void k_checksum(const uint32_t* __restrict__ p, size_t n, size_t index_offset, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        acc += (unsigned long long)p[j] * (2ull * (j + index_offset) + 1ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

