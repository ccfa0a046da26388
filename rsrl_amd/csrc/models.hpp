// models.hpp -- the two function-approximator "models" behind the trait-granular and shared-W kernels.
//
// A Model bundles a basis with its weight layout and gives the kernels four operations:
//   features(s)            basis.project(s)                                (lfa Basis::project)
//   q_all / q_index        Function<(S,)>::evaluate / Enumerable::evaluate_index   fa/linear.rs:303-311,360-362
//   update                 Handler<StateActionUpdate>: W[:,a] += scale*phi  fa/linear.rs:379-391 (per-env weights)
//   accumulate             the same term added to a delta buffer            (shared weights)
//
//   FourierModel<DOMAIN, ORDER>  dense features in VGPRs; W f32[A][F][Nw], learner index fastest
//   TileModel<DOMAIN, T>         T active indices (value 1.0);  W f32[Nw][F][A] (the reference's (F, A) rows per learner)
#pragma once

#include "kernels_reg.hpp"

namespace rsrl {

// wave64 sum via DPP (row_shr 1,2,4,8 then row_bcast 15 / 31); the total lands in lane 63
#define RSRL_DPP_ADD(v, ctrl, row_mask) \
    (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (row_mask), 0xf, false))
__device__ __forceinline__ float wave_sum_dpp_to_lane63(float v) {
    RSRL_DPP_ADD(v, 0x111, 0xf);   // row_shr:1
    RSRL_DPP_ADD(v, 0x112, 0xf);   // row_shr:2
    RSRL_DPP_ADD(v, 0x114, 0xf);   // row_shr:4
    RSRL_DPP_ADD(v, 0x118, 0xf);   // row_shr:8   -> lane 15 of each row holds the row total
    RSRL_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    RSRL_DPP_ADD(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
    return v;
}

// total broadcast to every lane
__device__ __forceinline__ float wave_sum_all(float v) {
    return __shfl(wave_sum_dpp_to_lane63(v), 63, 64);
}

// geometry of the basis that is not a template parameter
struct BasisGeom { int F; int tiles_per_dim; };

// 64-bit fixed point for every cross-learner sum of the shared-W modes: a term is scaled by the exact power of two 1/lsb,
// lsb = 2^(floor(log2 |lr|) - 28), clamped to +-2^42 and rounded to an integer; integer sums are exact whatever order the
// atomics retire in, and convert back with one rounding -- reproducible run to run and restated exactly by the oracle.
struct FxScale {
    float lsb, inv_lsb;
    __host__ __device__ __forceinline__ explicit FxScale(float lr) {
        uint32_t u = __builtin_bit_cast(uint32_t, lr);
        const uint32_t eb = (u >> 23) & 0xffu;
        const uint32_t ex = (eb < 30u ? 30u : eb) - 28u;
        lsb = __builtin_bit_cast(float, ex << 23); inv_lsb = __builtin_bit_cast(float, (254u - ex) << 23);
    }
};
// terms clamped so far on this device (a translation-unit-local symbol: every unit whose kernels quantise has a reader, ctx.hpp RSRL_DEFINE_FX_READER;
// rsrl_hip_fx_saturations adds them up).  A clamped term means the shared-W update departed from W += lr*e*phi: never silently.
static __device__ unsigned int g_fx_saturations;
__device__ __forceinline__ unsigned long long fx_quantise(float v, float inv_lsb) {
    const float raw = v * inv_lsb;
    const float sc = __builtin_amdgcn_fmed3f(raw, -4.398046511104e12f, 4.398046511104e12f);    // +-2^42: no wrap-around
    if (__builtin_expect(!(fabsf(raw) <= 4.398046511104e12f), 0)) atomicAdd(&g_fx_saturations, 1u);  // (NaN counts: it was clamped too)
    return (unsigned long long)(long long)rintf(sc);
}
__device__ __forceinline__ void fx_add(long long* p, unsigned long long q) { atomicAdd(reinterpret_cast<unsigned long long*>(p), q); }

template <int DOMAIN, int ORDER>
struct FourierModel {
    using Dom = Domain<DOMAIN>;
    using Bas = FourierReg<DOMAIN, ORDER>;
    static constexpr int D = Dom::D, A = Dom::A, F = Bas::F;
    static constexpr bool kDense = true, kSparse = false;
    struct Feat { float phi[F]; };
    __device__ static __forceinline__ void features(const float (&s)[D], const BasisGeom&, Feat& ft) { Bas::project(s, ft.phi); }
    __device__ static __forceinline__ int64_t widx(const Common& c, int64_t wi, int b, int f) {
        return ((int64_t)(b * F + f)) * c.w_stride + wi * c.w_ls;
    }
    __device__ static __forceinline__ void q_all(const Common& c, int64_t wi, const BasisGeom&, const Feat& ft, float (&q)[A]) {
        q_from_mem<A, F>(c.W, c.w_stride, wi * c.w_ls, ft.phi, q);
    }
    __device__ static __forceinline__ float q_index(const Common& c, int64_t wi, const BasisGeom&, const Feat& ft, int a) {
        constexpr int P = RSRL_DOT_SPLIT;
        float acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f % P] = fmaf(ft.phi[f], c.W[widx(c, wi, a, f)], acc[f % P]);
        return combine_partials<P>(acc);
    }
    __device__ static __forceinline__ void update(const Common& c, int64_t wi, const BasisGeom&, const Feat& ft, int a, float scale) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int64_t j = widx(c, wi, a, f);
            c.W[j] = fmaf(scale, ft.phi[f], c.W[j]);
        }
    }
    __device__ static __forceinline__ void write_features(const BasisGeom&, const Feat& ft, int64_t Mn, int64_t i,
                                                          float* __restrict__ fout, int32_t* __restrict__) {
#pragma unroll
        for (int f = 0; f < F; ++f) fout[(int64_t)f * Mn + i] = ft.phi[f];
    }
    __host__ __device__ static constexpr int F_or_1() { return F; }
    // Q(s,.) from a copy of the shared W[A][F] in LDS (uniform addresses: broadcast reads); same summation order as q_from_mem
    __device__ static __forceinline__ void q_all_lds(const float* __restrict__ shw, const Feat& ft, float (&q)[A]) {
        constexpr int P = RSRL_DOT_SPLIT;
#pragma unroll
        for (int b = 0; b < A; ++b) {
            float acc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f % P] = fmaf(ft.phi[f], shw[b * F + f], acc[f % P]);
            q[b] = combine_partials<P>(acc);
        }
    }
    // dW has the shared layout [A][F]
    // must be called by ALL lanes of the wave (uniform control flow); lanes without work pass valid = false
    __device__ static __forceinline__ void accumulate(long long* __restrict__ fx, const BasisGeom&, const Feat& ft, int a, float scale,
                                                      bool valid, float inv_lsb) {
        if (valid) {
#pragma unroll
            for (int f = 0; f < F; ++f) fx_add(&fx[a * F + f], fx_quantise(scale * ft.phi[f], inv_lsb));
        }
    }
};

// Dense grid tile coder (the build's deterministic definition; lfa's TileCoding is hashed
// and never instantiated by the reference).  fp32, non-fused ops, identical to oracle/rsrl_oracle.c:orc_tile_indices
// => indices are bit-exact:
//   s~_i = (s_i - lo_i)/(hi_i - lo_i);  u_i = s~_i*(B-1);  off_i(t) = ((t*(2i+1)) mod T)/T
//   cell_i = clamp((int)floorf(u_i + off_i), 0, B-1);  idx(t) = t*B^D + sum_i cell_i*B^i
template <int DOMAIN, int T>
struct TileModel {
    using Dom = Domain<DOMAIN>;
    static constexpr int D = Dom::D, A = Dom::A;
    static constexpr bool kDense = false, kSparse = true;
    static constexpr int kT = T, kDomain = DOMAIN;
    struct Feat { int idx[T]; };
    __host__ __device__ static constexpr int F_or_1() { return 1; }
    __device__ static __forceinline__ void q_all_lds(const float*, const Feat&, float (&)[A]) {}
    __device__ static __forceinline__ void features(const float (&s)[D], const BasisGeom& g, Feat& ft) {
        const int B = g.tiles_per_dim;
        int BD = 1;
#pragma unroll
        for (int i = 0; i < D; ++i) BD *= B;
        float u[D];
        static_for<0, D>([&](auto Ii) {
            constexpr int i = Ii;
            constexpr float lo = (float)Dom::lo_d(i), hi = (float)Dom::hi_d(i);
            const float num = s[i] - lo;
            constexpr float den = hi - lo;
            const float sc = div_const(num, den);                       // num / den, the same bits (device_core.hpp div_const)
            u[i] = sc * (float)(B - 1);
        });
        // cell = clamp((int)floorf(v), 0, B-1) == (int)floorf(clamp(v, 0, B-1)) for every v (the bounds are integers; a NaN gives 0 either
        // way: v_med3_f32 returns the minimum then): one v_med3_f32 + one v_cvt_flr_i32_f32 instead of floor, convert, max, min
        const float top = (float)(B - 1);
        static_for<0, T>([&](auto Tt) {
            constexpr int t = Tt;
            int lin = 0, stride = 1;
            static_for<0, D>([&](auto Ii) {
                constexpr int i = Ii;
                constexpr float off = (float)((t * (2 * i + 1)) % T) / (float)T;
                const float v = u[i] + off;
                const int cell = floor_to_int(__builtin_amdgcn_fmed3f(v, 0.0f, top));
                lin = (int)__umul24((unsigned)cell, (unsigned)stride) + lin;          // cell < 64, stride <= 2^24: v_mad_u32_u24
                stride *= B;
            });
            ft.idx[t] = t * BD + lin;
        });
    }
    __device__ static __forceinline__ int64_t widx(const Common& c, int64_t wi, const BasisGeom& g, int f, int b) {
        return (wi * (int64_t)g.F + f) * A + b;
    }
    __device__ static __forceinline__ void q_all(const Common& c, int64_t wi, const BasisGeom& g, const Feat& ft, float (&q)[A]) {
#pragma unroll
        for (int b = 0; b < A; ++b) q[b] = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int b = 0; b < A; ++b) q[b] = q[b] + c.W[widx(c, wi, g, ft.idx[t], b)];      // acc + w, tilings in order
    }
    __device__ static __forceinline__ float q_index(const Common& c, int64_t wi, const BasisGeom& g, const Feat& ft, int a) {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t) acc = acc + c.W[widx(c, wi, g, ft.idx[t], a)];
        return acc;
    }
    // Q(s,.) from ONE SHARED table: the table is addressed through a buffer descriptor (wave-uniform base in SGPRs) + a 32-bit byte
    // offset per lane -- one shift and one A-dword buffer load per tiling, no 64-bit address arithmetic (the per-learner form above
    // spends ~6 VALU instructions per gather on it).  Same loads, same order of additions: the same bits.
    // Precondition (host): the table is smaller than 2 GiB.
    __device__ static __forceinline__ void gather_shared(const float* __restrict__ W, const BasisGeom& g, const Feat& ft, float (&w)[T][A]) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, g.F * A * 4, 0x00020000);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int off = ft.idx[t] * (A * 4);
            // (the builtin's result goes through __builtin_bit_cast: assigned to an int vector of the same size it is silently narrowed to
            // its first element -- a one-dword load)
            if constexpr (A == 2) {
                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2v v = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
                w[t][0] = v.x; w[t][1] = v.y;
            } else if constexpr (A == 3) {
                typedef float f3v __attribute__((ext_vector_type(3)));
                const f3v v = __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0));
                w[t][0] = v.x; w[t][1] = v.y; w[t][2] = v.z;
            } else {
#pragma unroll
                for (int b = 0; b < A; ++b) w[t][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4 * b, 0, 0));
            }
        }
    }
    __device__ static __forceinline__ void sum_gathered(const float (&w)[T][A], float (&q)[A]) {
#pragma unroll
        for (int b = 0; b < A; ++b) q[b] = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int b = 0; b < A; ++b) q[b] = q[b] + w[t][b];          // acc + w, tilings in order
    }
    __device__ static __forceinline__ void q_all_shared(const float* __restrict__ W, const BasisGeom& g, const Feat& ft, float (&q)[A]) {
        float w[T][A];
        gather_shared(W, g, ft, w);
        sum_gathered(w, q);
    }
    __device__ static __forceinline__ void update(const Common& c, int64_t wi, const BasisGeom& g, const Feat& ft, int a, float scale) {
#pragma unroll
        for (int t = 0; t < T; ++t) c.W[widx(c, wi, g, ft.idx[t], a)] += scale;               // indices of distinct tilings never collide
    }
    __device__ static __forceinline__ void write_features(const BasisGeom&, const Feat& ft, int64_t Mn, int64_t i,
                                                          float* __restrict__, int32_t* __restrict__ iout) {
#pragma unroll
        for (int t = 0; t < T; ++t) iout[(int64_t)t * Mn + i] = ft.idx[t];
    }
    // Device-wide form (rsrl_hip_handle on a shared table; the driver loop when a tiling's slice does not fit LDS): the learner's
    // term as ONE integer, added to its T entries with device atomics -- the same integers k_tile_scatter adds through LDS.
    __device__ static __forceinline__ void accumulate(long long* __restrict__ fx, const BasisGeom&, const Feat& ft, int a, float scale,
                                                      bool valid, float inv_lsb) {
        if (!valid) return;
        const unsigned long long term = fx_quantise(scale, inv_lsb);     // the same integer goes to all T entries
#pragma unroll
        for (int t = 0; t < T; ++t) fx_add(&fx[ft.idx[t] * A + a], term);
    }
    // (The driver loop privatises each tiling's slice of the delta table in LDS: k_tile_scatter, kernels_util.hip.)  The accumulators are
    // 64-bit FIXED-POINT integers, not floats: ds_add_f32 retires ONE LANE PER ~3 CYCLES whatever the addresses are (193 cycles per
    // wave-instruction even for 64 conflict-free addresses, profiles/r02_ubench_lds_atomic.txt), ds_add_u64 costs 6 cycles for distinct
    // addresses and 2 per duplicate of the most crowded one.  A term lr*e is scaled by the power of two 1/lsb (exact) and rounded to an
    // integer: the sum is then EXACT and order-independent, and converting it back rounds once (in k_apply_rep).
    // lsb = 2^(floor(log2 lr) - 28): |e| up to 2^12 and 2^20 learners on one entry fit 63 bits; a term keeps its full 24-bit mantissa
    // down to |e| = 2^-4 and an absolute resolution of lr * 2^-28 below that.  The device-wide delta table is fixed-point too (dW64:
    // n_rep copies of cells*T*A 64-bit words): the update W += fl(sum * lsb) is bitwise reproducible from run to run and restated
    // exactly by the oracle (tests: bit-identical weights).
};

// Fourier basis of ANY order 1..7 on any domain, one thread per learner, features generated on the fly from the
// per-dimension tables (runtime digits => the tables are indexed dynamically and live in scratch/LDS): the
// catch-all for the (domain, order) pairs that have neither a register-family nor a wave-family kernel.
// Same arithmetic as FourierReg (tables by the angle-addition chain, complex product in dimension order, the
// 4-way interleaved dot product over the reference feature order).  W f32[A][F][Nw], learner index fastest.
template <int DOMAIN>
struct FourierGenericModel {
    using Dom = Domain<DOMAIN>;
    static constexpr int D = Dom::D, A = Dom::A;
    static constexpr bool kDense = false, kSparse = false;      // no register-resident phi: the shared-W block reduction is not available
    __host__ __device__ static constexpr int F_or_1() { return 1; }
    struct Feat { float ct[D][8], st[D][8]; };
    __device__ static __forceinline__ void q_all_lds(const float*, const Feat&, float (&)[A]) {}
    __device__ static __forceinline__ void features(const float (&s)[D], const BasisGeom& g, Feat& ft) {
        const int order = g.tiles_per_dim;     // BasisGeom::tiles_per_dim carries the Fourier order for this model
        static_for<0, D>([&](auto Dd) {
            constexpr int d = Dd;
            constexpr float lo = (float)Dom::lo_d(d), hi = (float)Dom::hi_d(d);
            constexpr float inv = 1.0f / (hi - lo);                 // as FourierTables::build
            const float sc = (s[d] - lo) * inv;
            ft.ct[d][0] = 1.0f; ft.st[d][0] = 0.0f;
            sincospi01(sc, ft.st[d][1], ft.ct[d][1]);
            for (int n = 2; n <= order; ++n) {
                ft.ct[d][n] = fmaf(-ft.st[d][n - 1], ft.st[d][1], ft.ct[d][n - 1] * ft.ct[d][1]);
                ft.st[d][n] = fmaf(ft.ct[d][n - 1], ft.st[d][1], ft.st[d][n - 1] * ft.ct[d][1]);
            }
        });
    }
    // phi of reference feature f (f = F-1 is the constant)
    __device__ static __forceinline__ float phi_at(const BasisGeom& g, const Feat& ft, int f) {
        if (f == g.F - 1) return 1.0f;
        const int n1 = g.tiles_per_dim + 1;
        int k = f + 1, c[D];
#pragma unroll
        for (int d = D - 1; d >= 0; --d) { c[d] = k % n1; k /= n1; }
        float re = ft.ct[0][c[0]], im = ft.st[0][c[0]];
#pragma unroll
        for (int d = 1; d < D; ++d) {
            const float cr = ft.ct[d][c[d]], sr = ft.st[d][c[d]];
            const float nre = fmaf(-im, sr, re * cr), nim = fmaf(re, sr, im * cr);
            re = nre; im = nim;
        }
        return re;
    }
    __device__ static __forceinline__ int64_t widx(const Common& c, int64_t wi, const BasisGeom& g, int b, int f) {
        return ((int64_t)b * g.F + f) * c.w_stride + wi;
    }
    __device__ static __forceinline__ void q_all(const Common& c, int64_t wi, const BasisGeom& g, const Feat& ft, float (&q)[A]) {
        float acc[A][4];
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[b][p] = 0.0f;
        for (int f0 = 0; f0 < g.F; f0 += 4) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int f = f0 + p;
                if (f < g.F) {
                    const float ph = phi_at(g, ft, f);
#pragma unroll
                    for (int b = 0; b < A; ++b) acc[b][p] = fmaf(ph, c.W[widx(c, wi, g, b, f)], acc[b][p]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < A; ++b) q[b] = (acc[b][0] + acc[b][1]) + (acc[b][2] + acc[b][3]);
    }
    __device__ static __forceinline__ float q_index(const Common& c, int64_t wi, const BasisGeom& g, const Feat& ft, int a) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int f0 = 0; f0 < g.F; f0 += 4) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int f = f0 + p;
                if (f < g.F) acc[p] = fmaf(phi_at(g, ft, f), c.W[widx(c, wi, g, a, f)], acc[p]);
            }
        }
        return (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    __device__ static __forceinline__ void update(const Common& c, int64_t wi, const BasisGeom& g, const Feat& ft, int a, float scale) {
        for (int f = 0; f < g.F; ++f) {
            const int64_t j = widx(c, wi, g, a, f);
            c.W[j] = fmaf(scale, phi_at(g, ft, f), c.W[j]);
        }
    }
    __device__ static __forceinline__ void write_features(const BasisGeom& g, const Feat& ft, int64_t Mn, int64_t i,
                                                          float* __restrict__ fout, int32_t* __restrict__) {
        for (int f = 0; f < g.F; ++f) fout[(int64_t)f * Mn + i] = phi_at(g, ft, f);
    }
    // the shared layout [A][F] in fixed point (used by rsrl_hip_handle in shared mode only)
    __device__ static __forceinline__ void accumulate(long long* __restrict__ fx, const BasisGeom& g, const Feat& ft, int a, float scale,
                                                      bool valid, float inv_lsb) {
        if (valid)
            for (int f = 0; f < g.F; ++f) fx_add(&fx[a * g.F + f], fx_quantise(scale * phi_at(g, ft, f), inv_lsb));
    }
};

// ---------------------------------------------------------------------------------------
// generic trait-granular kernels
// ---------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void load_state(const float* __restrict__ states, int64_t M_, int64_t i, float (&s)[M::D]) {
#pragma unroll
    for (int d = 0; d < M::D; ++d) s[d] = states[(int64_t)d * M_ + i];
}

// per-episode Domain::default() + initial policy.sample     examples/q_learning.rs:37-38
template <class M>
__global__ __launch_bounds__(kBlock) void k_reset(Common c, BasisGeom g, uint64_t t) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_envs) return;
    float s[D]; M::Dom::reset(s);
    typename M::Feat ft; float q[A];
    M::features(s, g, ft);
    M::q_all(c, c.shared ? 0 : i, g, ft, q);
    const U4 x = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_INIT);
#pragma unroll
    for (int d = 0; d < D; ++d) c.state[(int64_t)d * c.n_envs + i] = s[d];
    PolicyParams pol = c.pol;
    learner_eps_load(c, i, pol);                     // per-learner epsilon, when configured (a reset is not an episode end: no decay)
    c.action[i] = policy_sample<A>(pol, q, x);
    c.ep_step[i] = 0;
}

enum : int { QOP_EVALUATE = 0, QOP_FIND_MAX = 1, QOP_SAMPLE = 2, QOP_MODE = 3, QOP_PROBS = 4, QOP_FEATURES = 5,
              QOP_FIND_MIN = 6,      // Enumerable::find_min                                   core.rs:86-94
              QOP_EXPECTED = 7,      // Enumerable::expected_value(ps), ps = fin f32[A][M]     core.rs:107-116
              QOP_PROB_SA = 8,
              QOP_SAMPLE_STEP = 9,   // Policy::sample as the DRIVER LOOP draws it: `call` = the batch-step, stream BLK_STEP (rsrl_hip_policy_sample with
              QOP_SAMPLE_INIT = 10 };  // states = NULL: the ctx's own envs) -- before the first handle: the initial sample's stream BLK_INIT     // Function<(S, A)> of the policy, a = iin i32[M]         greedy.rs:46-60, epsilon_greedy.rs:49-63,
                                     //                                                        softmax.rs:84-92 (the raw action value), random.rs:28-32
// the tail every family's qop kernel shares: q -> the requested output of learner i
template <int A>
__device__ __forceinline__ void qop_finish(const Common& c, int op, const float (&q)[A], int64_t Mn, int64_t i, uint64_t call,
                                           float* __restrict__ fout, int32_t* __restrict__ iout, const float* __restrict__ fin,
                                           const int32_t* __restrict__ iin) {
    PolicyParams pol = c.pol;
    learner_eps_load(c, i, pol);                     // item i is evaluated with learner i's policy object (its own epsilon, when configured)
    if (op == QOP_EVALUATE) {
#pragma unroll
        for (int b = 0; b < A; ++b) fout[(int64_t)b * Mn + i] = q[b];
    } else if (op == QOP_FIND_MAX || op == QOP_FIND_MIN) {
        float v; const int bi = op == QOP_FIND_MAX ? find_max<A>(q, v) : find_min<A>(q, v);
        if (iout) iout[i] = bi;
        if (fout) fout[i] = v;
    } else if (op == QOP_SAMPLE || op == QOP_SAMPLE_STEP || op == QOP_SAMPLE_INIT) {
        const U4 x = draw(c.seed, (uint32_t)(c.env_offset + i), call, op == QOP_SAMPLE ? BLK_API : (op == QOP_SAMPLE_STEP ? BLK_STEP : BLK_INIT));
        iout[i] = policy_sample<A>(pol, q, x);
    } else if (op == QOP_MODE) {
        iout[i] = policy_mode<A>(pol, q);
    } else if (op == QOP_EXPECTED) {
        float p[A];
#pragma unroll
        for (int b = 0; b < A; ++b) p[b] = fin[(int64_t)b * Mn + i];
        fout[i] = expected_value<A>(q, p);
    } else if (op == QOP_PROB_SA) {
        fout[i] = policy_eval_sa<A>(pol, q, clamp_action<A>(iin[i]));
    } else {
        float p[A]; policy_probs<A>(pol, q, p);
#pragma unroll
        for (int b = 0; b < A; ++b) fout[(int64_t)b * Mn + i] = p[b];
    }
}

// Function<(S,)>::evaluate / Enumerable::find_max / Policy::{sample,mode} / policy probabilities / basis.project
//   fa/linear.rs:303-311, core.rs:96-105, policies/mod.rs:65-78
template <class M>
__global__ __launch_bounds__(kBlock) void k_qop(Common c, BasisGeom g, int op, const float* __restrict__ states, int64_t Mn,
                                                uint64_t call, float* __restrict__ fout, int32_t* __restrict__ iout,
                                                const float* __restrict__ fin, const int32_t* __restrict__ iin) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    float s[D]; load_state<M>(states, Mn, i, s);
    typename M::Feat ft;
    M::features(s, g, ft);
    if (op == QOP_FEATURES) { M::write_features(g, ft, Mn, i, fout, iout); return; }
    float q[A];
    M::q_all(c, c.shared ? 0 : i, g, ft, q);
    qop_finish<A>(c, op, q, Mn, i, call, fout, iout, fin, iin);
}

// Handler<&Transition>::handle on caller-supplied transitions (teacher forcing / drop-in use).
// per-env weights: learner m's column is updated in place.
// shared weights : lr*e*phi(s) is accumulated into the fixed-point delta table (exact, order-independent); k_fx_finalize turns
//                  it into dW and k_apply_dw applies it -- all M errors are computed against the same W_t (synchronous
//                  mini-batch rule).
template <class M>
__global__ __launch_bounds__(kBlock) void k_handle(Common c, BasisGeom g, const float* __restrict__ from, const int32_t* __restrict__ act,
                                                   const float* __restrict__ rew, const float* __restrict__ to,
                                                   const uint8_t* __restrict__ termf, int64_t Mn, uint64_t t,
                                                   float* __restrict__ td_out, long long* __restrict__ fx) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < Mn;
    const bool shared = c.shared != 0;
    typename M::Feat fs;
    int a = 0;
    float scale = 0.0f;
    if (valid) {
        const int64_t wi = shared ? 0 : i;
        float s[D], ns[D];
        load_state<M>(from, Mn, i, s);
        load_state<M>(to, Mn, i, ns);
        a = clamp_action<A>(act[i]);
        const float r = rew[i];
        const bool term = termf[i] != 0;
        typename M::Feat fn;
        M::features(s, g, fs);
        M::features(ns, g, fn);
        float q_s[A], q_n[A];
        M::q_all(c, wi, g, fs, q_s);
        M::q_all(c, wi, g, fn, q_n);
        U4 xin = U4{0, 0, 0, 0};
        if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_INNER);
        float e;
        PolicyParams apol = c.apol;
        if (c.apol_same) learner_eps_load(c, i, apol);             // the agent shares the behaviour policy object: this learner's epsilon
        const float delta = td_dispatch<A>(c.alg, apol, q_s, a, q_n, r, term, xin, e);
        scale = c.alg.lr * e;
        if (!shared) M::update(c, wi, g, fs, a, scale);
        if (td_out) td_out[i] = delta;
    }
    if (shared) M::accumulate(fx, g, fs, a, scale, valid, FxScale(c.alg.lr).inv_lsb);
}

// Domain::rollout(|s| policy.mode(s), Some(limit)) + n_states, weights read from memory      lib.rs:448-479, :340
// The Trajectory itself (lib.rs:334-409) when the caller asks for it: tr.states f32[step_limit][D][Mn] -- row 0 = `start`, row k =
// the observation of steps[k-1] --, tr.actions i32[step_limit-1][Mn] and tr.rewards f32[step_limit-1][Mn] = steps[k].1 / .2,
// tr.terminal u8[Mn] = the last observation is Observation::Terminal.  Rows past n_states are left untouched.
struct TrajOut { float* states; int32_t* actions; float* rewards; uint8_t* terminal; int64_t Mn; };
template <int D>
__device__ __forceinline__ void traj_record(const TrajOut& tr, int64_t i, int64_t k, const float (&s)[D], int a, float r) {
    // transition k (0-based): the observation it arrived at is state row k+1
    if (tr.states) {
#pragma unroll
        for (int d = 0; d < D; ++d) tr.states[((k + 1) * D + d) * tr.Mn + i] = s[d];
    }
    if (tr.actions) tr.actions[k * tr.Mn + i] = a;
    if (tr.rewards) tr.rewards[k * tr.Mn + i] = r;
}
template <class M>
__global__ __launch_bounds__(kBlock) void k_rollout(Common c, BasisGeom g, int64_t step_limit, uint32_t* __restrict__ n_states,
                                                    float* __restrict__ total_reward, int64_t Mn, TrajOut tr, RolloutPolicy rp) {
    constexpr int D = M::D, A = M::A;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mn) return;
    const int64_t wi = c.shared ? 0 : i;
    const uint32_t gid = (uint32_t)(c.env_offset + i);
    uint64_t kk = 0;                                 // action selections so far
    float s[D]; M::Dom::reset(s);
    if (tr.states) {
#pragma unroll
        for (int d = 0; d < D; ++d) tr.states[(int64_t)d * tr.Mn + i] = s[d];
    }
    typename M::Feat ft; float q[A], r, tot = 0.0f;
    M::features(s, g, ft); M::q_all(c, wi, g, ft, q);
    int a = rollout_action<A>(c.pol, rp, q, c.seed, gid, kk++);
    bool term = M::Dom::step(s, a, r);               // the first step is taken eagerly (lib.rs:457-459)
    int64_t steps = 0;
    while (steps < step_limit - 1) {
        traj_record<D>(tr, i, steps, s, a, r);
        steps += 1; tot += r;
        if (term) break;                             // successors() stops after a Terminal observation
        if (steps >= step_limit - 1) break;
        M::features(s, g, ft); M::q_all(c, wi, g, ft, q);
        a = rollout_action<A>(c.pol, rp, q, c.seed, gid, kk++);
        term = M::Dom::step(s, a, r);
    }
    n_states[i] = (uint32_t)(steps + 1);
    if (total_reward) total_reward[i] = tot;
    if (tr.terminal) tr.terminal[i] = (steps > 0 && term) ? 1 : 0;     // (step_limit = 1: no transition is kept)
}

// ---------------------------------------------------------------------------------------
// Driver loop, generic form: weights stay in memory.  Used for
//   * tile coding with per-learner tables (256 KiB per learner at 8 x 8^4 x 2: cannot live in registers);
//   * shared weights (both bases): phase A / apply / phase C per batch-step, because every learner's error is
//     taken against the same W_t and every learner then samples with the same W_{t+1}.
// ---------------------------------------------------------------------------------------

// per-learner weights, n_steps per launch, W in memory (no cross-learner dependence => fully fused)
template <class M>
__global__ __launch_bounds__(kBlock) void k_train_mem(Common c, BasisGeom g, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    constexpr int D = M::D, A = M::A;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        float s[D]; load_state<M>(c.state, N, i, s);
        int a = c.action[i];
        uint32_t ep = c.ep_step[i];
        typename M::Feat fs, fn;
        M::features(s, g, fs);
        float facc_abs = 0.0f, facc_r = 0.0f;
        PolicyParams pol = c.pol, apol = c.apol;
        const bool esched = c.eps != nullptr;        // per-learner epsilon schedule (examples/sarsa_lambda.rs:68)
        if (esched) { learner_eps_load(c, i, pol); if (c.apol_same) apol = pol; }
        for (int k = 0; k < n_steps; ++k) {
            const uint64_t t = t0 + (uint64_t)k;
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            const bool term = M::Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            if (term) M::Dom::reset(ns);
            M::features(ns, g, fn);
            float q_s[A], q_n[A];
            M::q_all(c, i, g, fs, q_s);
            M::q_all(c, i, g, fn, q_n);
            U4 xin = U4{0, 0, 0, 0};
            if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
            float e;
            const float delta = td_dispatch<A>(c.alg, apol, q_s, a, q_n, r, term, xin, e);
            M::update(c, i, g, fs, a, c.alg.lr * e);
            M::q_all(c, i, g, fn, q_n);                              // UPDATED weights
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            if (esched) { learner_eps_step(c, term | trunc, pol); if (c.apol_same) apol = pol; }
            int na = policy_sample<A>(pol, q_n, x);
            facc_abs += fabsf(delta); facc_r += r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                M::Dom::reset(ns);
                M::features(ns, g, fn);
                M::q_all(c, i, g, fn, q_n);
                const U4 xr = draw(c.seed, gid, t, BLK_RESET);
                na = policy_sample<A>(pol, q_n, xr);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
            fs = fn;
            a = na;
        }
        sum_abs = (double)facc_abs; sum_r = (double)facc_r;
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
        if (esched) c.eps[i] = pol.eps;
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// shared weights, phases C(t-1) + A(t) in ONE launch.  Both read the same weights W_t: phase C finishes the previous
// batch-step (policy.sample with the just-updated weights; finished episodes restart from Domain::default()), phase A
// runs the transition and takes the TD error of this step against W_t and adds the learner's term lr*e*phi(s) to the
// mini-batch delta.  phi(s) and Q(s,.) are computed once and serve both phases.  do_c = 0 on the first step of a
// train call (the previous call already ran its phase C; bit 1 of do_c: record Q(lambda)'s cut).  The env state becomes s'; flags[i] bit0 = terminal,
// bit1 = truncated, consumed by the next phase C.
//   tile coding : the learner's term and its T slice-relative entries are handed to k_tile_scatter (per-tiling slice of the delta table
//                 privatised in LDS, one device atomic per touched entry into one of n_rep copies of the table; k_apply_rep sums them);
//                 a slice too large for LDS (keys == nullptr): one device atomic per learner and tiling.
//   (the register-resident dense bases have a kernel of their own, k_shared_step below, and are not launched through this one)
template <class M>
__global__ __launch_bounds__(kBlock) void k_shared_ca(Common c, BasisGeom g, uint64_t t, int do_c, float* __restrict__ dW_base,
                                                      uint8_t* __restrict__ flags,
                                                      DevStats* __restrict__ stats, int n_rep, int64_t rep_stride,
                                                      const uint64_t* __restrict__ t_dev, uint16_t* __restrict__ keys = nullptr,
                                                      float* __restrict__ terms = nullptr) {
    if (t_dev) t += *t_dev;            // graph replay: the batch-step counter lives on the device, t is the node's offset
    if (c.dyn) { c.pol = c.dyn->pol; c.apol = c.dyn->apol; }
    // tile coding: the delta table is replicated n_rep times and block b adds into copy b % n_rep -- device atomics on one
    // 128-B line serialise at ~11 ns each and the learners crowd into a few lines; k_apply_rep sums the copies
    long long* __restrict__ fx = reinterpret_cast<long long*>(dW_base) + (int64_t)(blockIdx.x % (unsigned)n_rep) * rep_stride;
    constexpr int D = M::D, A = M::A;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    // dense basis: the shared W (A*F floats) is staged in LDS once per block -- with W in global memory every lane issues
    // the same 2*A*F uniform-address loads per step and the kernel is bound by VMEM issue, not arithmetic
    __shared__ __attribute__((aligned(16))) float sh_w[M::A * M::F_or_1()];
    if constexpr (M::kDense) {
        for (int j = threadIdx.x; j < M::A * M::F_or_1(); j += blockDim.x) sh_w[j] = c.W[j];
        __syncthreads();
    }
    typename M::Feat fs;
    int a = 0;
    float scale = 0.0f;
    if (i < N) {
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        float s[D], ns[D], q_s[A];
        uint32_t ep = c.ep_step[i];
        bool done = false;
        if (do_c & 1) done = (flags[i] & 3) != 0;
        if (done) { M::Dom::reset(s); ep = 0; }
        else load_state<M>(c.state, N, i, s);
        M::features(s, g, fs);
        float r = 0.0f; bool term = false;
        typename M::Feat fn;
        float q_n[A];
        if constexpr (M::kDense) M::q_all_lds(sh_w, fs, q_s);
        else if constexpr (M::kSparse) M::q_all_shared(c.W, g, fs, q_s);
        else M::q_all(c, 0, g, fs, q_s);
        if (do_c & 1) {                                                 // ---- phase C of batch-step t-1
            const U4 x = draw(c.seed, gid, t - 1, BLK_STEP);
            a = policy_sample<A>(c.pol, q_s, x);
        } else {
            a = c.action[i];
        }
        // ---- phase A of batch-step t
#pragma unroll
        for (int d = 0; d < D; ++d) ns[d] = s[d];
        term = M::Dom::step(ns, a, r);
        M::features(ns, g, fn);
        if constexpr (M::kDense) M::q_all_lds(sh_w, fn, q_n);
        else if constexpr (M::kSparse) M::q_all_shared(c.W, g, fn, q_n);
        else M::q_all(c, 0, g, fn, q_n);
        ep += 1;
        const bool trunc = !term && cap > 0 && ep >= cap;
        U4 xin = U4{0, 0, 0, 0};
        if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
        float e;
        const float delta = td_dispatch<A>(c.alg, c.apol, q_s, a, q_n, r, term, xin, e);
        scale = c.alg.lr * e;
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = ns[d];
        c.action[i] = a;
        c.ep_step[i] = ep;
        // (bit 2, asked for by do_c bit 1: Q(lambda)'s cut -- the action taken was not argmax_first of Q(s,.), q_lambda.rs:62-66 -- for the trace kernel)
        flags[i] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0) | (((do_c & 2) && a != argmax_first<A>(q_s)) ? 4 : 0));
        sum_abs = (double)fabsf(delta); sum_r = (double)r;
        if (term || trunc) { n_ep = 1; n_trunc = trunc ? 1 : 0; sum_len = ep; }
    } else {
        if constexpr (M::kDense) {
#pragma unroll
            for (int f = 0; f < M::F; ++f) fs.phi[f] = 0.0f;
        }
    }
    if constexpr (!M::kDense) {
        if constexpr (M::kSparse) {
            if (keys) {
                // the scatter has a kernel of its own (k_tile_scatter): this one hands over, per learner, its term lr*e (rounded to
                // fixed point there) and the T slice-relative entries it goes to (16 bits each: a slice has at most 8 192 entries)
                if (i < N) {
                    constexpr int T = (int)(sizeof(fs.idx) / sizeof(fs.idx[0]));
                    const int cells = g.F / T;
#pragma unroll
                    for (int tt = 0; tt < T; ++tt) keys[(int64_t)tt * N + i] = (uint16_t)((fs.idx[tt] - tt * cells) * A + a);
                    terms[i] = scale;
                }
            } else {
                M::accumulate(fx, g, fs, a, scale, i < N, FxScale(c.alg.lr).inv_lsb);
            }
        } else {
            M::accumulate(fx, g, fs, a, scale, i < N, FxScale(c.alg.lr).inv_lsb);
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// ---------------------------------------------------------------------------------------
// Shared weights, dense basis: ONE launch per batch-step (k_shared_step).
// A dependent kernel costs ~4 us on this machine whatever it does, so the delta reduction no longer has a kernel of its
// own: every block of batch-step t first folds the delta of batch-step t-1 into the weights itself (W_t = W_{t-1} + delta:
// identical in every block), keeps W_t in LDS for both phases, and block 0 writes it out for the next launch (two W buffers in
// ping-pong: a launch never writes what a block of the same launch may still read).  The kernel boundary is the only
// synchronisation.  fold = 0: nothing to fold (first step of a train call; peer-exchange mode, where the exchange kernel between
// the launches applies the sum itself); fold = 1: this rank's own table (single rank); fold = 2: the float delta dW_in that
// finalize -> all-reduce left behind between the launches (RCCL mode).
//   mode bit 0: phase C of the previous batch-step (policy.sample with W_t, episode restarts)
//   mode bit 1: phase A of this batch-step (transition, TD error against W_t, the learner's term into this block's row)
// ---------------------------------------------------------------------------------------
// The mini-batch delta travels between launches as 64-BIT FIXED-POINT TABLES (as for shared tile coding): every block rounds
// its 108-entry partial sum to integers of lsb = 2^(floor(log2 lr) - 28) and adds them with one device atomic each into one
// of kTabRep copies of the table; the next launch's prologue adds the kTabRep copies (integers: exact, any order) and converts
// back with one rounding.  Folding 256 float rows in every block instead read 110 KB per block from L2 (28 MB per batch-step,
// 2.7 us of an 10.7 us step); the copies are 14 KB.  Three table sets rotate with the batch-step counter: step t accumulates
// into set t mod 3, folds set (t-1) mod 3 and clears set (t+1) mod 3 -- a launch never clears what a block of the same launch
// may still read or add to, and the kernel boundary is the only synchronisation.

// Block-level sum of the learners' terms of a shared dense approximator, the per-wave part: out[b][f] = sum over the wave's 64
// learners k of [a_k == b] * (scale_k * phi_k[f]).  One learner's term is a rank-1 update -- exactly v_mfma_f32_4x4x1_16b_f32: 16
// independent 4x4 blocks, K = 1, D[blk][i][j] += A[blk][i] * B[blk][j].  With i = action and 4*blk + j = feature (= the lane),
// learner k's A operand is the indicator [a_k == lane % 4] (a bit of a ballot mask: v_bfe + v_cvt) and its B operand is its F terms
// ACROSS lanes -- the transpose of what the lanes hold, read back from the wave's own LDS tile (row = learner, padded to F + 1
// words: conflict-free both ways; all 64 reads issued before the first MFMA).  K = 1 means one exact product (the indicator is 0
// or 1) and one rounding per accumulation, in program order: the fp32 chain acc = fma(ind, v, acc), bit for bit (probe:
// scripts/ubench/mfma_4x4x1.hip).
// FOUR chains per wave (round 3): chain c takes the learners k = c mod 4 in ascending order, the wave's sum is
// (c0 + c1) + (c2 + c3) -- the convention of every dot product on this path.  One 64-long chain of DEPENDENT MFMAs cost 1.4 us
// per batch-step (two waves per SIMD, measured with the chain compiled out, scripts/gpu_exp_persist.sh): each MFMA waited for
// the one before it; four independent accumulators issue back to back.  The oracle restates the four chains
// (orc_run_train_shared_dev).
// The tile is FEATURE-major, tile[f][learner] with rows of 68 words: a lane writes its term of feature f next to its neighbours'
// (ds_write_b32, conflict-free) and reads FOUR learners' terms of its own feature with one ds_read_b128 (row stride 68 = 4 mod 64
// words: the lanes of a b128 service group land on disjoint banks) -- 16 wide reads per lane instead of 64 narrow ones.
constexpr int kRank1Stride = 68;
template <int A, int F>
__device__ __forceinline__ void wave_rank1_sum(float (*tile)[kRank1Stride] /* [F][68] of this wave, 16-byte aligned */, int lane, float scale,
                                               const float (&phi)[F], bool member, int a, float* __restrict__ out /* [A*F] in LDS */) {
    static_assert(F <= 64 && A <= 4, "features across the 64 lanes, actions across the 4 rows of an MFMA block");
#pragma unroll
    for (int f = 0; f < F; ++f) tile[f][lane] = scale * phi[f];
    unsigned long long m = 0;
#pragma unroll
    for (int b = 0; b < A; ++b) {
        const unsigned long long mb = __ballot(member && a == b);
        m = ((lane & 3) == b) ? mb : m;
    }
    const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
    const int fl = lane < F ? lane : F - 1;                     // lanes past the features belong to unused blocks
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) acc[ch] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
    float bv[64];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        const f4v v = *reinterpret_cast<const f4v*>(&tile[fl][4 * k4]);
        bv[4 * k4] = v.x; bv[4 * k4 + 1] = v.y; bv[4 * k4 + 2] = v.z; bv[4 * k4 + 3] = v.w;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const float av = (float)(((k < 32 ? mlo : mhi) >> (k & 31)) & 1u);
        acc[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv[k], acc[k & 3], 0, 0, 0);
    }
    const f4v tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (lane < F) {
#pragma unroll
        for (int b = 0; b < A; ++b) out[b * F + lane] = tot[b];
    }
}

constexpr int kTabRep = 16;
struct DeltaTab {
    long long *out, *zero; const long long* in; float lsb, inv_lsb;
    __device__ __forceinline__ DeltaTab(long long* tab, int n, float lr, uint64_t t) {
        const size_t set = (size_t)kTabRep * n;
        const unsigned r = (unsigned)(t % 3u);
        out = tab + set * r; in = tab + set * ((r + 2u) % 3u); zero = tab + set * ((r + 1u) % 3u);
        const uint32_t eb = (__float_as_uint(lr) >> 23) & 0xffu;
        const int ex = (int)(eb < 30u ? 30u : eb) - 28;
        lsb = __uint_as_float((uint32_t)ex << 23); inv_lsb = __uint_as_float((uint32_t)(254 - ex) << 23);
    }
    __device__ __forceinline__ unsigned long long quantise(float v) const { return fx_quantise(v, inv_lsb); }
};
template <class M, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_shared_step(Common c, BasisGeom g, uint64_t t, int mode, const float* __restrict__ W_in,
                                                        float* __restrict__ W_out, long long* __restrict__ tab, int fold,
                                                        const float* __restrict__ dW_in, uint8_t* __restrict__ flags,
                                                        DevStats* __restrict__ stats, const uint64_t* __restrict__ t_dev) {
    static_assert(M::kDense, "dense bases only");
    if (t_dev) t += *t_dev;
    if (c.dyn) { c.pol = c.dyn->pol; c.apol = c.dyn->apol; }
    constexpr int D = M::D, A = M::A, F = M::F, AF = A * F;
    const bool do_c = (mode & 1) != 0, do_a = (mode & 2) != 0;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    __shared__ __attribute__((aligned(16))) float sh_w[AF];
    // the learner's own loads go out FIRST: they do not depend on the weights, and their latency then overlaps the row loads
    // of the fold below instead of following them (two ~2 us round trips to memory the previous launch has just written)
    const int64_t il = i < N ? i : N - 1;
    float s_ld[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s_ld[d] = c.state[(int64_t)d * N + il];
    const uint32_t ep_ld = c.ep_step[il];
    const uint8_t flag_ld = do_c ? flags[il] : (uint8_t)0;
    const int a_ld = c.action[il];
    // ---- W_t = W_{t-1} + the previous batch-step's delta, in LDS (every block: integer sums => same bits everywhere)
    DeltaTab dt(tab, AF, c.alg.lr, t);
    {
        long long fsum = 0;
        if (fold == 1 && threadIdx.x < AF) {
            const long long* __restrict__ p = dt.in + threadIdx.x;
#pragma unroll
            for (int r = 0; r < kTabRep; ++r) fsum += p[r * AF];                 // kTabRep independent loads, one round trip
        }
        // fold = 1: this rank's own table (single rank); fold = 2: the delta as floats, already summed over the ranks by the exchange
        if (threadIdx.x < AF) {
            const float d = fold == 2 ? dW_in[threadIdx.x] : (float)fsum * dt.lsb;
            sh_w[threadIdx.x] = fold ? W_in[threadIdx.x] + d : W_in[threadIdx.x];
        }
        // the set the NEXT batch-step accumulates into (last read one launch ago) is cleared here, one copy per block
        for (int r = blockIdx.x; r < kTabRep; r += gridDim.x)
            if (threadIdx.x < AF) dt.zero[r * AF + threadIdx.x] = 0;
    }
    __syncthreads();
    if (W_out && blockIdx.x == 0)
        for (int j = threadIdx.x; j < AF; j += BLOCK) W_out[j] = sh_w[j];
    typename M::Feat fs;
    int a = 0;
    float scale = 0.0f;
    if (i < N) {
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        float s[D], ns[D], q_s[A];
        uint32_t ep = ep_ld;
        const bool done = do_c && flag_ld != 0;
        if (done) { M::Dom::reset(s); ep = 0; }
        else {
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = s_ld[d];
        }
        M::features(s, g, fs);
        M::q_all_lds(sh_w, fs, q_s);
        if (do_c) {                                                     // ---- phase C of batch-step t-1
            const U4 x = draw(c.seed, gid, t - 1, BLK_STEP);
            a = policy_sample<A>(c.pol, q_s, x);
        } else {
            a = a_ld;
        }
        if (do_a) {                                                     // ---- phase A of batch-step t
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            const bool term = M::Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            typename M::Feat fn;
            M::features(ns, g, fn);
            float q_n[A];
            M::q_all_lds(sh_w, fn, q_n);
            U4 xin = U4{0, 0, 0, 0};
            if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
            float e;
            const float delta = td_dispatch<A>(c.alg, c.apol, q_s, a, q_n, r, term, xin, e);
            scale = c.alg.lr * e;
#pragma unroll
            for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = ns[d];
            c.action[i] = a;
            c.ep_step[i] = ep;
            flags[i] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0));
            sum_abs = (double)fabsf(delta); sum_r = (double)r;
            if (term || trunc) { n_ep = 1; n_trunc = trunc ? 1 : 0; sum_len = ep; }
        } else {                                                        // closing launch: phase C only
            if (done) {
#pragma unroll
                for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
                c.ep_step[i] = 0;
            }
            c.action[i] = a;
        }
    } else {
#pragma unroll
        for (int f = 0; f < F; ++f) fs.phi[f] = 0.0f;
    }
    if (do_a) {
        // block-level sum of the learners' terms, fixed order (reproducible): per wave the MFMA rank-1 chains of wave_rank1_sum,
        // then the 8 per-wave results added in wave order
        constexpr int NWV = BLOCK / 64, H = NWV;
        static_assert(F <= 64 && A <= 4, "dense shared-W reduction: features across the 64 lanes, actions across the 4 rows of a block");
        __shared__ __attribute__((aligned(16))) float tile[NWV][F][kRank1Stride];
        __shared__ float part[H][AF];
        wave_rank1_sum<A, F>(tile[wave], lane, scale, fs.phi, i < N, a, part[wave]);
        __syncthreads();
        if (threadIdx.x < AF) {
            float tot = part[0][threadIdx.x];
#pragma unroll
            for (int h = 1; h < H; ++h) tot += part[h][threadIdx.x];
            atomicAdd(reinterpret_cast<unsigned long long*>(dt.out + (blockIdx.x % (unsigned)kTabRep) * AF + threadIdx.x), dt.quantise(tot));
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// shared weights, stand-alone phase C (closes the last batch-step of a train call): policy.sample with the updated
// weights; finished episodes restart from Domain::default() with a fresh sample.
template <class M>
__global__ __launch_bounds__(kBlock) void k_shared_c(Common c, BasisGeom g, uint64_t t, const uint8_t* __restrict__ flags) {
    constexpr int D = M::D, A = M::A;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t gid = (uint32_t)(c.env_offset + i);
    const bool done = (flags[i] & 3) != 0;
    float s[D];
    if (done) {
        M::Dom::reset(s);
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
        c.ep_step[i] = 0;
    } else {
        load_state<M>(c.state, N, i, s);
    }
    typename M::Feat ft; float q[A];
    M::features(s, g, ft);
    M::q_all(c, 0, g, ft, q);
    const U4 x = draw(c.seed, gid, t, BLK_STEP);
    c.action[i] = policy_sample<A>(c.pol, q, x);
}

}  // namespace rsrl
