// kernels_wave.hpp -- "wave family": ONE WAVEFRONT PER LEARNER for large Fourier bases
// (order 7 on a 4-D state space: F = 8^4 = 4096 features, 3 x 4096 weights per learner).
//
// A learner's weight matrix does not fit one lane's registers, but it fits one WAVE's:
// 64 lanes x 64 features x A actions.  The fused driver loop therefore still keeps W on-chip
// for a whole launch; HBM sees W once per launch (coalesced 16 B/lane loads), and the
// per-action dot products become per-lane partial sums + one DPP wave reduction whose
// result is wave-uniform (every branch on the action, the terminal flag or the episode
// counter is a scalar branch: no divergence anywhere).
//
// Internal feature order: k = c0*512 + c1*64 + c2*8 + c3 (the coefficient vector's digits,
// dimension 0 most significant) for ALL k in [0, F); k = 0 (all-zero coefficient, cos 0 = 1)
// plays the role of the constant feature that lfa's with_bias() stacks last.  Reference
// feature f  <->  k = (f + 1) mod F.   Lane l owns, for every chunk j = 0..7, the 8 consecutive
// k = j*512 + l*8 + v (v = 0..7): c0 = j, c1 = l>>3, c2 = l&7, c3 = v.
//
// HBM layout:  W  WT[N][A][F] (k order, learner-major: a wave streams its learner's rows
//              with 16 B per lane), WT = f32 or bf16.  bf16: arithmetic stays f32, every
//              UPDATED weight is rounded to bf16 with stochastic rounding (so fused and
//              single-step launches round identically); registers hold the rounded values.
#pragma once

#include "models.hpp"

namespace rsrl {

constexpr int kWaveOrder = 7;
constexpr int kWaveN1 = 8;

struct bf16_t { uint16_t bits; };

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __builtin_bit_cast(float, h << 16); }
// 16 rounding bits for element e (0..63) of a lane from the lane's 128-bit Philox block: the 16-bit window at bit (e >> 2) of
// word (e & 3) -- one v_bfe_u32 per weight.  Windows of neighbouring elements overlap (their roundings are correlated), each
// element's own bits are uniform, which is all unbiasedness needs; the previous multiplicative hash cost a quarter-rate
// v_mul_lo_u32 per weight (~8 % of the wave-step).
__device__ __forceinline__ uint32_t sr_bits(const U4& rnd, int e) {
    const uint32_t word = (e & 3) == 0 ? rnd.x : (e & 3) == 1 ? rnd.y : (e & 3) == 2 ? rnd.z : rnd.w;
    return (word >> (e >> 2)) & 0xffffu;
}
// stochastic rounding f32 -> bf16-representable f32: add 16 random bits below the kept mantissa, truncate
__device__ __forceinline__ float round_bf16_sr(float x, uint32_t rnd16) {
    uint32_t b = __builtin_bit_cast(uint32_t, x);
    b += (rnd16 & 0xffffu);
    return __builtin_bit_cast(float, b & 0xffff0000u);
}

template <class WT> struct WaveIO;
template <> struct WaveIO<float> {
    static constexpr bool kBf16 = false;
    // 8 consecutive weights of lane l at element offset `off` (multiple of 8)
    __device__ static __forceinline__ void load8(const float* __restrict__ base, int64_t off, float (&w)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(base + off);
        const float4 b = *reinterpret_cast<const float4*>(base + off + 4);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    }
    __device__ static __forceinline__ void store8(float* __restrict__ base, int64_t off, const float (&w)[8]) {
        *reinterpret_cast<float4*>(base + off) = make_float4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<float4*>(base + off + 4) = make_float4(w[4], w[5], w[6], w[7]);
    }
};
template <> struct WaveIO<bf16_t> {
    static constexpr bool kBf16 = true;
    __device__ static __forceinline__ void load8(const bf16_t* __restrict__ base, int64_t off, float (&w)[8]) {
        const uint4 p = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off);
        w[0] = bf16_to_f32(p.x & 0xffffu); w[1] = __builtin_bit_cast(float, p.x & 0xffff0000u);
        w[2] = bf16_to_f32(p.y & 0xffffu); w[3] = __builtin_bit_cast(float, p.y & 0xffff0000u);
        w[4] = bf16_to_f32(p.z & 0xffffu); w[5] = __builtin_bit_cast(float, p.z & 0xffff0000u);
        w[6] = bf16_to_f32(p.w & 0xffffu); w[7] = __builtin_bit_cast(float, p.w & 0xffff0000u);
    }
    // values are already bf16-representable: plain truncation packs them exactly
    __device__ static __forceinline__ void store8(bf16_t* __restrict__ base, int64_t off, const float (&w)[8]) {
        auto pk = [](float lo, float hi) {
            return (__builtin_bit_cast(uint32_t, lo) >> 16) | (__builtin_bit_cast(uint32_t, hi) & 0xffff0000u);
        };
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + off) =
            make_uint4(pk(w[0], w[1]), pk(w[2], w[3]), pk(w[4], w[5]), pk(w[6], w[7]));
    }
};

// wave total, broadcast as a wave-uniform value
__device__ __forceinline__ float wave_sum_uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_dpp_to_lane63(v)), 63));
}

template <int DOMAIN>
struct WaveFourier {
    using Dom = Domain<DOMAIN>;
    static constexpr int D = Dom::D, A = Dom::A, N1 = kWaveN1, F = 4096;
    static_assert(D == 4, "the wave family is laid out for 4-D state spaces");

    // ---- the per-dimension harmonic tables, SPREAD OVER THE LANES (round 5).  The tables are the register family's -- per dimension one
    // sincospi01 of the scaled state and the angle-addition chain cos/sin(n x) from (n-1) x, FourierTables::build -- but here the state is the
    // wave's, so 64 lanes building all 4 x 8 entries each would do the same work 64 times (it was ~230 issue slots of the wave-step, the packed
    // pair tables included).  Instead lane l builds ONE entry: dimension d = (l >> 3) & 3, harmonic n = l & 7, by running its own dimension's
    // chain and keeping step n -- the very operations of FourierTables::build on that entry, hence the same bits -- and the entries travel:
    // dimension 0 / 3 (indexed by the chunk j / the element v: wave-uniform) by v_readlane, dimension 1 / 2 (indexed by the lane's own c1 / c2)
    // by one ds_bpermute each.
    struct Harm { float c, s; };
    __device__ static __forceinline__ Harm harmonic_of_lane(const float (&sv)[D], int lane) {
        const int d = (lane >> 3) & 3, n = lane & 7;
        float sc[D];
        static_for<0, D>([&](auto Dd) {
            constexpr int dd = Dd;
            constexpr float lo = (float)Dom::lo_d(dd), hi = (float)Dom::hi_d(dd);
            constexpr float inv = 1.0f / (hi - lo);                          // as FourierTables::build
            sc[dd] = (sv[dd] - lo) * inv;
        });
        const float x = d == 0 ? sc[0] : (d == 1 ? sc[1] : (d == 2 ? sc[2] : sc[3]));
        float s1, c1;
        sincospi01(x, s1, c1);
        float pc = c1, ps = s1;                                              // harmonic 1
        Harm h{n == 0 ? 1.0f : c1, n == 0 ? 0.0f : s1};
#pragma unroll
        for (int m = 2; m < N1; ++m) {
            const float nc = fmaf(-ps, s1, pc * c1), nsn = fmaf(pc, s1, ps * c1);
            pc = nc; ps = nsn;
            h.c = (n == m) ? pc : h.c; h.s = (n == m) ? ps : h.s;
        }
        return h;
    }
    __device__ static __forceinline__ float from_lane(float v, int src) { return __shfl(v, src, 64); }
    __device__ static __forceinline__ float uni_lane(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }
    // phi[j][v] = cos(pi * (j*s~0 + c1*s~1 + c2*s~2 + v*s~3)), evaluated exactly like the register family
    // (per dimension one sincospi + the angle-addition chain, then the complex product over dimensions in
    // dimension order) -- the same op order as the f32 oracle.
    __device__ static __forceinline__ void project(const float (&s)[D], int lane, float (&phi)[8][8]) {
        const Harm h = harmonic_of_lane(s, lane);
        const int c1 = lane >> 3, c2 = lane & 7;
        const float e1r = from_lane(h.c, 8 + c1), e1i = from_lane(h.s, 8 + c1), e2r = from_lane(h.c, 16 + c2), e2i = from_lane(h.s, 16 + c2);
        float c3[8], s3[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) { c3[v] = uni_lane(h.c, 24 + v); s3[v] = uni_lane(h.s, 24 + v); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // ((E0[j] * E1[c1]) * E2[c2]) * E3[v], real part
            float re = uni_lane(h.c, j), im = uni_lane(h.s, j);
            float nre = fmaf(-im, e1i, re * e1r), nim = fmaf(re, e1i, im * e1r);
            re = nre; im = nim;
            nre = fmaf(-im, e2i, re * e2r); nim = fmaf(re, e2i, im * e2r);
            re = nre; im = nim;
#pragma unroll
            for (int v = 0; v < 8; ++v) phi[j][v] = fmaf(-im, s3[v], re * c3[v]);
        }
    }
    // per-lane partial of <phi, w_b> (4 interleaved chains), then the wave total (uniform)
    __device__ static __forceinline__ float dot(const float (&phi)[8][8], const float (&w)[8][8]) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int v = 0; v < 8; ++v) acc[v & 3] = fmaf(phi[j][v], w[j][v], acc[v & 3]);
        return wave_sum_uniform((acc[0] + acc[1]) + (acc[2] + acc[3]));
    }
    template <class WT>
    __device__ static __forceinline__ void load_w(const WT* __restrict__ Wi, int lane, float (&w)[A][8][8]) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) WaveIO<WT>::load8(Wi, (int64_t)b * F + j * 512 + lane * 8, w[b][j]);
    }
    template <class WT>
    __device__ static __forceinline__ void store_w(WT* __restrict__ Wi, int lane, const float (&w)[A][8][8]) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) WaveIO<WT>::store8(Wi, (int64_t)b * F + j * 512 + lane * 8, w[b][j]);
    }
    // ---- the fused loop's forms: packed pairs, features streamed through LDS ------------------------------------------
    // The fused loop keeps ONLY the weights in registers (A x 32 register pairs).  Features are produced one 8-wide chunk at
    // a time, folded into the A running dot products at once and parked in the wave's own LDS buffer [chunk][lane][8]
    // (two buffers: phi(s) of the current step, phi(s') of the next); the column update and Q(s',a) read them back.  With both
    // phi buffers in registers next to W (320 values) the compiler had to shuttle ~130 of them through AGPRs around every use.
    // Element (j, v) goes through exactly the operations of project() / dot() / update_col(), pairs (v, v+1) per packed op.
    using PairTab = typename FourierReg<DOMAIN, kWaveOrder>::PairTables;
    struct Stream {
        float c0[8], s0[8];              // dimension-0 harmonics (chunk index j)
        float e1r, e1i, e2r, e2i;        // this lane's dimension-1 / dimension-2 harmonics
        f2 c3[4], s3[4];                 // dimension-3 harmonics, pairs (v, v+1)
    };
    __device__ static __forceinline__ void stream_begin(const float (&s)[D], int lane, Stream& st) {
        const Harm h = harmonic_of_lane(s, lane);
        const int c1 = lane >> 3, c2 = lane & 7;
        st.e1r = from_lane(h.c, 8 + c1); st.e1i = from_lane(h.s, 8 + c1); st.e2r = from_lane(h.c, 16 + c2); st.e2i = from_lane(h.s, 16 + c2);
        // dimension 0 / 3 are indexed by the chunk / the element, the same in every lane: held as wave-uniform scalars they cost no vector
        // registers next to W (32 values that otherwise push the loop past its register budget)
#pragma unroll
        for (int j = 0; j < 8; ++j) { st.c0[j] = uni_lane(h.c, j); st.s0[j] = uni_lane(h.s, j); }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            st.c3[p] = f2{uni_lane(h.c, 24 + 2 * p), uni_lane(h.c, 24 + 2 * p + 1)};
            st.s3[p] = f2{uni_lane(h.s, 24 + 2 * p), uni_lane(h.s, 24 + 2 * p + 1)};
        }
    }
    __device__ static __forceinline__ void stream_chunk(const Stream& st, int j, f2 (&phi)[4]) {
        float re = st.c0[j], im = st.s0[j];
        float nre = fmaf(-im, st.e1i, re * st.e1r), nim = fmaf(re, st.e1i, im * st.e1r);
        re = nre; im = nim;
        nre = fmaf(-im, st.e2i, re * st.e2r); nim = fmaf(re, st.e2i, im * st.e2r);
        re = nre; im = nim;
#pragma unroll
        for (int p = 0; p < 4; ++p) phi[p] = __builtin_elementwise_fma(splat2(-im), st.s3[p], splat2(re) * st.c3[p]);
    }
    __device__ static __forceinline__ void lds_put(float* __restrict__ P, int j, const f2 (&phi)[4]) {
        float4* q = reinterpret_cast<float4*>(P + j * 512);
        q[0] = make_float4(phi[0].x, phi[0].y, phi[1].x, phi[1].y);
        q[1] = make_float4(phi[2].x, phi[2].y, phi[3].x, phi[3].y);
    }
    __device__ static __forceinline__ void lds_get(const float* __restrict__ P, int j, f2 (&phi)[4]) {
        const float4* q = reinterpret_cast<const float4*>(P + j * 512);
        const float4 a = q[0], b = q[1];
        phi[0] = f2{a.x, a.y}; phi[1] = f2{a.z, a.w}; phi[2] = f2{b.x, b.y}; phi[3] = f2{b.z, b.w};
    }
    // phi(s) -> LDS buffer P (this lane's slots), Q(s, b) for every action (wave-uniform)
    __device__ static __forceinline__ void stream_project_q(const float (&s)[D], int lane, float* __restrict__ P, const f2 (&w)[A][8][4],
                                                            float (&q)[A]) {
        Stream st;
        stream_begin(s, lane, st);
        f2 acc[A][2];
#pragma unroll
        for (int b = 0; b < A; ++b) { acc[b][0] = splat2(0.0f); acc[b][1] = splat2(0.0f); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f2 phi[4];
            stream_chunk(st, j, phi);
            lds_put(P, j, phi);
#pragma unroll
            for (int b = 0; b < A; ++b)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[b][p & 1] = __builtin_elementwise_fma(phi[p], w[b][j][p], acc[b][p & 1]);
        }
#pragma unroll
        for (int b = 0; b < A; ++b) q[b] = wave_sum_uniform((acc[b][0].x + acc[b][0].y) + (acc[b][1].x + acc[b][1].y));
    }
    // W[:,a] += scale * phi(s) from the LDS buffer Ps, then Q(s',a) with the updated column against phi(s') in Pn
    template <class WT>
    __device__ static __forceinline__ float stream_update_q(f2 (&wa)[8][4], const float* __restrict__ Ps, const float* __restrict__ Pn, float scale,
                                                            const U4& rnd) {
        f2 acc[2] = {splat2(0.0f), splat2(0.0f)};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f2 ps[4], pn[4];
            lds_get(Ps, j, ps);
            lds_get(Pn, j, pn);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f2 x = __builtin_elementwise_fma(splat2(scale), ps[p], wa[j][p]);
                if constexpr (WaveIO<WT>::kBf16) {
                    x.x = round_bf16_sr(x.x, sr_bits(rnd, j * 8 + 2 * p));
                    x.y = round_bf16_sr(x.y, sr_bits(rnd, j * 8 + 2 * p + 1));
                }
                wa[j][p] = x;
                acc[p & 1] = __builtin_elementwise_fma(pn[p], x, acc[p & 1]);
            }
        }
        return wave_sum_uniform((acc[0].x + acc[0].y) + (acc[1].x + acc[1].y));
    }
    template <class WT>
    __device__ static __forceinline__ void load_w2(const WT* __restrict__ Wi, int lane, f2 (&w)[A][8][4]) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float w8[8];
                WaveIO<WT>::load8(Wi, (int64_t)b * F + j * 512 + lane * 8, w8);
#pragma unroll
                for (int p = 0; p < 4; ++p) w[b][j][p] = f2{w8[2 * p], w8[2 * p + 1]};
            }
    }
    template <class WT>
    __device__ static __forceinline__ void store_w2(WT* __restrict__ Wi, int lane, const f2 (&w)[A][8][4]) {
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float w8[8];
#pragma unroll
                for (int p = 0; p < 4; ++p) { w8[2 * p] = w[b][j][p].x; w8[2 * p + 1] = w[b][j][p].y; }
                WaveIO<WT>::store8(Wi, (int64_t)b * F + j * 512 + lane * 8, w8);
            }
    }
    // Q(s,.) with W streamed from memory (granular ops)
    template <class WT>
    __device__ static __forceinline__ void q_from_mem(const WT* __restrict__ Wi, int lane, const float (&phi)[8][8], float (&q)[A]) {
#pragma unroll
        for (int b = 0; b < A; ++b) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float w8[8];
                WaveIO<WT>::load8(Wi, (int64_t)b * F + j * 512 + lane * 8, w8);
#pragma unroll
                for (int v = 0; v < 8; ++v) acc[v & 3] = fmaf(phi[j][v], w8[v], acc[v & 3]);
            }
            q[b] = wave_sum_uniform((acc[0] + acc[1]) + (acc[2] + acc[3]));
        }
    }
    // W[:,a] += scale*phi (one column, a is wave-uniform); bf16: stochastic rounding of every updated weight
    template <class WT>
    __device__ static __forceinline__ void update_col(float (&wa)[8][8], const float (&phi)[8][8], float scale, const U4& rnd) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                float x = fmaf(scale, phi[j][v], wa[j][v]);
                if constexpr (WaveIO<WT>::kBf16) {
                    x = round_bf16_sr(x, sr_bits(rnd, j * 8 + v));
                }
                wa[j][v] = x;
            }
    }
};

enum : uint32_t { BLK_SR_BASE = 16 };   // stochastic-rounding draws: block = 16 + lane

// ---------------------------------------------------------------------------------------
// fused driver loop, one wave per learner (semantics identical to k_train_reg)
// ---------------------------------------------------------------------------------------
template <int DOMAIN, class WT, bool ESCHED = false>
__global__ __launch_bounds__(kBlock) void k_train_wave(Common c, WT* __restrict__ Wbase, uint64_t t0, int n_steps,
                                                       DevStats* __restrict__ stats) {
    // ESCHED: the per-learner epsilon schedule (Common::eps; examples/sarsa_lambda.rs:68) -- an instantiation of its own, as for k_train_reg: the
    // learner is the wave's, so its epsilon stays wave-uniform (scalar registers) and the schedule-free loop is not one instruction longer
    using WF = WaveFourier<DOMAIN>;
    using Dom = Domain<DOMAIN>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);      // learner of this wave (uniform)
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        WT* Wi = Wbase + i * (int64_t)(A * F);
        float s[D];
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
        int a = __builtin_amdgcn_readfirstlane(c.action[i]);
        uint32_t ep = c.ep_step[i];
        f2 w[A][8][4];
        WF::template load_w2<WT>(Wi, lane, w);
        // this wave's two feature buffers in LDS, [chunk][lane][8]: only this wave touches them (no barriers)
        __shared__ __attribute__((aligned(16))) float sh_phi[kBlock / 64][2][8 * 64 * 8];
        float* const P0 = &sh_phi[threadIdx.x >> 6][0][lane * 8];
        float* const P1 = &sh_phi[threadIdx.x >> 6][1][lane * 8];
        float q_s[A];
        WF::stream_project_q(s, lane, P0, w, q_s);
        float facc_abs = 0.0f, facc_r = 0.0f;
        PolicyParams pol = c.pol;
        if constexpr (ESCHED) learner_eps_load(c, i, pol);

        auto one_step = [&](const float* __restrict__ Ps, float* __restrict__ Pn, uint64_t t) {
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            bool term;
            if constexpr (DOMAIN == 2) term = Dom::step_uniform(ns, a, r, lane);      // wave-uniform state: trigonometry across lanes
            else term = Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            if (term) Dom::reset(ns);
            float q_n[A];
            WF::stream_project_q(ns, lane, Pn, w, q_n);
            U4 xin = U4{0, 0, 0, 0};
            if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
            float e;
            const float delta = td_dispatch<A>(c.alg, (ESCHED && c.apol_same) ? pol : c.apol, q_s, a, q_n, r, term, xin, e, lane);
            const float scale = c.alg.lr * e;
            U4 rnd = U4{0, 0, 0, 0};
            if constexpr (WaveIO<WT>::kBf16) rnd = draw(c.seed, gid, t, BLK_SR_BASE + (uint32_t)lane);
            // a is wave-uniform: a scalar branch picks the column, 64 fma per lane instead of A*64
            float qa = 0.0f;
            static_for<0, A>([&](auto Bb) {
                constexpr int b = Bb;
                if (a == b) qa = WF::template stream_update_q<WT>(w[b], Ps, Pn, scale, rnd);     // Q(s',a) with the UPDATED column
            });
#pragma unroll
            for (int b = 0; b < A; ++b) q_n[b] = (a == b) ? qa : q_n[b];
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            if constexpr (ESCHED) learner_eps_step(c, term | trunc, pol);       // the episode's last handle is done: its end decays epsilon
            int na = policy_sample<A>(pol, q_n, x, lane);
            facc_abs += fabsf(delta); facc_r += r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                Dom::reset(ns);
                WF::stream_project_q(ns, lane, Pn, w, q_n);
                const U4 xr = draw(c.seed, gid, t, BLK_RESET);
                na = policy_sample<A>(pol, q_n, xr, lane);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
#pragma unroll
            for (int b = 0; b < A; ++b) q_s[b] = q_n[b];
            a = __builtin_amdgcn_readfirstlane(na);
        };
        int k = 0;
        for (; k + 1 < n_steps; k += 2) {
            one_step(P0, P1, t0 + (uint64_t)k);
            one_step(P1, P0, t0 + (uint64_t)k + 1);
        }
        if (k < n_steps) one_step(P0, P1, t0 + (uint64_t)k);

        WF::template store_w2<WT>(Wi, lane, w);
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
            c.action[i] = a;
            c.ep_step[i] = ep;
            if constexpr (ESCHED) c.eps[i] = pol.eps;
            sum_abs = (double)facc_abs; sum_r = (double)facc_r;
        } else {
            n_ep = 0; n_trunc = 0; sum_len = 0;                      // the wave's statistics are counted once (lane 0)
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// ---------------------------------------------------------------------------------------
// bf16 weights, TWO waves per SIMD (round 5): the weights stay PACKED in the registers
// ---------------------------------------------------------------------------------------
// k_train_wave keeps W as fp32 values: 192 VGPRs per lane, 411-432 registers with everything else, ONE wave per SIMD -- and > 80 % of its
// ~1 720 issue slots per env-step are the learner's scalar work (RK4, two softmaxes, Philox, rounding) done 64-wide for one learner with nothing
// to hide behind.  bf16 weights are bf16-representable by construction (every update is rounded), so the registers can hold them as they lie in
// memory: two per dword, A x 32 = 96 VGPRs.  A pair is opened with two integer instructions where it is used (v_lshlrev_b32 16 / v_and_b32
// 0xffff0000: exact) and closed with one v_perm_b32 after its update; the arithmetic between is the fp32 arithmetic of k_train_wave in the
// same order -- every bit of W, Q, delta and the trajectory is unchanged (tests/test_gpu_bitwise.py, test_gpu_fullsize.py: still bitwise
// against the oracle).  What that buys: <= 256 registers, so a SECOND wave is resident on every SIMD to issue into the first one's dependency
// stalls -- and, so that eight waves fit a CU's 160 KiB of LDS, only phi(s) lives there (16 KiB per wave instead of 32): the update pass rebuilds each
// chunk of phi(s') from 16 kept values (WaveFourierPk::Tail) and writes it over the chunk of phi(s) it has just consumed.
struct WavePk {
    // shift count and mask of open(), each pass's own OPAQUE scalars (an empty asm statement defines them): written as literals, the compiler
    // recognises the pair opened in the projection pass as the one the update pass opens again and KEEPS the 192 opened halves alive between
    // the two -- in scratch memory -- instead of spending the two integer instructions again
    uint32_t sh, mask;
    // (the mask lives in a VECTOR register: on gfx950 a VALU instruction with a scalar-register operand issues at the slow rate -- v_and_b32 v, s, v
    // 4.9 cycles against 2.5 with two waves per SIMD, profiles/r05_ubench_valu_pair.txt; the shift is a slow instruction either way)
    __device__ __forceinline__ WavePk() : sh(16u), mask(0xffff0000u) { asm volatile("" : "+s"(sh), "+v"(mask)); }
    __device__ __forceinline__ f2 open(uint32_t w) const { return f2{__builtin_bit_cast(float, w << sh), __builtin_bit_cast(float, w & mask)}; }
    // both halves already bf16-representable (low 16 bits zero): bytes 3:2 of each
    // (scalar arguments on purpose: with an f2 argument, `bit_cast<uint32_t>(x.y)` of a vector assembled element by element came out of ROCm 7.2's
    // clang as the LOW element -- v_perm_b32 v, lo, lo -- the same mis-extraction scripts/ubench/pk_forward.hip ran into with an f2 asm output)
    __device__ static __forceinline__ uint32_t close(float lo, float hi) {
        return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x07060302u);
    }
};

template <int DOMAIN>
struct WaveFourierPk : WaveFourier<DOMAIN> {
    using Base = WaveFourier<DOMAIN>;
    using typename Base::Stream;
    static constexpr int D = Base::D, A = Base::A, F = Base::F;
    // What of phi(s') survives from the projection to the update: per chunk j the complex product E0[j] * E1[c1] * E2[c2] of this lane (16 values) and
    // the wave-uniform dimension-3 harmonics (scalar registers).  The chunk's 8 features are ONE more packed multiply + fma each from there
    // (the tail of stream_chunk): the update pass recomputes them -- 8 packed instructions per chunk, bit for bit what the projection pass fed into
    // the dot products -- instead of holding 64 feature values in registers (which spilled) or in a second LDS buffer (which does not fit eight waves).
    struct Tail { float re[8], im[8]; f2 c3[4], s3[4]; };
    __device__ static __forceinline__ void chunk_head(const Stream& st, int j, float& re_out, float& im_out) {
        float re = st.c0[j], im = st.s0[j];
        float nre = fmaf(-im, st.e1i, re * st.e1r), nim = fmaf(re, st.e1i, im * st.e1r);
        re = nre; im = nim;
        nre = fmaf(-im, st.e2i, re * st.e2r); nim = fmaf(re, st.e2i, im * st.e2r);
        re_out = nre; im_out = nim;
    }
    __device__ static __forceinline__ void chunk_tail(const Tail& tl, int j, f2 (&phi)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) phi[p] = __builtin_elementwise_fma(splat2(-tl.im[j]), tl.s3[p], splat2(tl.re[j]) * tl.c3[p]);
    }
    // Q(s, b) for every action (wave-uniform) -- the sums of stream_project_q, term for term -- and the Tail of phi(s)
    __device__ static __forceinline__ void project_q(const float (&s)[D], int lane, const uint32_t (&wp)[A][8][4], float (&q)[A], Tail& tl) {
        Stream st;
        Base::stream_begin(s, lane, st);
#pragma unroll
        for (int p = 0; p < 4; ++p) { tl.c3[p] = st.c3[p]; tl.s3[p] = st.s3[p]; }
        f2 acc[A][2];
        const WavePk pk;
#pragma unroll
        for (int b = 0; b < A; ++b) { acc[b][0] = splat2(0.0f); acc[b][1] = splat2(0.0f); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            chunk_head(st, j, tl.re[j], tl.im[j]);
            f2 phi[4];
            chunk_tail(tl, j, phi);
#pragma unroll
            for (int b = 0; b < A; ++b)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[b][p & 1] = __builtin_elementwise_fma(phi[p], pk.open(wp[b][j][p]), acc[b][p & 1]);
            // a chunk is self-contained (24 opened halves, 12 packed fmas): the scheduler may not pull the next chunks' unpacking up in front of
            // it -- hoisted, the opened pairs of several chunks are live at once
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < A; ++b) q[b] = wave_sum_uniform((acc[b][0].x + acc[b][0].y) + (acc[b][1].x + acc[b][1].y));
    }
    // W[:,a] += scale * phi(s) (phi(s) read from the wave's LDS buffer P, chunk by chunk, each chunk then OVERWRITTEN with phi(s') -- next step's
    // phi(s)), every updated weight rounded stochastically; returns Q(s',a) with the updated column (stream_update_q<bf16_t>, term for term)
    __device__ static __forceinline__ float update_q(uint32_t (&wa)[8][4], float* __restrict__ P, const Tail& tn, float scale, const U4& rnd) {
        f2 acc[2] = {splat2(0.0f), splat2(0.0f)};
        const WavePk pk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f2 ps[4], pn[4];
            Base::lds_get(P, j, ps);
            chunk_tail(tn, j, pn);
            Base::lds_put(P, j, pn);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f2 u = __builtin_elementwise_fma(splat2(scale), ps[p], pk.open(wa[j][p]));
                const float x0 = round_bf16_sr(u.x, sr_bits(rnd, j * 8 + 2 * p));
                const float x1 = round_bf16_sr(u.y, sr_bits(rnd, j * 8 + 2 * p + 1));
                wa[j][p] = WavePk::close(x0, x1);
                acc[p & 1] = __builtin_elementwise_fma(pn[p], f2{x0, x1}, acc[p & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return wave_sum_uniform((acc[0].x + acc[0].y) + (acc[1].x + acc[1].y));
    }
    __device__ static __forceinline__ void put_all(float* __restrict__ P, const Tail& tl) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f2 phi[4];
            chunk_tail(tl, j, phi);
            Base::lds_put(P, j, phi);
        }
    }
};

template <int DOMAIN, bool ESCHED = false>
__global__ __launch_bounds__(kBlock, 2) void k_train_wave_pk(Common c, bf16_t* __restrict__ Wbase, uint64_t t0, int n_steps, DevStats* __restrict__ stats) {
    using WF = WaveFourierPk<DOMAIN>;
    using Dom = Domain<DOMAIN>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t N = c.n_envs;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);      // learner of this wave (uniform)
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < N) {
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        bf16_t* Wi = Wbase + i * (int64_t)(A * F);
        float s[D];
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
        int a = __builtin_amdgcn_readfirstlane(c.action[i]);
        uint32_t ep = c.ep_step[i];
        uint32_t wp[A][8][4];                                                        // the lane's 8 x 8 weights of every action, two per register
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(Wi) + (int64_t)b * F + j * 512 + lane * 8);
                wp[b][j][0] = v.x; wp[b][j][1] = v.y; wp[b][j][2] = v.z; wp[b][j][3] = v.w;
            }
        // this wave's feature buffer in LDS, [chunk][lane][8]: phi of the CURRENT state; only this wave touches it (no barriers)
        __shared__ __attribute__((aligned(16))) float sh_phi[kBlock / 64][8 * 64 * 8];
        float* const P = &sh_phi[threadIdx.x >> 6][lane * 8];
        float q_s[A];
        {
            typename WF::Tail t0;
            WF::project_q(s, lane, wp, q_s, t0);
            WF::put_all(P, t0);
        }
        float facc_abs = 0.0f, facc_r = 0.0f;
        PolicyParams pol = c.pol;
        if constexpr (ESCHED) learner_eps_load(c, i, pol);          // (k_train_wave: the learner's epsilon is wave-uniform)
        for (int k = 0; k < n_steps; ++k) {
            const uint64_t t = t0 + (uint64_t)k;
            // the step's Philox blocks depend on nothing but (learner, t): drawn FIRST, their integer work sits in the same basic block as the
            // transition's long dependent chain and fills its stalls
            const U4 rnd = draw(c.seed, gid, t, BLK_SR_BASE + (uint32_t)lane);
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            float ns[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ns[d] = s[d];
            float r;
            bool term;
            if constexpr (DOMAIN == 2) term = Dom::step_uniform(ns, a, r, lane);      // wave-uniform state: trigonometry across lanes
            else term = Dom::step(ns, a, r);
            ep += 1;
            const bool trunc = !term && cap > 0 && ep >= cap;
            if (term) Dom::reset(ns);
            float q_n[A];
            typename WF::Tail tn;
            WF::project_q(ns, lane, wp, q_n, tn);
            U4 xin = U4{0, 0, 0, 0};
            if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, gid, t, BLK_INNER);
            float e;
            const float delta = td_dispatch<A>(c.alg, (ESCHED && c.apol_same) ? pol : c.apol, q_s, a, q_n, r, term, xin, e, lane);
            const float scale = c.alg.lr * e;
            float qa = 0.0f;
            static_for<0, A>([&](auto Bb) {
                constexpr int b = Bb;
                if (a == b) qa = WF::update_q(wp[b], P, tn, scale, rnd);                // Q(s',a) with the UPDATED column; LDS now holds phi(s')
            });
#pragma unroll
            for (int b = 0; b < A; ++b) q_n[b] = (a == b) ? qa : q_n[b];
            if constexpr (ESCHED) learner_eps_step(c, term | trunc, pol);
            int na = policy_sample<A>(pol, q_n, x, lane);
            facc_abs += fabsf(delta); facc_r += r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                Dom::reset(ns);
                WF::project_q(ns, lane, wp, q_n, tn);
                WF::put_all(P, tn);
                const U4 xr = draw(c.seed, gid, t, BLK_RESET);
                na = policy_sample<A>(pol, q_n, xr, lane);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
#pragma unroll
            for (int b = 0; b < A; ++b) q_s[b] = q_n[b];
            a = __builtin_amdgcn_readfirstlane(na);
        }
#pragma unroll
        for (int b = 0; b < A; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(Wi) + (int64_t)b * F + j * 512 + lane * 8) = make_uint4(wp[b][j][0], wp[b][j][1], wp[b][j][2], wp[b][j][3]);
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
            c.action[i] = a;
            c.ep_step[i] = ep;
            if constexpr (ESCHED) c.eps[i] = pol.eps;
            sum_abs = (double)facc_abs; sum_r = (double)facc_r;
        } else {
            n_ep = 0; n_trunc = 0; sum_len = 0;                      // the wave's statistics are counted once (lane 0)
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// ---------------------------------------------------------------------------------------
// trait-granular kernels, one wave per item
// ---------------------------------------------------------------------------------------
template <int DOMAIN, class WT>
__global__ __launch_bounds__(kBlock) void k_wave_reset(Common c, const WT* __restrict__ Wbase, uint64_t t) {
    using WF = WaveFourier<DOMAIN>; using Dom = Domain<DOMAIN>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (i >= c.n_envs) return;
    float s[D]; Dom::reset(s);
    float phi[8][8], q[A];
    WF::project(s, lane, phi);
    WF::template q_from_mem<WT>(Wbase + i * (int64_t)(A * F), lane, phi, q);
    const U4 x = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_INIT);
    PolicyParams pol = c.pol;
    learner_eps_load(c, i, pol);                     // per-learner epsilon, when configured (a reset is not an episode end: no decay)
    const int a = policy_sample<A>(pol, q, x);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) c.state[(int64_t)d * c.n_envs + i] = s[d];
        c.action[i] = a;
        c.ep_step[i] = 0;
    }
}

template <int DOMAIN, class WT>
__global__ __launch_bounds__(kBlock) void k_wave_qop(Common c, const WT* __restrict__ Wbase, int op, const float* __restrict__ states,
                                                     int64_t Mn, uint64_t call, float* __restrict__ fout, int32_t* __restrict__ iout,
                                                     const float* __restrict__ fin, const int32_t* __restrict__ iin) {
    using WF = WaveFourier<DOMAIN>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (i >= Mn) return;
    float s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = states[(int64_t)d * Mn + i];
    float phi[8][8];
    WF::project(s, lane, phi);
    if (op == QOP_FEATURES) {               // reference order: feature f = k - 1 for k >= 1, the constant (k = 0) last
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int k = j * 512 + lane * 8 + v;
                fout[(int64_t)((k + F - 1) % F) * Mn + i] = phi[j][v];
            }
        return;
    }
    float q[A];
    WF::template q_from_mem<WT>(Wbase + i * (int64_t)(A * F), lane, phi, q);
    const bool is_sample = op == QOP_SAMPLE || op == QOP_SAMPLE_STEP || op == QOP_SAMPLE_INIT;
    if (lane != 0 && !is_sample) return;
    if (is_sample) {                        // (the softmax path may spread its exponentials over lanes: every lane evaluates)
        const U4 x = draw(c.seed, (uint32_t)(c.env_offset + i), call, op == QOP_SAMPLE ? BLK_API : (op == QOP_SAMPLE_STEP ? BLK_STEP : BLK_INIT));
        PolicyParams pol = c.pol;
        learner_eps_load(c, i, pol);                 // item i is evaluated with learner i's policy object (qop_finish)
        const int a = policy_sample<A>(pol, q, x);
        if (lane == 0) iout[i] = a;
        return;
    }
    qop_finish<A>(c, op, q, Mn, i, call, fout, iout, fin, iin);
}

template <int DOMAIN, class WT>
__global__ __launch_bounds__(kBlock) void k_wave_handle(Common c, WT* __restrict__ Wbase, const float* __restrict__ from,
                                                        const int32_t* __restrict__ act, const float* __restrict__ rew,
                                                        const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                        int64_t Mn, uint64_t t, float* __restrict__ td_out) {
    using WF = WaveFourier<DOMAIN>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (i >= Mn) return;
    WT* Wi = Wbase + i * (int64_t)(A * F);
    float s[D], ns[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { s[d] = from[(int64_t)d * Mn + i]; ns[d] = to[(int64_t)d * Mn + i]; }
    const int a = clamp_action<A>(__builtin_amdgcn_readfirstlane(act[i]));
    const float r = rew[i];
    const bool term = termf[i] != 0;
    float phi_s[8][8], phi_n[8][8], q_s[A], q_n[A];
    WF::project(s, lane, phi_s);
    WF::project(ns, lane, phi_n);
    WF::template q_from_mem<WT>(Wi, lane, phi_s, q_s);
    WF::template q_from_mem<WT>(Wi, lane, phi_n, q_n);
    U4 xin = U4{0, 0, 0, 0};
    if (c.alg.kind == ALG_SARSA) xin = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_INNER);
    float e;
    PolicyParams apol = c.apol;
    if (c.apol_same) learner_eps_load(c, i, apol);              // the agent shares the behaviour policy object: this learner's epsilon (k_handle)
    const float delta = td_dispatch<A>(c.alg, apol, q_s, a, q_n, r, term, xin, e);
    const float scale = c.alg.lr * e;
    U4 rnd = U4{0, 0, 0, 0};
    if constexpr (WaveIO<WT>::kBf16) rnd = draw(c.seed, (uint32_t)(c.env_offset + i), t, BLK_SR_BASE + (uint32_t)lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float w8[1][8];
        const int64_t off = (int64_t)a * F + j * 512 + lane * 8;
        WaveIO<WT>::load8(Wi, off, w8[0]);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            float x = fmaf(scale, phi_s[j][v], w8[0][v]);
            if constexpr (WaveIO<WT>::kBf16) {
                x = round_bf16_sr(x, sr_bits(rnd, j * 8 + v));
            }
            w8[0][v] = x;
        }
        WaveIO<WT>::store8(Wi, off, w8[0]);
    }
    if (td_out && lane == 0) td_out[i] = delta;
}

template <int DOMAIN, class WT>
__global__ __launch_bounds__(kBlock) void k_wave_rollout(Common c, const WT* __restrict__ Wbase, int64_t step_limit,
                                                         uint32_t* __restrict__ n_states, float* __restrict__ total_reward, int64_t Mn, TrajOut tr,
                                                         RolloutPolicy rp) {
    using WF = WaveFourier<DOMAIN>; using Dom = Domain<DOMAIN>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (i >= Mn) return;
    const uint32_t gid = (uint32_t)(c.env_offset + i);
    uint64_t kk = 0;
    float w[A][8][8];
    WF::template load_w<WT>(Wbase + i * (int64_t)(A * F), lane, w);
    float s[D]; Dom::reset(s);
    if (tr.states && lane == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) tr.states[(int64_t)d * tr.Mn + i] = s[d];
    }
    float phi[8][8], q[A], r, tot = 0.0f;
    WF::project(s, lane, phi);
#pragma unroll
    for (int b = 0; b < A; ++b) q[b] = WF::dot(phi, w[b]);
    int a = rollout_action<A>(c.pol, rp, q, c.seed, gid, kk++);
    bool term = Dom::step(s, a, r);
    int64_t steps = 0;
    while (steps < step_limit - 1) {
        if (lane == 0) traj_record<D>(tr, i, steps, s, a, r);
        steps += 1; tot += r;
        if (term) break;
        if (steps >= step_limit - 1) break;
        WF::project(s, lane, phi);
#pragma unroll
        for (int b = 0; b < A; ++b) q[b] = WF::dot(phi, w[b]);
        a = rollout_action<A>(c.pol, rp, q, c.seed, gid, kk++);
        term = Dom::step(s, a, r);
    }
    if (lane == 0) {
        n_states[i] = (uint32_t)(steps + 1);
        if (total_reward) total_reward[i] = tot;
        if (tr.terminal) tr.terminal[i] = (steps > 0 && term) ? 1 : 0;
    }
}

// get / set of one learner's weights in the reference format f32[F][A] (feature f <-> k = (f+1) mod F)
template <class WT>
__global__ void k_wave_weights_get(const WT* __restrict__ Wi, int F, int A, float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F * A) return;
    const int f = j / A, b = j % A, k = (f + 1) % F;
    float tmp[8];
    WaveIO<WT>::load8(Wi, (int64_t)b * F + (k & ~7), tmp);
    float v = tmp[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) v = ((k & 7) == u) ? tmp[u] : v;
    out[j] = v;
}
template <class WT>
__global__ void k_wave_weights_set(WT* __restrict__ W, int64_t first, int64_t count, int F, int A, const float* __restrict__ in) {
    // one thread per group of 8 consecutive k of one (learner, action); bf16: round to nearest even
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int groups = F / 8;
    if (g >= count * A * groups) return;
    const int64_t li = g / (A * groups);
    const int b = (int)((g / groups) % A), k0 = (int)(g % groups) * 8;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = k0 + u, f = (k + F - 1) % F;
        float x = in[(int64_t)f * A + b];
        if constexpr (WaveIO<WT>::kBf16) {
            uint32_t bits = __builtin_bit_cast(uint32_t, x);
            bits += 0x7fffu + ((bits >> 16) & 1u);
            x = __builtin_bit_cast(float, bits & 0xffff0000u);
        }
        v[u] = x;
    }
    WaveIO<WT>::store8(W + (first + li) * (int64_t)(A * F), (int64_t)b * F + k0, v);
}

}  // namespace rsrl
