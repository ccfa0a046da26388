// device_core.hpp -- device-side building blocks of the TD-control hot path (gfx950).
//
// Everything here is the MI355X restatement of the reference arithmetic; each
// block cites the reference source it stands in for (paths under the reference
// repository).  Compiled with -ffp-contract=off: every fused multiply-add is
// written out as fmaf so that the op order is explicit (and identical to the
// f32 instantiation of the test oracle).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <type_traits>
#include <utility>

namespace rsrl {

// ---------------------------------------------------------------------------------------
// compile-time loop: static_for<0,N>([&](auto I){ constexpr int i = I; ... })
// keeps every register-array index a compile-time constant (no scratch).
// ---------------------------------------------------------------------------------------
template <int B, int E, class Fn>
__host__ __device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (B < E) {
        fn(std::integral_constant<int, B>{});
        static_for<B + 1, E>(fn);
    }
}
constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }

// ---------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al. SC'11).  Stands in for rand 0.7's
// StdRng / thread_rng streams (examples/q_learning.rs:22, sarsa.rs:61): one independent
// stream per (seed, GLOBAL env id), addressed by (batch-step, block) -- no state in HBM.
// ---------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

// a ^ b ^ c in one instruction on the device (v_bitop3_b32, truth table 0x96): the compiler emits two v_xor for it,
// 20 extra instructions in the ten rounds below, on the dependent chain of every draw
__host__ __device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

__host__ __device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                     uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
enum : uint32_t { BLK_STEP = 0, BLK_RESET = 1, BLK_INNER = 2, BLK_INIT = 3, BLK_API = 4, BLK_ROLLOUT = 5 };
// key = (seed_lo, seed_hi); counter = (t_lo, t_hi, env_id, block): the whole 128-bit block
__host__ __device__ __forceinline__ U4 draw_block(uint64_t seed, uint32_t env_id, uint64_t t, uint32_t block) {
    return philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), env_id, block, (uint32_t)seed, (uint32_t)(seed >> 32));
}
// The per-step draws -- the behaviour policy's sample (BLK_STEP; the sample after an episode restart is that step's one and only
// behaviour sample and uses the same draw: BLK_RESET is an alias) and the agent's own sample (BLK_INNER, sarsa.rs:61) -- need
// TWO words: x = explore?, y = z = the uniform pick (a random action and a tie-break / softmax u exclude each other).  Two
// consecutive batch-steps therefore share one Philox block, addressed by t >> 1: the even step takes words 0, 1, the odd step
// words 2, 3 -- the fused loop generates one block per two steps.  Every other stream (initial sample BLK_INIT, API calls
// BLK_API, stochastic-rounding blocks) takes a whole block per (t, block).
__host__ __device__ __forceinline__ U4 half_block(const U4& p, bool odd) {
    const uint32_t x = odd ? p.z : p.x, y = odd ? p.w : p.y;
    return U4{x, y, y, 0u};
}
__host__ __device__ __forceinline__ U4 draw(uint64_t seed, uint32_t env_id, uint64_t t, uint32_t block) {
    if (block <= BLK_INNER)
        return half_block(draw_block(seed, env_id, t >> 1, block == BLK_INNER ? BLK_INNER : BLK_STEP), (t & 1u) != 0);
    return draw_block(seed, env_id, t, block);
}
__host__ __device__ __forceinline__ uint32_t mulhi_u32(uint32_t x, uint32_t n) {
    return (uint32_t)(((uint64_t)x * n) >> 32);
}

// ---------------------------------------------------------------------------------------
// Domains.  clip! = lb.max(ub.min(x)), wrap! = repeated +-(ub-lb)   (rsrl_domains/src/macros.rs:3-24)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float clipf(float lb, float x, float ub) { return fmaxf(lb, fminf(ub, x)); }
__device__ __forceinline__ float wrapf(float lb, float x, float ub) {
    const float diff = ub - lb;
    while (x > ub) x -= diff;
    while (x < lb) x += diff;
    return x;
}

constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------
// Branch-free fp32 trigonometry (no large-argument slow path => one basic block, so the
// scheduler can interleave it with the Philox / FMA chains of a lone wave per SIMD).
// Measured against f64: |err| <= 9e-8 (<= 1.7 ulp) on the stated ranges.
// ---------------------------------------------------------------------------------------
// (sin, cos) of r + q*pi/2 from (s, c) = (sin r, cos r), q in 0..3:  (s, c), (c, -s), (-s, -c), (-c, s).
// Written as one swap (two selects) and two sign-bit xors: as a ternary chain the compiler turned it into exec-mask
// branches with half of the sine polynomial sunk into them -- a lone wave per SIMD pays every taken branch in full.
// x < 0 ? all ones : 0 as ONE v_ashrrev_i32, written as the instruction: the optimiser canonicalises `x >> 31` into sext(x < 0)
// and the backend emits v_cmp + v_cndmask for that -- an SGPR round trip with a wait state in front of the select (gfx950), which a
// lone wave per SIMD pays in full.  Selects on small integers go through such masks and v_bfi_b32 instead (same values).
__device__ __forceinline__ uint32_t sign_mask(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m;
    asm("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(m) : "v"(x));
    return m;
#else
    return (uint32_t)(x >> 31);
#endif
}
__device__ __forceinline__ float bitsel(uint32_t m, float x, float y) {          // m ? x : y for an all-ones / all-zeros m
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, x) & m) | (__builtin_bit_cast(uint32_t, y) & ~m));
}
// (sin, cos) of the reduced argument -> (sin, cos) of the angle in quadrant q: swapped when q is odd, sin negated in quadrants 2, 3, cos
// in quadrants 1, 2.  Seven integer instructions, written as the instructions: the mask by v_bfe_i32 (bit 0 sign-extended), each select ONE
// v_bitop3_b32 (bitfield insert, truth table 0xca), each sign flip ONE v_bitop3_b32 (x ^ (t & 0x80000000), 0x78) -- the generic
// (x & m) | (y & ~m) spelling compiled to twelve.  The same bits: integer logic only.
__device__ __forceinline__ void quadrant_select(int q, float s, float c, float& sn, float& cs) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t su = __builtin_bit_cast(uint32_t, s), cu = __builtin_bit_cast(uint32_t, c);
    const uint32_t swap = (uint32_t)__builtin_amdgcn_sbfe(q, 0, 1);     // q odd
    const uint32_t a = __builtin_amdgcn_bitop3_b32(swap, cu, su, 0xca);  // swap ? c : s
    const uint32_t b = __builtin_amdgcn_bitop3_b32(swap, su, cu, 0xca);  // swap ? s : c
    const uint32_t ta = (uint32_t)q << 30;                              // bit 31 = q & 2       : sin negative in quadrants 2, 3
    const uint32_t tb = ((uint32_t)q << 30) + 0x40000000u;              // bit 31 = (q + 1) & 2 : cos negative in quadrants 1, 2
    sn = __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(a, ta, 0x80000000u, 0x78));
    cs = __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(b, tb, 0x80000000u, 0x78));
#else
    const uint32_t swap = sign_mask(q << 31);
    const float a = bitsel(swap, c, s);
    const float b = bitsel(swap, s, c);
    const uint32_t sa = ((uint32_t)q & 2u) << 30;
    const uint32_t sb = (((uint32_t)q + 1u) & 2u) << 30;
    sn = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a) ^ sa);
    cs = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, b) ^ sb);
#endif
}
// sin(pi x), cos(pi x) for x in [0, 1] (scaled states s~): q = rint(2x), r = x - q/2 in [-1/4, 1/4]
__device__ __forceinline__ void sincospi01(float x, float& sn, float& cs) {
    const float q = rintf(x * 2.0f);
    const float r = fmaf(q, -0.5f, x);                                  // exact
    const float u = r * r;
    float ps = 0.08100174367427826f;
    ps = fmaf(ps, u, -0.5992020964622498f);
    ps = fmaf(ps, u, 2.5501625537872314f);
    ps = fmaf(ps, u, -5.167712688446045f);
    ps = fmaf(ps, u, 3.1415927410125732f);
    const float s = ps * r;                                             // sin(pi r)
    float pc = 0.23132924735546112f;
    pc = fmaf(pc, u, -1.335044503211975f);
    pc = fmaf(pc, u, 4.058707237243652f);
    pc = fmaf(pc, u, -4.934802055358887f);
    const float c = fmaf(pc, u, 1.0f);                                  // cos(pi r)
    quadrant_select((int)q, s, c, sn, cs);                              // q = 0, 1, 2
}
// the same for two arguments at once (two state dimensions side by side in a register pair): v_pk_mul_f32 / v_pk_fma_f32
// round each half exactly like the scalar instructions, so the results are bit-identical to two sincospi01 calls
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat2(float x) { return f2{x, x}; }
__device__ __forceinline__ void sincospi01_x2(f2 x, f2& sn, f2& cs) {
    const f2 x2 = x * splat2(2.0f);
    const f2 q = f2{rintf(x2.x), rintf(x2.y)};
    const f2 r = __builtin_elementwise_fma(q, splat2(-0.5f), x);
    const f2 u = r * r;
    f2 ps = splat2(0.08100174367427826f);
    ps = __builtin_elementwise_fma(ps, u, splat2(-0.5992020964622498f));
    ps = __builtin_elementwise_fma(ps, u, splat2(2.5501625537872314f));
    ps = __builtin_elementwise_fma(ps, u, splat2(-5.167712688446045f));
    ps = __builtin_elementwise_fma(ps, u, splat2(3.1415927410125732f));
    const f2 s = ps * r;
    f2 pc = splat2(0.23132924735546112f);
    pc = __builtin_elementwise_fma(pc, u, splat2(-1.335044503211975f));
    pc = __builtin_elementwise_fma(pc, u, splat2(4.058707237243652f));
    pc = __builtin_elementwise_fma(pc, u, splat2(-4.934802055358887f));
    const f2 c = __builtin_elementwise_fma(pc, u, splat2(1.0f));
    float s0, c0, s1, c1;
    quadrant_select((int)q.x, s.x, c.x, s0, c0);
    quadrant_select((int)q.y, s.y, c.y, s1, c1);
    sn = f2{s0, s1}; cs = f2{c0, c1};
}
// {a[SA], b[SB]}: a register pair assembled from halves of two others (two v_mov_b32; the one-instruction v_pk_mov_b32 spelling measured 1 % slower
// in k_train_reg, round 4: scripts/ab/round6_pruned_knobs.patch)
template <int SA, int SB>
__device__ __forceinline__ f2 pk_pick(f2 a, f2 b) { return f2{SA ? a.y : a.x, SB ? b.y : b.x}; }
// sin(x), cos(x) for |x| <= 100: Cody-Waite reduction by pi/2 (two-term, fma), polynomials on [-pi/4, pi/4]
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs) {
    const float n = rintf(x * 0.6366197466850281f);
    float r = fmaf(n, -1.5707963705062866f, x);
    r = fmaf(n, 4.371138828673793e-08f, r);
    const f2 u = splat2(r * r);
    f2 p = f2{2.715809387154877e-06f, 2.4362980184378102e-05f};        // (sine, cosine) polynomials side by side: same Horner
    p = __builtin_elementwise_fma(p, u, f2{-0.00019839033484458923f, -0.001388643286190927f});   // steps, one packed fma each
    p = __builtin_elementwise_fma(p, u, f2{0.008333328180015087f, 0.04166661202907562f});
    p = __builtin_elementwise_fma(p, u, f2{-0.1666666716337204f, -0.5f});
    p = __builtin_elementwise_fma(p, u, splat2(1.0f));
    const float s = p.x * r;
    const float c = p.y;
    quadrant_select(((int)n) & 3, s, c, sn, cs);
}
__device__ __forceinline__ float cos_cw(float x) { float s, c; sincos_cw(x, s, c); return c; }
// ---- wave-uniform evaluation (one learner per wavefront: every lane holds the same scalars).  Independent evaluations of one
// function are spread over lanes 0, 1, 2 and read back with v_readlane: one instruction stream instead of three.  Lane k runs
// exactly the instructions every lane would have run on argument k, so the bits are unchanged.  All 64 lanes must be active.
__device__ __forceinline__ float lane_bcast(float v, int k) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
__device__ __forceinline__ float lane_pick3(int lane, float x0, float x1, float x2) {
    const float t = (lane == 2) ? x2 : x0;
    return (lane == 1) ? x1 : t;
}
// exp(x) for the softmax policy: n = rint(x log2 e), two-term Cody-Waite reduction by ln 2, degree-5 polynomial on top of
// 1 + r, scaled by 2^n (v_ldexp_f32).  Below -87 the result is 0 (no denormals), above 88.5 +inf.  Written out (instead
// of expf -> v_exp_f32, whose bits are unspecified) so that the test oracle can restate it and compare the softmax paths
// bitwise; <= 1.5 ulp.
__device__ __forceinline__ float exp_dev(float x) {
    const float n = rintf(x * 1.4426950216293335f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.4286067653302337e-06f, r);
    float p = 1.9875691e-4f;
    p = fmaf(p, r, 1.3981999e-3f);
    p = fmaf(p, r, 8.3334519e-3f);
    p = fmaf(p, r, 4.1665795e-2f);
    p = fmaf(p, r, 1.6666665e-1f);
    p = fmaf(p, r, 5.0000001e-1f);
    p = fmaf(p, r * r, r);
    p = p + 1.0f;
    const float y = ldexpf(p, (int)n);
    return (x < -87.0f) ? 0.0f : ((x > 88.5f) ? __builtin_inff() : y);
}

template <int DOMAIN> struct Domain;

// MountainCar          rsrl_domains/src/mountain_car/discrete.rs:8-102
template <> struct Domain<0> {
    static constexpr int D = 2, A = 3;
    __host__ __device__ static constexpr double lo_d(int i) { return i == 0 ? -1.2 : -0.07; }
    __host__ __device__ static constexpr double hi_d(int i) { return i == 0 ? 0.6 : 0.07; }
    __device__ static __forceinline__ void reset(float (&s)[D]) { s[0] = -0.5f; s[1] = 0.0f; }   // :68-70
    __device__ static __forceinline__ bool is_terminal(const float (&s)[D]) { return s[0] >= 0.6f; }  // :76-82
    // update_state + dv (:58-65): v first, the NEW v moves x; no velocity reset at the left wall
    // the action-independent part of the transition: the fused loop evaluates it as soon as the state is known, long before
    // the next action is chosen, which takes the cos chain off the step's critical path (policy -> action -> transition)
    struct Pre { float cos3x; };
    __device__ static __forceinline__ Pre pre(const float (&s)[D]) { return Pre{cos_cw(3.0f * s[0])}; }
    __device__ static __forceinline__ bool step(float (&s)[D], int a, float& r) { return step(s, a, r, pre(s)); }
    __device__ static __forceinline__ bool step(float (&s)[D], int a, float& r, const Pre& p) {
        const float act = (float)(a - 1);                              // ALL_ACTIONS [-1,0,1] (:22)
        const float dv = 0.001f * act + -0.0025f * p.cos3x;
        const float v = clipf(-0.07f, s[1] + dv, 0.07f);
        const float x = clipf(-1.2f, s[0] + v, 0.6f);
        s[0] = x; s[1] = v;
        const bool term = x >= 0.6f;
        r = term ? 0.0f : -1.0f;                                       // REWARD_GOAL / REWARD_STEP (:19-20)
        return term;
    }
};

// x / d for a COMPILE-TIME constant d, bit for bit the correctly rounded fp32 quotient the reference's `/` is (ode.rs:36 `/ 6.0`,
// cart_pole.rs:60 `/ TOTAL_MASS`, the tile coder's (s - lo) / (hi - lo)): through f64 -- convert, ONE multiplication by RN64(1 / d),
// convert back: three instructions where the IEEE fp32 division expands to ~11 (v_div_scale x 2, v_rcp, four fmas, v_div_fmas,
// v_div_fixup).  The double product is within 2^-52 of x / d, and x / d is never that close to a rounding boundary of fp32 unless it is
// representable (d x midpoint has more than 24 significant bits), so the second rounding decides as the first would have.  PROVEN for
// every divisor on the path by trying all 2^32 inputs -- zeros of both signs, denormals, infinities, NaNs -- on the CPU:
// oracle/check_constdiv.c, tests/test_oracle_round4.py::test_constant_division_is_the_ieee_quotient.  The oracle keeps the `/`.
__device__ __forceinline__ float div_const(float x, float d) { return (float)((double)x * (1.0 / (double)d)); }
// (int)floorf(x), saturating, NaN -> 0: ONE instruction (v_cvt_flr_i32_f32) instead of v_floor_f32 + v_cvt_i32_f32
__device__ __forceinline__ int floor_to_int(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return (int)floorf(x);
#endif
}

// classical RK4, f ignores time      rsrl_domains/src/ode.rs:1-43
template <class Grad>
__device__ __forceinline__ void rk4(const Grad& f, float (&y)[4], float dx) {
    float k1[4], k2[4], k3[4], k4[4], t[4];
    f(y, k1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { k1[i] *= dx; t[i] = y[i] + k1[i] / 2.0f; }
    f(t, k2);
#pragma unroll
    for (int i = 0; i < 4; ++i) { k2[i] *= dx; t[i] = y[i] + k2[i] / 2.0f; }
    f(t, k3);
#pragma unroll
    for (int i = 0; i < 4; ++i) { k3[i] *= dx; t[i] = y[i] + k3[i]; }
    f(t, k4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k4[i] *= dx;
        y[i] += div_const(k1[i] + 2.0f * k2[i] + 2.0f * k3[i] + k4[i], 6.0f);      // (/ 6.0: the same bits, a third of the instructions)
    }
}

// CartPole             rsrl_domains/src/cart_pole.rs:7-121, consts.rs:4-10
template <> struct Domain<1> {
    static constexpr int D = 4, A = 2;
    __host__ __device__ static constexpr double lo_d(int i) {
        return i == 0 ? -2.4 : i == 1 ? -6.0 : i == 2 ? -(kPi / 15.0) : -2.0;
    }
    __host__ __device__ static constexpr double hi_d(int i) { return -lo_d(i); }
    __device__ static __forceinline__ void reset(float (&s)[D]) { s[0] = s[1] = s[2] = s[3] = 0.0f; }   // :75-77
    __device__ static __forceinline__ bool is_terminal(const float (&s)[D]) {                          // :83-97
        constexpr float TW = (float)(kPi / 15.0);
        return s[0] <= -2.4f || s[0] >= 2.4f || s[2] <= -TW || s[2] >= TW;
    }
    struct Pre {};                                                    // nothing of the RK4 step is action-independent
    __device__ static __forceinline__ Pre pre(const float (&)[D]) { return Pre{}; }
    __device__ static __forceinline__ bool step(float (&s)[D], int a, float& r, const Pre&) { return step(s, a, r); }
    __device__ static __forceinline__ bool step(float (&s)[D], int a, float& r) {
        constexpr float TW = (float)(kPi / 15.0);
        const float force = (a == 0) ? -10.0f : 10.0f;                 // ALL_ACTIONS (:26)
        auto grad = [force](const float (&y)[4], float (&out)[4]) {    // CartPole::grad (:52-72)
            constexpr float G = 9.8f, FOUR_THIRDS = (float)(4.0 / 3.0);
            constexpr float POLE_COM = 0.5f, POLE_MOMENT = 0.5f * 0.1f, TOTAL_MASS = 1.0f + 0.1f;
            const float dx = y[1], theta = y[2], dtheta = y[3];
            float sin_t, cos_t;
            sincos_cw(theta, sin_t, cos_t);
            const float z = div_const(force + POLE_MOMENT * dtheta * dtheta * sin_t, TOTAL_MASS);
            const float numer = G * sin_t - cos_t * z;
            const float denom = FOUR_THIRDS * POLE_COM - POLE_MOMENT * cos_t * cos_t;
            const float ddtheta = numer / denom;
            out[0] = dx; out[2] = dtheta; out[3] = ddtheta;
            out[1] = z - POLE_COM * ddtheta * cos_t;
        };
        float ns[4] = {s[0], s[1], s[2], s[3]};
        rk4(grad, ns, 0.02f);                                          // update_state (:39-50)
        s[0] = clipf(-2.4f, ns[0], 2.4f);
        s[1] = clipf(-6.0f, ns[1], 6.0f);
        s[2] = clipf(-TW, ns[2], TW);
        s[3] = clipf(-2.0f, ns[3], 2.0f);
        const bool term = is_terminal(s);
        r = term ? -1.0f : 0.0f;                                       // REWARD_TERMINAL / REWARD_STEP (:23-24)
        return term;
    }
};

// Acrobot              rsrl_domains/src/acrobot.rs:8-152
template <> struct Domain<2> {
    static constexpr int D = 4, A = 3;
    __host__ __device__ static constexpr double lo_d(int i) {
        return i == 0 ? -kPi : i == 1 ? -kPi : i == 2 ? -4.0 * kPi : -9.0 * kPi;
    }
    __host__ __device__ static constexpr double hi_d(int i) { return -lo_d(i); }
    __device__ static __forceinline__ void reset(float (&s)[D]) { s[0] = s[1] = s[2] = s[3] = 0.0f; }   // :111-113
    __device__ static __forceinline__ bool is_terminal(const float (&s)[D]) {                          // :56-58
        return cos_cw(s[0]) + cos_cw(s[0] + s[1]) < -1.0f;
    }
    struct Pre {};
    __device__ static __forceinline__ Pre pre(const float (&)[D]) { return Pre{}; }
    __device__ static __forceinline__ bool step(float (&s)[D], int a, float& r, const Pre&) { return step(s, a, r); }
    __device__ static __forceinline__ bool step(float (&s)[D], int a, float& r) {
        constexpr float PI_ = (float)kPi;
        const float torque = (float)(a - 1);                           // ALL_ACTIONS (:35-36)
        auto grad = [torque](const float (&y)[4], float (&out)[4]) {   // Acrobot::grad (:81-108)
            constexpr float M1 = 1.0f, M2 = 1.0f, L1 = 1.0f, LC1 = 0.5f, LC2 = 0.5f, I1 = 1.0f, I2 = 1.0f, G = 9.8f;
            constexpr float PI_OVER_2 = (float)(kPi / 2.0);
            const float theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
            float sin_t2, cos_t2;
            sincos_cw(theta2, sin_t2, cos_t2);
            const float d1 = M1 * LC1 * LC1 + M2 * (L1 * L1 + LC2 * LC2 + 2.0f * L1 * LC2 * cos_t2) + I1 + I2;
            const float d2 = M2 * (LC2 * LC2 + L1 * LC2 * cos_t2) + I2;
            const float phi2 = M2 * LC2 * G * cos_cw(theta1 + theta2 - PI_OVER_2);
            const float phi1 = -1.0f * L1 * LC2 * dtheta2 * dtheta2 * sin_t2
                             - 2.0f * M2 * L1 * LC2 * dtheta2 * dtheta1 * sin_t2
                             + (M1 * LC1 + M2 * L1) * G * cos_cw(theta1 - PI_OVER_2)
                             + phi2;
            const float dd1 = (torque + d2 / d1 * phi1 - M2 * L1 * LC2 * dtheta1 * dtheta1 * sin_t2 - phi2)
                            / (M2 * LC2 * LC2 + I2 - d2 * d2 / d1);
            out[0] = dtheta1; out[1] = dtheta2; out[2] = dd1;
            out[3] = -(d2 * dd1 + phi1) / d1;
        };
        float ns[4] = {s[0], s[1], s[2], s[3]};
        rk4(grad, ns, 0.2f);                                           // update_state (:60-79)
        s[0] = wrapf(-PI_, ns[0], PI_);
        s[1] = wrapf(-PI_, ns[1], PI_);
        s[2] = clipf(-4.0f * PI_, ns[2], 4.0f * PI_);
        s[3] = clipf(-9.0f * PI_, ns[3], 9.0f * PI_);
        const bool term = is_terminal(s);
        r = term ? 0.0f : -1.0f;                                       // REWARD_TERMINAL / REWARD_STEP (:31-32)
        return term;
    }
    // the same transition for a WAVE-UNIFORM state (wave family): the three trigonometric evaluations of a gradient -- and the two
    // of the terminal test -- run side by side in lanes 0..2 (one sincos instead of three per gradient: 14 -> 5 per step)
    __device__ static __forceinline__ bool step_uniform(float (&s)[D], int a, float& r, int lane) {
        constexpr float PI_ = (float)kPi;
        const float torque = (float)(a - 1);
        auto grad = [torque, lane](const float (&y)[4], float (&out)[4]) {
            constexpr float M1 = 1.0f, M2 = 1.0f, L1 = 1.0f, LC1 = 0.5f, LC2 = 0.5f, I1 = 1.0f, I2 = 1.0f, G = 9.8f;
            constexpr float PI_OVER_2 = (float)(kPi / 2.0);
            const float theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
            float sn, cs;
            sincos_cw(lane_pick3(lane, theta2, theta1 + theta2 - PI_OVER_2, theta1 - PI_OVER_2), sn, cs);
            const float sin_t2 = lane_bcast(sn, 0), cos_t2 = lane_bcast(cs, 0), cos_12 = lane_bcast(cs, 1), cos_1 = lane_bcast(cs, 2);
            const float d1 = M1 * LC1 * LC1 + M2 * (L1 * L1 + LC2 * LC2 + 2.0f * L1 * LC2 * cos_t2) + I1 + I2;
            const float d2 = M2 * (LC2 * LC2 + L1 * LC2 * cos_t2) + I2;
            const float phi2 = M2 * LC2 * G * cos_12;
            const float phi1 = -1.0f * L1 * LC2 * dtheta2 * dtheta2 * sin_t2
                             - 2.0f * M2 * L1 * LC2 * dtheta2 * dtheta1 * sin_t2
                             + (M1 * LC1 + M2 * L1) * G * cos_1
                             + phi2;
            const float dd1 = (torque + d2 / d1 * phi1 - M2 * L1 * LC2 * dtheta1 * dtheta1 * sin_t2 - phi2)
                            / (M2 * LC2 * LC2 + I2 - d2 * d2 / d1);
            out[0] = dtheta1; out[1] = dtheta2; out[2] = dd1;
            out[3] = -(d2 * dd1 + phi1) / d1;
        };
        float ns[4] = {s[0], s[1], s[2], s[3]};
        rk4(grad, ns, 0.2f);
        s[0] = wrapf(-PI_, ns[0], PI_);
        s[1] = wrapf(-PI_, ns[1], PI_);
        s[2] = clipf(-4.0f * PI_, ns[2], 4.0f * PI_);
        s[3] = clipf(-9.0f * PI_, ns[3], 9.0f * PI_);
        float sn, cs;
        sincos_cw(lane_pick3(lane, s[0], s[0] + s[1], s[0]), sn, cs);
        const bool term = lane_bcast(cs, 0) + lane_bcast(cs, 1) < -1.0f;
        r = term ? 0.0f : -1.0f;
        return term;
    }
};

// ---------------------------------------------------------------------------------------
// argmax helpers and policies
// ---------------------------------------------------------------------------------------

// Enumerable::find_max: fold `if acc.1 > x {acc} else {(i,x)}` => ties go to the LAST index   core.rs:96-105
template <int A>
__device__ __forceinline__ int find_max(const float (&q)[A], float& val) {
    int bi = 0; float bv = q[0];
#pragma unroll
    for (int i = 1; i < A; ++i) { if (!(bv > q[i])) { bi = i; bv = q[i]; } }
    val = bv;
    return bi;
}
// argmaxima: tolerance test first, running max NOT raised by near-ties      utils.rs:6-21
// returns the set as a bitmask (A <= 8 on this path)
template <int A>
__device__ __forceinline__ uint32_t argmaxima_mask(const float (&q)[A]) {
    // i = 0 against mx = -FLT_MAX, folded: the tolerance test can only hit q[0] == -FLT_MAX itself, so q[0] joins the set
    // iff q[0] >= -FLT_MAX (not NaN, not -inf) and the running max becomes max(q[0], -FLT_MAX) either way
    // (v_med3_f32 with +inf: fmaxf's value for every input, NaN included -- the median of three returns min3 when an operand is NaN --
    // without the canonicalising v_max_f32 x, x the compiler puts in front of fmaxf when x comes out of a bitwise select)
#if defined(__HIP_DEVICE_COMPILE__)
    float mx = __builtin_amdgcn_fmed3f(q[0], -FLT_MAX, __builtin_inff());
#else
    float mx = fmaxf(q[0], -FLT_MAX);
#endif
    uint32_t mask = (q[0] >= -FLT_MAX) ? 1u : 0u;
#pragma unroll
    for (int i = 1; i < A; ++i) {                                     // selects only: no exec-mask branches
        const float qi = q[i];
        const bool near = fabsf(qi - mx) < 1e-7f;
        const bool up = (!near) & (qi > mx);
        const uint32_t m_or = mask | (1u << i), m_up = up ? (1u << i) : mask;
        mask = near ? m_or : m_up;
        mx = up ? qi : mx;
    }
    return mask;
}
// argmax_first: index moves only if y - x > 1e-7                             utils.rs:23-34
template <int A>
__device__ __forceinline__ int argmax_first(const float (&v)[A]) {
    int bi = 0; float bx = -FLT_MAX;
#pragma unroll
    for (int j = 0; j < A; ++j) { if (v[j] - bx > 1e-7f) { bi = j; bx = v[j]; } }
    return bi;
}
// position of the (k+1)-th set bit of mask, k < popc(mask).  A <= 3: one lookup in a 48-bit table (2 bits per (mask, k)).
template <int A>
__device__ __forceinline__ int kth_set_bit(uint32_t mask, int k) {
    if constexpr (A <= 3) {
        constexpr uint64_t table = [] {
            uint64_t t = 0;
            for (uint32_t m = 0; m < 8; ++m)
                for (int kk = 0; kk < 3; ++kk) {
                    int seen = 0, pos = 0;
                    for (int b = 0; b < 3; ++b)
                        if ((m >> b) & 1u) { if (seen == kk) pos = b; ++seen; }
                    t |= (uint64_t)pos << (2 * (m * 3 + kk));
                }
            return t;
        }();
        const uint32_t sh = 2u * (mask * 3u + (uint32_t)k);
        return (int)((uint32_t)(table >> sh) & 3u);
    } else {
#pragma unroll
        for (int j = 0; j < A - 1; ++j) {                             // drop the k lowest set bits
            const uint32_t dropped = mask & (mask - 1);
            mask = (j < k) ? dropped : mask;
        }
        return __ffs((int)mask) - 1;
    }
}
// Greedy::sample -> argmax_choose_rng: the single maximum, else a uniform pick among the
// maxima with the caller's rng                                 greedy.rs:77-81, utils.rs:63-79
template <int A>
__device__ __forceinline__ int greedy_sample(const float (&q)[A], uint32_t x_tie) {
    const uint32_t m0 = argmaxima_mask<A>(q);
    const int n0 = __popc(m0);
    // no maximum at all (every Q is NaN or -inf: a diverged learner; the reference panics with "No valid maxima",
    // utils.rs:70-76): the action indexes a weight column, so it must stay in [0, A) -- a uniform pick among all A
    const bool none = n0 == 0;
    const uint32_t mask = none ? ((1u << A) - 1u) : m0;
    const int n = none ? A : n0;
    return kth_set_bit<A>(mask, (int)mulhi_u32(x_tie, (uint32_t)n));       // a single maximum: k = 0
}
// caller-supplied actions index weight columns: keep them in [0, A) whatever the caller passed
template <int A>
__device__ __forceinline__ int clamp_action(int a) { return a < 0 ? 0 : (a > A - 1 ? A - 1 : a); }
// softmax_stable + softmax                                                   softmax.rs:15-37
// ulane >= 0: q is wave-uniform and all 64 lanes are active -- element i is evaluated by lane i only (one exponential and one
// division instead of A of each) and broadcast
template <int A>
__device__ __forceinline__ void softmax_probs(const float (&q)[A], float tau, float (&p)[A], int ulane = -1) {
    float m = q[0];
#pragma unroll
    for (int i = 1; i < A; ++i) m = (q[i] > m) ? q[i] : m;
    if (ulane >= 0) {
        static_assert(A <= 3, "lane_pick3");
        const float ql = lane_pick3(ulane, q[0], q[A > 1 ? 1 : 0], q[A > 2 ? 2 : 0]);
        const float el = exp_dev((ql - m) / tau);
        float z = 0.0f;
#pragma unroll
        for (int i = 0; i < A; ++i) { p[i] = lane_bcast(el, i); z += p[i]; }
        const float dl = fminf(el / z, FLT_MAX);
#pragma unroll
        for (int i = 0; i < A; ++i) p[i] = lane_bcast(dl, i);
        return;
    }
    float z = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) { p[i] = exp_dev((q[i] - m) / tau); z += p[i]; }
#pragma unroll
    for (int i = 0; i < A; ++i) p[i] = fminf(p[i] / z, FLT_MAX);
}
// sample_probs_with_rng: first index whose cumulative probability exceeds u, else the last   policies/mod.rs:45-61
template <int A>
__device__ __forceinline__ int sample_probs(const float (&p)[A], uint32_t x) {
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    float acc = 0.0f; int res = A - 1; bool found = false;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        acc = acc + p[i];
        if (!found && acc > u) { res = i; found = true; }
    }
    return res;
}
enum : int { POL_GREEDY = 0, POL_EGREEDY = 1, POL_SOFTMAX = 2, POL_RANDOM = 3 };
enum : int { ALG_QLEARNING = 0, ALG_SARSA = 1, ALG_ESARSA = 2, ALG_PAL = 5 };

struct PolicyParams { int kind; uint32_t eps_thr; float eps; float tau; };

// Policy::sample.  x.x = explore draw, x.y = random action, x.z = tie-break / softmax u.
//   EpsilonGreedy::sample  epsilon_greedy.rs:74-80 (gen_bool(eps) ? Random : Greedy)
//   Random::sample         random.rs:43-45         (Uniform(0, A))
//   Softmax::sample        softmax.rs:131-139
// YZ: the caller guarantees x.y == x.z (the per-step draws, half_block).  EpsilonGreedy then needs ONE scaled pick instead of two:
// a random action is the pick among the full set {0..A-1}, kth_set_bit(2^A - 1, k) = k -- the same action as mulhi(x.y, A) -- and
// the empty set of maxima already took that route.  One quarter-rate multiply and two selects less per sample; the same results.
template <int A, bool YZ = false>
__device__ __forceinline__ int policy_sample(const PolicyParams& pp, const float (&q)[A], const U4& x, int ulane = -1) {
    switch (pp.kind) {
    case POL_GREEDY: return greedy_sample<A>(q, x.z);
    case POL_EGREEDY: {
        if constexpr (YZ) {
            const uint32_t m0 = argmaxima_mask<A>(q);
            const bool explore = (x.x >> 8) < pp.eps_thr;
            const uint32_t mask = (explore | (m0 == 0u)) ? ((1u << A) - 1u) : m0;
            return kth_set_bit<A>(mask, (int)mulhi_u32(x.y, (uint32_t)__popc(mask)));
        }
        const int g = greedy_sample<A>(q, x.z), u = (int)mulhi_u32(x.y, (uint32_t)A);
        const bool explore = (x.x >> 8) < pp.eps_thr;
        return explore ? u : g;
    }
    case POL_SOFTMAX: { float p[A]; softmax_probs<A>(q, pp.tau, p, ulane); return sample_probs<A>(p, x.z); }
    default: return (int)mulhi_u32(x.y, (uint32_t)A);
    }
}
// Policy::mode   greedy.rs:83 (find_max), epsilon_greedy.rs:82, softmax.rs:141-143 (argmax_first of probs)
template <int A>
__device__ __forceinline__ int policy_mode(const PolicyParams& pp, const float (&q)[A]) {
    if (pp.kind == POL_SOFTMAX) { float p[A]; softmax_probs<A>(q, pp.tau, p); return argmax_first<A>(p); }
    float v; return find_max<A>(q, v);
}
// Function<(S,)> of the policy: action probabilities
//   greedy.rs:30-44, epsilon_greedy.rs:38-45, softmax.rs:74-82, random.rs:22-26
template <int A>
__device__ __forceinline__ void policy_probs(const PolicyParams& pp, const float (&q)[A], float (&p)[A], int ulane = -1) {
    if (pp.kind == POL_SOFTMAX) { softmax_probs<A>(q, pp.tau, p, ulane); return; }
    if (pp.kind == POL_RANDOM) {
#pragma unroll
        for (int i = 0; i < A; ++i) p[i] = 1.0f / (float)A;
        return;
    }
    const uint32_t mask = argmaxima_mask<A>(q);
    const float pg = 1.0f / (float)max(1, __popc(mask));      // empty set (all Q NaN / -inf): all-zero greedy part
#pragma unroll
    for (int i = 0; i < A; ++i) p[i] = ((mask >> i) & 1u) ? pg : 0.0f;
    if (pp.kind == POL_EGREEDY) {
        const float pr = pp.eps / (float)A;
#pragma unroll
        for (int i = 0; i < A; ++i) p[i] = pr + p[i] * (1.0f - pp.eps);
    }
}

// Domain::rollout's closure (lib.rs:448-479 takes any FnMut(&S) -> A): sample = 0 -> the ctx's policy.mode (the README's greedy
// evaluation); 1 -> policy.sample of `pp`, the k-th selection of a learner drawing the whole Philox block ((call << 32) | k, BLK_ROLLOUT)
struct RolloutPolicy { int sample; PolicyParams pp; uint64_t call; };
template <int A>
__device__ __forceinline__ int rollout_action(const PolicyParams& mode_pol, const RolloutPolicy& rp, const float (&q)[A], uint64_t seed, uint32_t gid, uint64_t k) {
    if (!rp.sample) return policy_mode<A>(mode_pol, q);
    return policy_sample<A>(rp.pp, q, draw_block(seed, gid, (rp.call << 32) | (k & 0xffffffffull), BLK_ROLLOUT));
}

struct AlgoParams { int kind; float gamma, lr, alpha; };

// TD error of the three agents from Q(s,a), Q(s',.) (PRE-update W)
//   QLearning::handle q_learning.rs:51-71 | SARSA::handle sarsa.rs:53-75 | ExpectedSARSA::handle expected_sarsa.rs:45-66
// returns delta; `e` is the error sent to the approximator (alpha*delta for ExpectedSARSA, :64)
template <int A>
__device__ __forceinline__ float td_error(const AlgoParams& ap, const PolicyParams& pp, float qsa, const float (&qn)[A],
                                          float r, bool term, const U4& x_inner, float& e, int ulane = -1) {
    // the bootstrap value is computed whether or not the transition is terminal and SELECTED afterwards (no state is
    // consumed by it: the draws are counter-based), so that the step is one basic block
    float boot;
    if (ap.kind == ALG_QLEARNING) {
        find_max<A>(qn, boot);
    } else if (ap.kind == ALG_SARSA) {
        const int na = policy_sample<A>(pp, qn, x_inner, ulane);      // agent's own draw (sarsa.rs:61)
        boot = qn[0];
#pragma unroll
        for (int i = 1; i < A; ++i) boot = (na == i) ? qn[i] : boot;
    } else {
        float p[A]; policy_probs<A>(pp, qn, p, ulane);
        boot = 0.0f;
#pragma unroll
        for (int i = 0; i < A; ++i) boot = boot + qn[i] * p[i];       // fold(0.0, acc + q*p)
    }
    const float d_term = r - qsa, d_boot = r + ap.gamma * boot - qsa;
    const float delta = term ? d_term : d_boot;
    e = (ap.kind == ALG_ESARSA) ? ap.alpha * delta : delta;
    return delta;
}

// PAL::handle  control/td/pal.rs:34-60 (persistent advantage learning): needs Q(s,.) in full.
//   td = r + gamma*Q(s',a*) - Q(s,a);  residual = max(td - alpha*(Q(s,a*) - Q(s,a)), td - alpha*(Q(s',na*) - Q(s',a)))
//   a* / na* = argmax_first (utils.rs:23-34); terminal: r - Q(s,a); the error sent on is alpha * residual (:57)
template <int A>
__device__ __forceinline__ float td_error_pal(const AlgoParams& ap, const float (&qs)[A], const float (&qn)[A], int a, float r,
                                              bool term, float& e) {
    float qsa = qs[0], qna = qn[0];
#pragma unroll
    for (int i = 1; i < A; ++i) { qsa = (a == i) ? qs[i] : qsa; qna = (a == i) ? qn[i] : qna; }
    const int as = argmax_first<A>(qs), nas = argmax_first<A>(qn);
    float qs_star = qs[0], qn_at_as = qn[0], qn_star = qn[0];
#pragma unroll
    for (int i = 1; i < A; ++i) {
        qs_star = (as == i) ? qs[i] : qs_star; qn_at_as = (as == i) ? qn[i] : qn_at_as; qn_star = (nas == i) ? qn[i] : qn_star;
    }
    const float td = r + ap.gamma * qn_at_as - qsa;
    const float al = td - ap.alpha * (qs_star - qsa);
    const float alt = td - ap.alpha * (qn_star - qna);
    const float delta = term ? (r - qsa) : fmaxf(al, alt);
    e = ap.alpha * delta;
    return delta;
}

// one entry point for all one-step agents: Q(s,.) in full, the action taken, Q(s',.) -> (delta, error sent on)
template <int A>
__device__ __forceinline__ float td_dispatch(const AlgoParams& ap, const PolicyParams& pp, const float (&qs)[A], int a,
                                             const float (&qn)[A], float r, bool term, const U4& x_inner, float& e, int ulane = -1) {
    if (ap.kind == ALG_PAL) return td_error_pal<A>(ap, qs, qn, a, r, term, e);
    float qsa = qs[0];
#pragma unroll
    for (int i = 1; i < A; ++i) qsa = (a == i) ? qs[i] : qsa;
    return td_error<A>(ap, pp, qsa, qn, r, term, x_inner, e, ulane);
}

// ---------------------------------------------------------------------------------------
// Fourier basis (lfa::basis::Fourier + with_bias, call site examples/q_learning.rs:24).
// F = (ORDER+1)^D; coefficient vectors in lexicographic order (last dim fastest), all-zero
// skipped, constant 1 LAST.  Separable evaluation: per dimension ONE sincospi + the
// angle-addition chain, then a complex product over dimensions -- 2 transcendental calls
// instead of 35 for MountainCar order 5, and closer to the f64 value than cos(pi*fl(c.s~)).
// ---------------------------------------------------------------------------------------
template <int DOMAIN, int ORDER>
struct FourierTables {
    using Dom = Domain<DOMAIN>;
    static constexpr int D = Dom::D, N1 = ORDER + 1;
    float ct[D][N1], st[D][N1];
    __device__ __forceinline__ void build(const float (&s)[D]) {
        static_for<0, D>([&](auto Dd) {
            constexpr int d = Dd;
            constexpr float lo = (float)Dom::lo_d(d), hi = (float)Dom::hi_d(d);
            // s~ = (s - lo) * fl(1/(hi - lo)): a true division is a ~10-deep dependent sequence on the critical path of every
            // step; the f32 oracle mirrors the multiply, the f64 oracle keeps the reference's division (<= 1.5 ulp apart)
            constexpr float inv = 1.0f / (hi - lo);
            const float sc = (s[d] - lo) * inv;
            ct[d][0] = 1.0f; st[d][0] = 0.0f;
            if constexpr (ORDER >= 1) sincospi01(sc, st[d][1], ct[d][1]);
            static_for<2, N1>([&](auto Nn) {
                constexpr int n = Nn;
                ct[d][n] = fmaf(-st[d][n - 1], st[d][1], ct[d][n - 1] * ct[d][1]);
                st[d][n] = fmaf(ct[d][n - 1], st[d][1], st[d][n - 1] * ct[d][1]);
            });
        });
    }
};

// register-resident projection: every feature index is a compile-time constant.
// Even D: the per-dimension tables are built for two dimensions at a time in register pairs (packed fp32 multiply / fma
// take one issue slot for both), and for D = 2 the products are formed two features at a time as well.  Same operations in
// the same order per element as FourierTables::build + the scalar product below: bit-identical features.
template <int DOMAIN, int ORDER>
struct FourierReg {
    using Dom = Domain<DOMAIN>;
    static constexpr int D = Dom::D, N1 = ORDER + 1, F = ipow(N1, D);
    static constexpr bool kPairs = (D % 2 == 0) && ORDER >= 1;
    struct PairTables {
        f2 ct[D / 2 > 0 ? D / 2 : 1][N1], st[D / 2 > 0 ? D / 2 : 1][N1];
        __device__ __forceinline__ float c(int d, int n) const { return (d & 1) ? ct[d >> 1][n].y : ct[d >> 1][n].x; }
        __device__ __forceinline__ float s(int d, int n) const { return (d & 1) ? st[d >> 1][n].y : st[d >> 1][n].x; }
        __device__ __forceinline__ void build(const float (&sv)[D]) {
            static_for<0, D / 2>([&](auto Pp) {
                constexpr int p = Pp, d0 = 2 * p, d1 = 2 * p + 1;
                constexpr float lo0 = (float)Dom::lo_d(d0), hi0 = (float)Dom::hi_d(d0), lo1 = (float)Dom::lo_d(d1), hi1 = (float)Dom::hi_d(d1);
                constexpr float inv0 = 1.0f / (hi0 - lo0), inv1 = 1.0f / (hi1 - lo1);       // as FourierTables::build
                const f2 sc = (f2{sv[d0], sv[d1]} - f2{lo0, lo1}) * f2{inv0, inv1};
                ct[p][0] = splat2(1.0f); st[p][0] = splat2(0.0f);
                sincospi01_x2(sc, st[p][1], ct[p][1]);
                static_for<2, N1>([&](auto Nn) {
                    constexpr int n = Nn;
                    ct[p][n] = __builtin_elementwise_fma(-st[p][n - 1], st[p][1], ct[p][n - 1] * ct[p][1]);
                    st[p][n] = __builtin_elementwise_fma(ct[p][n - 1], st[p][1], st[p][n - 1] * ct[p][1]);
                });
            });
        }
    };
    // one feature (k = 1 .. F-1) from per-dimension tables
    template <int k, class T>
    __device__ static __forceinline__ float feature(const T& tb) {
        constexpr int c0 = (k / ipow(N1, D - 1)) % N1;                 // digits of k, most significant = dimension 0
        float re = tb.c(0, c0), im = tb.s(0, c0);
        static_for<1, D>([&](auto Dd) {
            constexpr int d = Dd;
            constexpr int cd = (k / ipow(N1, D - 1 - d)) % N1;
            if constexpr (cd == 0) {
                // multiply by (1, 0): exact identity
            } else if constexpr (d == 1 && c0 == 0) {
                re = tb.c(d, cd); im = tb.s(d, cd);                    // (1,0) * z == z exactly
            } else {
                const float nre = fmaf(-im, tb.s(d, cd), re * tb.c(d, cd));
                const float nim = fmaf(re, tb.s(d, cd), im * tb.c(d, cd));
                re = nre; im = nim;
            }
        });
        return re;
    }
    struct ScalarTables {
        FourierTables<DOMAIN, ORDER> t;
        __device__ __forceinline__ float c(int d, int n) const { return t.ct[d][n]; }
        __device__ __forceinline__ float s(int d, int n) const { return t.st[d][n]; }
    };
    __device__ static __forceinline__ void project(const float (&s)[D], float (&phi)[F]) {
        if constexpr (kPairs) {
            PairTables tb;
            tb.build(s);
            if constexpr (D == 2) {
                // features 2j, 2j+1 (k = 2j+1, 2j+2) with the same dimension-0 harmonic c0 >= 1 and both dimension-1 harmonics
                // >= 1: one packed multiply + one packed fma, the dimension-0 factor broadcast to both halves
                static_for<0, F / 2>([&](auto Jj) {
                    constexpr int j = Jj, ka = 2 * j + 1, kb = 2 * j + 2;
                    constexpr int a0 = ka / N1, a1 = ka % N1, b0 = kb / N1, b1 = kb % N1;
                    if constexpr (kb < F && a0 == b0 && a0 >= 1 && a1 >= 1 && b1 >= 1) {
                        const f2 c1 = pk_pick<1, 1>(tb.ct[0][a1], tb.ct[0][b1]), s1 = pk_pick<1, 1>(tb.st[0][a1], tb.st[0][b1]);
                        const f2 v = __builtin_elementwise_fma(splat2(-tb.s(0, a0)), s1, splat2(tb.c(0, a0)) * c1);
                        phi[2 * j] = v.x; phi[2 * j + 1] = v.y;
                    } else if constexpr (kb < F && a0 == 0 && b0 == 0) {
                        // row 0: the dimension-1 harmonics themselves -- the very pair the rows below multiply by
                        const f2 v = pk_pick<1, 1>(tb.ct[0][a1], tb.ct[0][b1]);
                        phi[2 * j] = v.x; phi[2 * j + 1] = v.y;
                    } else if constexpr (kb < F && a0 == 0 && b0 == 1 && b1 == 0) {
                        const f2 v = pk_pick<1, 0>(tb.ct[0][a1], tb.ct[0][1]);      // (cos of the last dimension-1 harmonic, cos of dimension 0's first)
                        phi[2 * j] = v.x; phi[2 * j + 1] = v.y;
                    } else {
                        phi[2 * j] = feature<ka>(tb);
                        if constexpr (kb < F) phi[2 * j + 1] = feature<kb>(tb);
                    }
                });
            } else {
                static_for<1, F>([&](auto Kk) { constexpr int k = Kk; phi[k - 1] = feature<k>(tb); });
            }
        } else {
            ScalarTables tb;
            tb.t.build(s);
            static_for<1, F>([&](auto Kk) { constexpr int k = Kk; phi[k - 1] = feature<k>(tb); });
        }
        phi[F - 1] = 1.0f;
    }
};

}  // namespace rsrl
