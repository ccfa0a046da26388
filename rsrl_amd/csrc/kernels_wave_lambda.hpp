// kernels_wave_lambda.hpp -- eligibility-trace control (SARSALambda / QLambda, rsrl/src/control/td/sarsa_lambda.rs:53-98,
// q_lambda.rs:56-99; trace rules rsrl/src/traces.rs:188-240) on the WAVE family: Fourier order 7 on a 4-D state space,
// F = 4096 features, one wavefront per learner (kernels_wave.hpp); weights f32, or (round 6) bf16 with stochastic rounding and the trace in f32.
//
// Every entry of W moves at every step (W += alpha * residual * Z), and W + Z are 96 KiB per learner: neither the registers of
// a wave (k_train_wave keeps W alone there, 192 of them) nor its share of the LDS hold both, so this family member is a memory
// sweep like the traces on tile coding (kernels_lambda_tile.hpp): per learner-step
//     Q(s', .) with the pre-update weights       W read once (48 KiB)
//     one fused sweep over (Z, W)                z = rule(rate * z + [b == a] * phi(s));  w += scale * z;  Q_post(s', b) += phi(s') * w
//                                                Z and W read and written once each (192 KiB), 16 B per lane, coalesced
// with the layout of the wave family (f32[N][A][F], internal index k = j*512 + lane*8 + v) for both matrices.  phi(s) and phi(s')
// live in registers (64 + 64 per lane).  Q(s', .) with the UPDATED weights -- what the behaviour policy samples from, and the
// next step's Q(s, .) -- falls out of the sweep: every column's lane partial runs over (j, v) in dot()'s order, four chains,
// then the wave total.  Element by element the operations are project() / dot() / trace_merge() / fmaf of the granular kernels:
// bit-identical to the oracle's wave-order loop (orc_run_train_wave with a lambda agent).
// bf16 weights (WT = bf16_t; the storage format of BASELINE configs[4], here for the trace agents): W is read as bf16, the arithmetic and the
// trace stay f32, and EVERY entry of W -- all of them move -- is rounded back by stochastic rounding with the lane's Philox block 16 + 64 * b + lane
// of the step (column b; the 16-bit window of element e = j * 8 + v as in k_train_wave), so that Q(s', .) of the sweep is the dot product over
// the stored values.  W traffic halves: 4 A F + 2 x 2 A F bytes per learner-step instead of 4 x 4 A F.
#pragma once

#include "kernels_wave.hpp"
#include "kernels_lambda.hpp"

namespace rsrl {

// from == nullptr: the driver loop, n_steps batch-steps of the wave's learner.  Otherwise Handler::handle on ONE caller-supplied
// transition per learner (Mn of them).
template <int DOMAIN, class WT = float>
__global__ __launch_bounds__(kBlock) void k_wave_lambda(Common c, LambdaParams lp, WT* __restrict__ Wbase, uint64_t t0, int n_steps,
                                                        DevStats* __restrict__ stats, const float* __restrict__ from, const int32_t* __restrict__ act,
                                                        const float* __restrict__ rew, const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                        int64_t Mn, float* __restrict__ td_out) {
    using WF = WaveFourier<DOMAIN>;
    using Dom = Domain<DOMAIN>;
    using IO = WaveIO<WT>;
    using IOZ = WaveIO<float>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t N = c.n_envs;
    const bool driver = from == nullptr;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);      // learner of this wave (uniform)
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < (driver ? N : Mn)) {
        const bool sarsa = c.alg.kind == ALG_SARSA_LAMBDA;
        AlgoParams alg = c.alg; alg.kind = sarsa ? ALG_SARSA : ALG_QLEARNING;      // the TD target formula
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        WT* __restrict__ Wi = Wbase + i * (int64_t)(A * F);
        float* __restrict__ Zi = lp.Z + i * (int64_t)(A * F);
        float s[D];
        int a; uint32_t ep = 0;
        if (driver) {
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
            a = __builtin_amdgcn_readfirstlane(c.action[i]); ep = c.ep_step[i];
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = from[(int64_t)d * Mn + i];
            a = clamp_action<A>(__builtin_amdgcn_readfirstlane(act[i]));
        }
        float phi_s[8][8], q_s[A];
        WF::project(s, lane, phi_s);
        WF::template q_from_mem<WT>(Wi, lane, phi_s, q_s);
        float facc_abs = 0.0f, facc_r = 0.0f;
        // the per-learner epsilon schedule (Common::eps, examples/sarsa_lambda.rs:68): the learner is the wave's, its epsilon wave-uniform
        PolicyParams pol = c.pol;
        learner_eps_load(c, i, pol);
        for (int k = 0; k < (driver ? n_steps : 1); ++k) {
            const uint64_t t = t0 + (uint64_t)k;
            float ns[D], r;
            bool term, trunc = false;
            if (driver) {
#pragma unroll
                for (int d = 0; d < D; ++d) ns[d] = s[d];
                term = Dom::step(ns, a, r);
                ep += 1;
                trunc = !term && cap > 0 && ep >= cap;
                if (term) Dom::reset(ns);
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) ns[d] = to[(int64_t)d * Mn + i];
                r = rew[i]; term = termf[i] != 0;
            }
            float phi_n[8][8], q_n[A];
            WF::project(ns, lane, phi_n);
            WF::template q_from_mem<WT>(Wi, lane, phi_n, q_n);              // PRE-update weights
            // ---- trace decay rate: Q(lambda) cuts the trace unless the action taken was argmax_first of Q(s,.)   q_lambda.rs:62-66
            float rate_eff = lp.rate;
            if (!sarsa) rate_eff = (a != argmax_first<A>(q_s)) ? 0.0f : lp.rate;
            U4 xin = U4{0, 0, 0, 0};
            if (sarsa) xin = draw(c.seed, gid, t, BLK_INNER);                  // the agent's own draw (sarsa_lambda.rs:78)
            float e;
            const float delta = td_error<A>(alg, (c.eps && c.apol_same) ? pol : c.apol, select_a<A>(q_s, a), q_n, r, term, xin, e);
            const float scale = lp.alpha * delta;
            // ---- the fused sweep: trace, weights, and Q(s', .) with the updated weights
#pragma unroll
            for (int b = 0; b < A; ++b) {
                const bool hit = a == b;                                       // wave-uniform
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                U4 rnd = U4{0, 0, 0, 0};
                if constexpr (IO::kBf16) rnd = draw(c.seed, gid, t, BLK_SR_BASE + 64u * (uint32_t)b + (uint32_t)lane);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t off = (int64_t)b * F + j * 512 + lane * 8;
                    float z8[8], w8[8];
                    IOZ::load8(Zi, off, z8);
                    IO::load8(Wi, off, w8);
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const float zz = trace_merge(lp.trace, rate_eff, z8[v], hit ? phi_s[j][v] : 0.0f);
                        float ww = fmaf(scale, zz, w8[v]);
                        if constexpr (IO::kBf16) ww = round_bf16_sr(ww, sr_bits(rnd, j * 8 + v));
                        acc[v & 3] = fmaf(phi_n[j][v], ww, acc[v & 3]);
                        z8[v] = term ? 0.0f : zz;                              // trace.reset() after a terminal transition
                        w8[v] = ww;
                    }
                    IOZ::store8(Zi, off, z8);
                    IO::store8(Wi, off, w8);
                }
                q_n[b] = wave_sum_uniform((acc[0] + acc[1]) + (acc[2] + acc[3]));
            }
            if (!driver) { if (lane == 0 && td_out) td_out[i] = delta; break; }
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            if (c.eps) learner_eps_step(c, term | trunc, pol);                 // the episode's last handle is done: its end decays epsilon
            int na = policy_sample<A>(pol, q_n, x);
            facc_abs += fabsf(delta); facc_r += r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {                                                       // step cap: new episode; the trace is NOT reset
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                Dom::reset(ns);
                WF::project(ns, lane, phi_n);
                WF::template q_from_mem<WT>(Wi, lane, phi_n, q_n);
                const U4 xr = draw(c.seed, gid, t, BLK_RESET);
                na = policy_sample<A>(pol, q_n, xr);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
#pragma unroll
            for (int b = 0; b < A; ++b) q_s[b] = q_n[b];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int v = 0; v < 8; ++v) phi_s[j][v] = phi_n[j][v];
            a = __builtin_amdgcn_readfirstlane(na);
        }
        if (driver && lane == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
            c.action[i] = a;
            c.ep_step[i] = ep;
            if (c.eps) c.eps[i] = pol.eps;
            sum_abs = (double)facc_abs; sum_r = (double)facc_r;
        } else {
            n_ep = 0; n_trunc = 0; sum_len = 0;
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

}  // namespace rsrl
