// kernels_wave_aux.hpp -- the agents with a SECOND per-learner matrix, and the prediction agents, on the WAVE family (Fourier order 7 on a 4-D
// state space, F = 4096, one wavefront per learner; round 5 -- the reference's agents are generic over the approximator; weights f32, or (round 6)
// bf16 with stochastic rounding -- the SECOND matrix (fa_td's weights / the trace) stays f32):
//   GreedyGQ::handle   rsrl/src/control/td/greedy_gq.rs:73-141      fa_q = W, fa_td = V (the ctx's auxiliary matrix), both f32[N][A][F]
//   TD::handle         rsrl/src/prediction/td/td.rs:31-59           one weight column w, f32[N][1][F]
//   TDLambda::handle   rsrl/src/prediction/td/td_lambda.rs:41-78    + the trace z (auxiliary matrix, f32[N][1][F]); rules traces.rs:188-240
//
// Like the eligibility-trace agents of this family (kernels_wave_lambda.hpp) these cannot keep their matrices on the chip -- W + V are 96 KiB per
// learner -- so a learner-step is a memory sweep, 16 B per lane, coalesced, in the family's layout (internal index k = j*512 + lane*8 + v):
//   GreedyGQ   Q(s',.) with the pre-update W and td_est = <phi(s), V[:,a]>  (W once, one column of V), then ONE sweep over the columns of W that
//              move (a: += lr*delta*phi(s); argmax of Q(s',.), non-terminal: += lr*(-gamma*td_est)*phi(s'), in that order when they coincide) and
//              over V[:,a] (+= lr_td*(delta - td_est)*phi(s)); Q(s',.) with the UPDATED W -- what the behaviour policy samples from and the next
//              step's Q(s,.) -- falls out of the sweep.  Columns that do not move are read once more for that dot product only.
//   TD         V(s') with the pre-update w, w += lr*td*phi(s), V(s') with the updated w out of the same sweep.
//   TDLambda   z = rule(rate*z + phi(s)) (rate 0 after a terminal transition: the trace was reset), w += td * z, V(s') with the updated w.
// Element by element the operations are those of the register-family kernels (kernels_gq.hpp, kernels_td.hpp); every dot product runs over (j, v)
// in WaveFourier::dot()'s order: bit-identical to the oracle's wave-order loop (orc_run_train_wave).
// bf16 (WT = bf16_t): W is read as bf16, arithmetic f32; every entry of W that is STORED is rounded once, after all of the step's updates of it, by
// stochastic rounding with the lane's Philox block 16 + 64 * b + lane of the step (column b; window of element e = j * 8 + v as in k_train_wave),
// and the dot products of the sweep run over the rounded values -- what a fresh evaluation would read.
#pragma once

#include "kernels_wave.hpp"
#include "kernels_gq.hpp"
#include "kernels_td.hpp"
#include "kernels_qsigma.hpp"

namespace rsrl {

enum : int { WAUX_GQ = 0, WAUX_TD = 1, WAUX_TDL = 2 };

struct WaveAuxParams {
    int mode;          // WAUX_*
    float* aux;        // GreedyGQ: V [N][A][F]; TDLambda: z [N][1][F]; TD: unused
    float lr_td;       // GreedyGQ: SGD rate of fa_td
    float rate;        // TDLambda: gamma*lambda (Dutch: * (1 - alpha))
    int trace;         // TDLambda: TRACE_*
};

// <phi, column> with the column streamed from memory, dot()'s order (4 interleaved chains over (j, v), then the wave total)
template <int DOMAIN, class WT = float>
__device__ __forceinline__ float wave_col_dot(const WT* __restrict__ col, int lane, const float (&phi)[8][8]) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float w8[8];
        WaveIO<WT>::load8(col, (int64_t)j * 512 + lane * 8, w8);
#pragma unroll
        for (int v = 0; v < 8; ++v) acc[v & 3] = fmaf(phi[j][v], w8[v], acc[v & 3]);
    }
    return wave_sum_uniform((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

// from == nullptr: the driver loop, n_steps batch-steps of the wave's learner.  Otherwise Handler::handle on ONE caller-supplied transition per
// learner (Mn of them; the prediction agents ignore `act`).
template <int DOMAIN, class WT = float>
__global__ __launch_bounds__(kBlock) void k_wave_aux(Common c, WaveAuxParams ap, WT* __restrict__ Wbase, uint64_t t0, int n_steps,
                                                     DevStats* __restrict__ stats, const float* __restrict__ from, const int32_t* __restrict__ act,
                                                     const float* __restrict__ rew, const float* __restrict__ to, const uint8_t* __restrict__ termf,
                                                     int64_t Mn, float* __restrict__ td_out) {
    using WF = WaveFourier<DOMAIN>;
    using Dom = Domain<DOMAIN>;
    using IO = WaveIO<WT>;
    using IOX = WaveIO<float>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t N = c.n_envs;
    const bool driver = from == nullptr;
    const bool gq = ap.mode == WAUX_GQ;
    const int Aw = gq ? A : 1;                                                         // columns of the weight matrix
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);      // learner of this wave (uniform)
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < (driver ? N : Mn)) {
        PolicyParams pol = c.pol;
        if (!gq) pol.kind = POL_RANDOM;                                                // prediction: the only policy that needs no Q
        const float gamma = c.alg.gamma, lr = c.alg.lr;
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        WT* __restrict__ Wi = Wbase + i * (int64_t)Aw * F;
        float* __restrict__ Xi = ap.aux ? ap.aux + i * (int64_t)Aw * F : nullptr;
        float s[D];
        int a = 0; uint32_t ep = 0;
        if (driver) {
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = c.state[(int64_t)d * N + i];
            a = __builtin_amdgcn_readfirstlane(c.action[i]); ep = c.ep_step[i];
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = from[(int64_t)d * Mn + i];
            if (gq) a = clamp_action<A>(__builtin_amdgcn_readfirstlane(act[i]));
        }
        float phi_s[8][8], q_s[A];
#pragma unroll
        for (int b = 0; b < A; ++b) q_s[b] = 0.0f;
        WF::project(s, lane, phi_s);
#pragma unroll
        for (int b = 0; b < A; ++b) if (b < Aw) q_s[b] = wave_col_dot<DOMAIN, WT>(Wi + (int64_t)b * F, lane, phi_s);       // GQ: Q(s,.); prediction: q_s[0] = V(s)
        float facc_abs = 0.0f, facc_r = 0.0f;
        bool cut = false;                                                              // TDLambda: the previous transition was terminal
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
#pragma unroll
            for (int b = 0; b < A; ++b) q_n[b] = 0.0f;
            WF::project(ns, lane, phi_n);
#pragma unroll
            for (int b = 0; b < A; ++b) if (b < Aw) q_n[b] = wave_col_dot<DOMAIN, WT>(Wi + (int64_t)b * F, lane, phi_n);   // PRE-update weights
            float delta;
            if (gq) {
                const float qsa = select_a<A>(q_s, a);
                const float td_est = wave_col_dot<DOMAIN>(Xi + (int64_t)a * F, lane, phi_s);
                float qmax;
                const int na_star = __builtin_amdgcn_readfirstlane(find_max<A>(q_n, qmax));
                delta = term ? (r - qsa) : (r + gamma * qmax - qsa);
                const float sc1 = lr * delta, sc2 = lr * (-gamma * td_est), sc3 = ap.lr_td * (delta - td_est);
#pragma unroll
                for (int b = 0; b < A; ++b) {
                    const bool hit1 = a == b, hit2 = !term && na_star == b;            // wave-uniform
                    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    U4 rnd = U4{0, 0, 0, 0};
                    if constexpr (IO::kBf16) { if (hit1 || hit2) rnd = draw(c.seed, gid, t, BLK_SR_BASE + 64u * (uint32_t)b + (uint32_t)lane); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int64_t off = (int64_t)b * F + j * 512 + lane * 8;
                        float w8[8];
                        IO::load8(Wi, off, w8);
                        if (hit1) {
#pragma unroll
                            for (int v = 0; v < 8; ++v) w8[v] = fmaf(sc1, phi_s[j][v], w8[v]);
                            float v8[8];
                            IOX::load8(Xi, off, v8);
#pragma unroll
                            for (int v = 0; v < 8; ++v) v8[v] = fmaf(sc3, phi_s[j][v], v8[v]);
                            IOX::store8(Xi, off, v8);
                        }
                        if (hit2) {
#pragma unroll
                            for (int v = 0; v < 8; ++v) w8[v] = fmaf(sc2, phi_n[j][v], w8[v]);
                        }
                        if (hit1 || hit2) {
                            if constexpr (IO::kBf16) {
#pragma unroll
                                for (int v = 0; v < 8; ++v) w8[v] = round_bf16_sr(w8[v], sr_bits(rnd, j * 8 + v));
                            }
                            IO::store8(Wi, off, w8);
                        }
#pragma unroll
                        for (int v = 0; v < 8; ++v) acc[v & 3] = fmaf(phi_n[j][v], w8[v], acc[v & 3]);
                    }
                    q_n[b] = wave_sum_uniform((acc[0] + acc[1]) + (acc[2] + acc[3]));
                }
            } else {
                delta = term ? (r - q_s[0]) : (r + gamma * q_n[0] - q_s[0]);
                const bool lam = ap.mode == WAUX_TDL;
                const float rate_eff = cut ? 0.0f : ap.rate;
                const float sc = lr * delta;
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                U4 rnd = U4{0, 0, 0, 0};
                if constexpr (IO::kBf16) rnd = draw(c.seed, gid, t, BLK_SR_BASE + (uint32_t)lane);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t off = (int64_t)j * 512 + lane * 8;
                    float w8[8], z8[8];
                    IO::load8(Wi, off, w8);
                    if (lam) {
                        IOX::load8(Xi, off, z8);
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            float zz = fmaf(rate_eff, z8[v], 1.0f * phi_s[j][v]);      // WBuf::decay_add with the indicator 1
                            if (ap.trace == TRACE_SATURATE) zz = __builtin_amdgcn_fmed3f(zz, -1.0f, 1.0f);
                            w8[v] = fmaf(delta, zz, w8[v]);                            // ScaledGradientUpdate{alpha: td_error}: no learning rate
                            z8[v] = term ? 0.0f : zz;                                  // trace.reset() after a terminal transition
                        }
                        IOX::store8(Xi, off, z8);
                    } else {
#pragma unroll
                        for (int v = 0; v < 8; ++v) w8[v] = fmaf(sc, phi_s[j][v], w8[v]);
                    }
                    if constexpr (IO::kBf16) {
#pragma unroll
                        for (int v = 0; v < 8; ++v) w8[v] = round_bf16_sr(w8[v], sr_bits(rnd, j * 8 + v));
                    }
                    IO::store8(Wi, off, w8);
#pragma unroll
                    for (int v = 0; v < 8; ++v) acc[v & 3] = fmaf(phi_n[j][v], w8[v], acc[v & 3]);
                }
                q_n[0] = wave_sum_uniform((acc[0] + acc[1]) + (acc[2] + acc[3]));
                cut = term;
            }
            if (!driver) { if (lane == 0 && td_out) td_out[i] = delta; break; }
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            int na = policy_sample<A>(pol, q_n, x);
            facc_abs += fabsf(delta); facc_r += r;
            if (term) { n_ep += 1; sum_len += ep; ep = 0; }
            if (trunc) {                                                               // step cap: new episode (a trace is NOT reset)
                n_ep += 1; n_trunc += 1; sum_len += ep; ep = 0;
                Dom::reset(ns);
                WF::project(ns, lane, phi_n);
#pragma unroll
                for (int b = 0; b < A; ++b) if (b < Aw) q_n[b] = wave_col_dot<DOMAIN, WT>(Wi + (int64_t)b * F, lane, phi_n);
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
            sum_abs = (double)facc_abs; sum_r = (double)facc_r;
        } else {
            n_ep = 0; n_trunc = 0; sum_len = 0;
        }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// ---------------------------------------------------------------------------------------
// QSigma on the wave family (q_sigma.rs:80-202; kernels_qsigma.hpp has the register / memory families' form and the one repaired line of the reference)
// ---------------------------------------------------------------------------------------
// The n-step backup, its propagation and every decision are wave-uniform scalar work (all lanes alike, the ring in memory as on the other families: lane 0 writes
// it); what the 64 lanes share is the approximator: features = the lane's 64 of the 4096, Q = lane partials + the wave total in dot()'s order, the anchor's column
// update = one coalesced sweep of that column.  Operation by operation qsigma_handle<M> (kernels_qsigma.hpp), with what the loop knows used instead of re-read
// (round 6): Q(s,a) is the carried Q(s',.) of the previous step (same state, same weights, the same dot product); Q(s',.) after the anchor's update differs from the
// pre-update evaluation in the anchor's column only; and that column -- read ONCE, held in 64 registers per lane -- gives the anchor's Q and, after the update, its
// dot product with phi(s') over the stored values: per learner-step W is read 4 F and written F instead of 9 F and F (146 -> 123.5 us per batch-step at 8 192
// learners).  bf16 (WT = bf16_t, round 6): the column's entries are rounded stochastically before they are stored (Philox block 16 + 64 * column + lane of the step,
// window of element e = j * 8 + v), the dot product with phi(s') runs over the rounded values.
// from == nullptr: the driver loop, n_steps batch-steps of the wave's learner; otherwise Handler::handle on ONE caller-supplied transition per learner (Mn of them).
template <int DOMAIN, class WT = float>
__global__ __launch_bounds__(kBlock) void k_wave_qsigma(Common c, QsParams qp, WT* __restrict__ Wbase, uint64_t t0, int n_steps, DevStats* __restrict__ stats,
                                                        const float* __restrict__ from, const int32_t* __restrict__ act, const float* __restrict__ rew,
                                                        const float* __restrict__ to, const uint8_t* __restrict__ termf, int64_t Mn, float* __restrict__ td_out) {
    using WF = WaveFourier<DOMAIN>;
    using Dom = Domain<DOMAIN>;
    using IO = WaveIO<WT>;
    constexpr int D = WF::D, A = WF::A, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t N = c.n_envs;
    const bool driver = from == nullptr;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);      // learner of this wave (uniform)
    unsigned long long n_ep = 0, n_trunc = 0, sum_len = 0;
    double sum_abs = 0.0, sum_r = 0.0;
    if (i < (driver ? N : Mn)) {
        const uint32_t gid = (uint32_t)(c.env_offset + i);
        const uint32_t cap = c.max_episode_steps;
        WT* __restrict__ Wi = Wbase + i * (int64_t)(A * F);
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
        float facc_abs = 0.0f, facc_r = 0.0f;
        float phi_s[8][8], q_s[A];
        WF::project(s, lane, phi_s);
#pragma unroll
        for (int b = 0; b < A; ++b) q_s[b] = wave_col_dot<DOMAIN, WT>(Wi + (int64_t)b * F, lane, phi_s);
        const int n = qp.n_steps;
        uint32_t head = qp.head[i], len = qp.len[i];
        const int64_t fs_stride = (int64_t)n * N;
        auto at = [&](int field, uint32_t slot) -> float& { return qp.buf[(int64_t)field * fs_stride + (int64_t)slot * N + i]; };
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
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) ns[d] = to[(int64_t)d * Mn + i];
                r = rew[i]; term = termf[i] != 0;
            }
            const U4 xin = draw(c.seed, gid, t, BLK_INNER);
            // ---- QSigma::handle (q_sigma.rs:138-201)
            const float qa = select_a<A>(q_s, a);                                      // :140
            float phi_n[8][8], q_n[A];
#pragma unroll
            for (int b = 0; b < A; ++b) q_n[b] = 0.0f;
            WF::project(ns, lane, phi_n);                                             // (a terminal s' too: the anchor's sweep multiplies by it, the result is dropped)
            float residual, pi, mu;
            if (term) {
                residual = r - qa; pi = 0.0f; mu = 1.0f;                              // :142-153
            } else {
#pragma unroll
                for (int b = 0; b < A; ++b) q_n[b] = wave_col_dot<DOMAIN, WT>(Wi + (int64_t)b * F, lane, phi_n);
                const int na = policy_sample<A>(c.apol, q_n, xin);                   // :157 the agent's own draw
                const float nqsna = select_a<A>(q_n, na);
                float exp_nqs;
                const uint32_t mask = argmaxima_mask_max<A>(q_n, exp_nqs);            // :161
                pi = ((mask >> na) & 1u) ? 1.0f / (float)__popc(mask) : 0.0f;         // :163-167
                mu = policy_eval_sa<A>(c.apol, q_n, na);                              // :168
                residual = r + c.alg.gamma * (qp.sigma * nqsna + (1.0f - qp.sigma) * exp_nqs) - qa;    // :170-171
            }
            // ---- update_backup (:107-128): push, then one update of the anchor once n_steps entries are held
            {
                const uint32_t slot = (head + len) % (uint32_t)n;
                if (lane == 0) {
#pragma unroll
                    for (int d = 0; d < D; ++d) at(d, slot) = s[d];
                    at(D, slot) = __int_as_float(a);
                    at(D + 1, slot) = qa; at(D + 2, slot) = residual; at(D + 3, slot) = pi; at(D + 4, slot) = mu;
                }
                len += 1;
            }
            const bool restart = driver && (term || trunc);
            if ((int)len >= n) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                // lane 0's ring stores above are read back by every lane below
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // Backup::propagate (:46-63) with the dead out-of-bounds z update of the last iteration dropped (kernels_qsigma.hpp header)
                float gret = at(D + 1, head), z = 1.0f, isr = 1.0f;
                for (int kk = 0; kk < n; ++kk) {
                    const uint32_t s1 = (head + (uint32_t)kk) % (uint32_t)n;
                    gret += z * at(D + 2, s1);
                    if (kk + 1 < n) {
                        const uint32_t s2 = (head + (uint32_t)kk + 1u) % (uint32_t)n;
                        z *= c.alg.gamma * ((1.0f - qp.sigma) * at(D + 3, s2) + qp.sigma);
                    }
                    isr *= 1.0f - qp.sigma + qp.sigma * at(D + 3, s1) / at(D + 4, s1);
                }
                float as_[D];
#pragma unroll
                for (int d = 0; d < D; ++d) as_[d] = at(d, head);
                const int aa = __builtin_amdgcn_readfirstlane(clamp_action<A>(__float_as_int(at(D, head))));
                head = (head + 1u) % (uint32_t)n; len -= 1;                           // pop (:116)
                float phi_a[8][8];
                WF::project(as_, lane, phi_a);
                WT* __restrict__ col = Wi + (int64_t)aa * F;
                float wc[8][8];
                float qacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    IO::load8(col, (int64_t)j * 512 + lane * 8, wc[j]);
#pragma unroll
                    for (int v = 0; v < 8; ++v) qacc[v & 3] = fmaf(phi_a[j][v], wc[j][v], qacc[v & 3]);
                }
                const float qsa = wave_sum_uniform((qacc[0] + qacc[1]) + (qacc[2] + qacc[3]));      // :117 with the CURRENT weights
                const float scale = c.alg.lr * (qp.alpha * isr * (gret - qsa));                      // :122; Handler<StateActionUpdate>: W[:,a] += lr*error*phi
                U4 rnd = U4{0, 0, 0, 0};
                if constexpr (IO::kBf16) rnd = draw(c.seed, gid, t, BLK_SR_BASE + 64u * (uint32_t)aa + (uint32_t)lane);
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        float x = fmaf(scale, phi_a[j][v], wc[j][v]);
                        if constexpr (IO::kBf16) x = round_bf16_sr(x, sr_bits(rnd, j * 8 + v));
                        wc[j][v] = x;
                        acc[v & 3] = fmaf(phi_n[j][v], x, acc[v & 3]);
                    }
                    IO::store8(col, (int64_t)j * 512 + lane * 8, wc[j]);
                }
                if (!restart && !term) {
                    const float qpost = wave_sum_uniform((acc[0] + acc[1]) + (acc[2] + acc[3]));
#pragma unroll
                    for (int b = 0; b < A; ++b) q_n[b] = (b == aa) ? qpost : q_n[b];
                }
            }
            if (term) len = 0;                                                        // backup.clear() (:154)
            if (!driver) { if (lane == 0 && td_out) td_out[i] = residual; break; }
            if (restart) {
                n_ep += 1; n_trunc += trunc ? 1 : 0; sum_len += ep; ep = 0;
                Dom::reset(ns);
                WF::project(ns, lane, phi_n);
#pragma unroll
                for (int b = 0; b < A; ++b) q_n[b] = wave_col_dot<DOMAIN, WT>(Wi + (int64_t)b * F, lane, phi_n);      // the UPDATED weights at s0
            }
            const U4 x = draw(c.seed, gid, t, BLK_STEP);
            a = __builtin_amdgcn_readfirstlane(policy_sample<A>(c.pol, q_n, x));
            facc_abs += fabsf(residual); facc_r += r;
#pragma unroll
            for (int d = 0; d < D; ++d) s[d] = ns[d];
#pragma unroll
            for (int b = 0; b < A; ++b) q_s[b] = q_n[b];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int v = 0; v < 8; ++v) phi_s[j][v] = phi_n[j][v];
        }
        if (lane == 0) {
            qp.head[i] = head; qp.len[i] = len;
            if (driver) {
#pragma unroll
                for (int d = 0; d < D; ++d) c.state[(int64_t)d * N + i] = s[d];
                c.action[i] = a;
                c.ep_step[i] = ep;
                sum_abs = (double)facc_abs; sum_r = (double)facc_r;
            }
        }
        if (lane != 0 || !driver) { n_ep = 0; n_trunc = 0; sum_len = 0; }
    }
    if (stats) block_stats_accumulate(stats, n_ep, n_trunc, sum_len, sum_abs, sum_r);
}

// V(s) of a prediction agent for M caller-supplied states (Function<(S,)> of the ScalarLFA), one wave per state
template <int DOMAIN, class WT = float>
__global__ __launch_bounds__(kBlock) void k_wave_v_evaluate(const WT* __restrict__ Wbase, const float* __restrict__ states, int64_t Mn, float* __restrict__ out) {
    using WF = WaveFourier<DOMAIN>;
    constexpr int D = WF::D, F = WF::F;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (i >= Mn) return;
    float s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = states[(int64_t)d * Mn + i];
    float phi[8][8];
    WF::project(s, lane, phi);
    const float v = wave_col_dot<DOMAIN, WT>(Wbase + i * (int64_t)F, lane, phi);
    if (lane == 0) out[i] = v;
}

}  // namespace rsrl
