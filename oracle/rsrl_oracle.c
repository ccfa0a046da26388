/*
 * rsrl_oracle.c -- CPU oracle for the rsrl TD-control hot path.
 *
 * ======================  TEST INFRASTRUCTURE ONLY  ==========================
 * This file restates, in plain C, the algorithm of the reference path
 *   env.transition -> agent.handle -> policy.sample
 * (rsrl/examples/q_learning.rs:34-55) for QLearning / SARSA / ExpectedSARSA
 * over LFA<Fourier|TileCoding> on MountainCar / CartPole / Acrobot.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (rsrl_amd/, librsrl_hip.so) never links, imports or
 * calls anything in oracle/.
 *
 * PINNING STATUS
 *   pinned by the reference's own known-answer tests (tests/test_oracle_golden.py):
 *     - CartPole RK4 dynamics        rsrl_domains/src/cart_pole.rs:143-183
 *     - initial observations         cart_pole.rs:128-141, acrobot.rs:159-174, mountain_car/discrete.rs:109-120
 *     - MountainCar terminal pred.   mountain_car/discrete.rs:122-137
 *     - Greedy argmax/probabilities  rsrl/src/policies/greedy.rs:96-168
 *     - EpsilonGreedy probabilities  rsrl/src/policies/epsilon_greedy.rs:115-145 (+ frequencies :95-113)
 *     - Random frequencies           rsrl/src/policies/random.rs:58-76
 *     - Softmax degenerate cases     rsrl/src/policies/softmax.rs:240-256 (+ documented intent :258-291)
 *   PARITY UNPINNED (third-party arithmetic absent from /root/reference, no
 *   Cargo.lock, no reference test touches it; the reference cannot be built
 *   here -- no cargo/rustc):
 *     - crate lfa = "0.15" (rsrl/Cargo.toml:32): Fourier::project, with_bias,
 *       LFA::vector evaluate / update_index, optim::SGD.  Restated from the
 *       crate's published algorithm.
 *     - lfa TileCoding is hashed with a Rust BuildHasher and is never
 *       instantiated by the reference; the dense grid coder below is this
 *       build's own deterministic definition.
 *     - crate rand = "0.7" streams: replaced by Philox4x32-10 (pinned against
 *       the Random123 known-answer vectors); parity on stochastic paths is
 *       distributional, exact on deterministic sub-paths.
 *   ROUND 5 (rsrl_oracle_impl.h): orc_run_teacher (one batch-step as a teacher:
 *     fp32-rounded successor states + the handled transitions, for the
 *     teacher-forced f64 comparisons of every BASELINE configuration);
 *     orc_run_train_wave also in the wave order for GreedyGQ / TD / TDLambda /
 *     QSigma; orc_run_train_sparse_lambda (traces.rs:5-12 over params/sparse.rs:
 *     eligibility traces over one shared tile table, sparse per learner, checked
 *     against a dense numpy restatement in tests/test_oracle_round5.py).
 * ============================================================================
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rsrl_oracle.h"

/* ------------------------------------------------------------------ */
/* shared (type-independent) pieces                                    */
/* ------------------------------------------------------------------ */

int orc_domain_dim(int domain) { return domain == ORC_MOUNTAIN_CAR ? 2 : 4; }
/* action_space(): Ordinal::new(3|2|3)   discrete.rs:101, cart_pole.rs:120, acrobot.rs:151 */
int orc_domain_actions(int domain) { return domain == ORC_CART_POLE ? 2 : 3; }
/* state_space() bounds   discrete.rs:97-99, cart_pole.rs:112-118, acrobot.rs:143-149 */
void orc_domain_bounds(int domain, double* lo, double* hi) {
    switch (domain) {
    case ORC_MOUNTAIN_CAR:
        lo[0] = -1.2; hi[0] = 0.6; lo[1] = -0.07; hi[1] = 0.07; break;
    case ORC_CART_POLE:
        lo[0] = -2.4; hi[0] = 2.4; lo[1] = -6.0; hi[1] = 6.0;
        lo[2] = -(M_PI / 15.0); hi[2] = M_PI / 15.0; lo[3] = -2.0; hi[3] = 2.0; break;
    default:
        lo[0] = -M_PI; hi[0] = M_PI; lo[1] = -M_PI; hi[1] = M_PI;
        lo[2] = -4.0 * M_PI; hi[2] = 4.0 * M_PI; lo[3] = -9.0 * M_PI; hi[3] = 9.0 * M_PI; break;
    }
}

void orc_basis_init(orc_basis* b, int domain, int kind, int order, int n_tilings, int tiles_per_dim) {
    int i;
    memset(b, 0, sizeof(*b));
    b->kind = kind; b->dim = orc_domain_dim(domain);
    b->order = order; b->n_tilings = n_tilings; b->tiles_per_dim = tiles_per_dim;
    orc_domain_bounds(domain, b->lo, b->hi);
    for (i = 0; i < b->dim; i++) { b->lo_f[i] = (float)b->lo[i]; b->hi_f[i] = (float)b->hi[i]; }
}
int orc_basis_nfeat(const orc_basis* b) {
    int i, n = 1;
    if (b->kind == ORC_FOURIER) { for (i = 0; i < b->dim; i++) n *= (b->order + 1); return n; }
    for (i = 0; i < b->dim; i++) n *= b->tiles_per_dim;
    return n * b->n_tilings;
}
static const double* basis_lo_f64(const orc_basis* b) { return b->lo; }
static const double* basis_hi_f64(const orc_basis* b) { return b->hi; }
static const float*  basis_lo_f32(const orc_basis* b) { return b->lo_f; }
static const float*  basis_hi_f32(const orc_basis* b) { return b->hi_f; }

/* Dense grid tile coder (this build's definition; the
 * reference never instantiates lfa::TileCoding).  ALL arithmetic is fp32 with
 * non-fused ops (build with -ffp-contract=off) so the device reproduces the
 * indices bit-exactly:
 *   s~_i   = (s_i - lo_i) / (hi_i - lo_i)
 *   u_i    = s~_i * (B - 1)
 *   off_i  = ((t*(2i+1)) mod T) / T
 *   cell_i = min(B-1, max(0, (int)floorf(u_i + off_i)))
 *   idx(t) = t*B^D + sum_i cell_i * B^i                                  */
void orc_tile_indices(const orc_basis* b, const float* s, int* idx) {
    int B = b->tiles_per_dim, T = b->n_tilings, D = b->dim, t, i, BD = 1;
    for (i = 0; i < D; i++) BD *= B;
    for (t = 0; t < T; t++) {
        int lin = 0, stride = 1;
        for (i = 0; i < D; i++) {
            volatile float num = s[i] - b->lo_f[i];
            volatile float den = b->hi_f[i] - b->lo_f[i];
            volatile float sc = num / den;
            volatile float u = sc * (float)(B - 1);
            volatile float off = (float)((t * (2 * i + 1)) % T) / (float)T;
            volatile float v = u + off;
            int cell = (int)floorf(v);
            if (cell < 0) cell = 0;
            if (cell > B - 1) cell = B - 1;
            lin += cell * stride; stride *= B;
        }
        idx[t] = t * BD + lin;
    }
}

/* Philox4x32-10 (Salmon et al., SC'11; Random123).  Counter-based, shared
 * bit-for-bit with the device code (rsrl_amd/csrc/philox.hpp). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    int r;
    for (r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* Draw for (seed, global env id, batch-step t, block):  key = (seed_lo, seed_hi); counter = (t_lo, t_hi, env_id, block).
 * The per-step draws (ORC_BLK_STEP with its alias ORC_BLK_RESET, ORC_BLK_INNER) use two words -- out[0] = explore?,
 * out[1] = out[2] = the uniform pick -- and two consecutive batch-steps share one Philox block addressed by t >> 1 (even
 * step: words 0, 1; odd step: words 2, 3), as on the device (rsrl_amd/csrc/device_core.hpp draw()).  Every other block index
 * takes the whole 128-bit block at counter t. */
void orc_draw(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t block, uint32_t out[4]) {
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    if (block <= ORC_BLK_INNER) {
        const uint64_t th = t >> 1;
        uint32_t ctr[4] = { (uint32_t)th, (uint32_t)(th >> 32), (uint32_t)env_id, block == ORC_BLK_INNER ? ORC_BLK_INNER : ORC_BLK_STEP };
        uint32_t p[4];
        orc_philox4x32_10(ctr, key, p);
        out[0] = (t & 1u) ? p[2] : p[0];
        out[1] = out[2] = (t & 1u) ? p[3] : p[1];
        out[3] = 0u;
    } else {
        uint32_t ctr[4] = { (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)env_id, block };
        orc_philox4x32_10(ctr, key, out);
    }
}
uint32_t orc_mulhi(uint32_t x, uint32_t n) { return (uint32_t)(((uint64_t)x * n) >> 32); }
uint32_t orc_eps_threshold(double eps) {
    double v = eps * 16777216.0;
    if (v <= 0.0) return 0u;
    if (v >= 16777216.0) return 16777216u;
    return (uint32_t)v;
}

void orc_agent_init(orc_agent* ag, int domain, int basis_kind, int order, int n_tilings, int tiles_per_dim,
                    int algo, int policy, int shared_w, uint64_t seed, int64_t env_offset,
                    double gamma, double lr, double alpha, double epsilon, double tau,
                    uint32_t max_episode_steps) {
    memset(ag, 0, sizeof(*ag));
    ag->domain = domain; ag->algo = algo; ag->policy = policy; ag->shared_w = shared_w;
    ag->n_actions = orc_domain_actions(domain);
    orc_basis_init(&ag->basis, domain, basis_kind, order, n_tilings, tiles_per_dim);
    ag->seed = seed; ag->env_offset = env_offset;
    ag->gamma = gamma; ag->lr = lr; ag->alpha = alpha; ag->epsilon = epsilon; ag->tau = tau;
    ag->eps_thr = orc_eps_threshold(epsilon);
    ag->max_episode_steps = max_episode_steps;
    ag->lambda = 0.0; ag->trace = ORC_TRACE_ACCUMULATE; ag->lr_td = 0.0;
    ag->apolicy = policy; ag->aepsilon = epsilon; ag->atau = tau; ag->aeps_thr = ag->eps_thr;
    ag->sigma = 0.0; ag->n_steps = 1;
    ag->apol_same = 1; ag->eps_decay = 1.0; ag->eps_min = 0.0;
}

/* The trace's decay Trace::new(gamma * lambda) (sarsa_lambda.rs:47-52; Dutch: x (1 - alpha), traces.rs:233-239): the agents' parameters are f64 in
 * the reference's API, so the product is taken in f64 and rounded ONCE to the instantiation's type -- the same value the device's host side
 * hands its kernels (rsrl_hip.hip make_lambda / make_td).  (Until round 5 the f32 instantiations multiplied the two ROUNDED factors; for most pairs
 * that is the same float, for (0.9, 0.9) it is not: found by tests/fuzz_parity.py.) */
static double orc_trace_rate(const orc_agent* ag) {
    double rate = ag->gamma * ag->lambda;
    if (ag->trace == ORC_TRACE_DUTCH) rate *= (1.0 - ag->alpha);
    return rate;
}

/* ------------------------------------------------------------------ */
/* instantiate the type-generic body: f64 (reference precision), f32   */
/* ------------------------------------------------------------------ */
#define R double
#define FN(name) name##_f64
#define RC(name) name
#define RMAX DBL_MAX
#include "rsrl_oracle_impl.h"
#undef R
#undef FN
#undef RC
#undef RMAX

#define R float
#define FN(name) name##_f32
#define RC(name) name##f
#define RMAX FLT_MAX
#define ORC_SEPARABLE 1
#include "rsrl_oracle_impl.h"
#undef ORC_SEPARABLE
#undef R
#undef FN
#undef RC
#undef RMAX

/* f32 again with the device's own sincos / exp polynomials restated (bitwise comparison with the HIP path) */
static const float*  basis_lo_f32d(const orc_basis* b) { return b->lo_f; }
static const float*  basis_hi_f32d(const orc_basis* b) { return b->hi_f; }
#define R float
#define FN(name) name##_f32d
#define RC(name) name##f
#define RMAX FLT_MAX
#define ORC_SEPARABLE 1
#define ORC_DEVTRIG 1
#include "rsrl_oracle_impl.h"
#undef ORC_DEVTRIG
#undef ORC_SEPARABLE
#undef R
#undef FN
#undef RC
#undef RMAX
