// rsrl.hpp -- C++ host-side mirror of the rsrl trait surface for the TD-control hot path, on top of the
// C ABI (include/rsrl_hip.h).  The reference is Rust and this image has no Rust toolchain, so the host
// side above the ABI is C++ with the reference's names, argument meaning and error behaviour:
//
//   rsrl::domains::{MountainCar, CartPole, Acrobot}      rsrl_domains/src/{mountain_car/discrete,cart_pole,acrobot}.rs
//   rsrl::fa::linear::{basis::Fourier, optim::SGD, LFA}  rsrl/src/fa/linear.rs (re-exports of crate lfa)
//   rsrl::policies::{Greedy, EpsilonGreedy, Softmax, Random}   rsrl/src/policies/*.rs
//   rsrl::control::td::{QLearning, SARSA, ExpectedSARSA, PAL, SARSALambda, QLambda, GreedyGQ, QSigma}   rsrl/src/control/td/*.rs
//   rsrl::make_shared / Shared<T>                              rsrl/src/core.rs:13-44
//
// Everything is BATCHED: a Domain is N environments, a state is a column of a [D][N] array.  The objects
// are light descriptors until `bind()` creates the device context (one per (domain, q_func, agent, policy)
// composition, exactly the object graph of examples/q_learning.rs:19-32).  Errors: the reference panics /
// returns Err; here every failure throws rsrl::Error carrying the ABI status and message.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rsrl_hip.h"

namespace rsrl {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != RSRL_HIP_OK) throw Error(rc, rsrl_hip_last_error());
}

// core.rs:13-44 -- Shared<T> is Rc<RefCell<T>>; clones alias the same object
template <class T> using Shared = std::shared_ptr<T>;
template <class T> Shared<T> make_shared(T t) { return std::make_shared<T>(std::move(t)); }

// ---- rsrl_domains ---------------------------------------------------------------------------------
namespace domains {
struct Transition {                       // lib.rs:130-142, batched SoA
    std::vector<float> from, to;          // [D][N]
    std::vector<int32_t> action;          // [N]
    std::vector<float> reward;            // [N]
    std::vector<uint8_t> terminal;        // [N]  (Observation::Terminal flag of `to`, lib.rs:170)
};
struct Trajectory {                       // lib.rs:334-409, batched: `start` = states row 0, steps[k] = (states row k+1, actions[k], rewards[k])
    std::vector<uint32_t> n_states;       // [N]  Trajectory::n_states()  (:340); n_transitions = n_states - 1
    std::vector<float> total_reward;      // [N]  Trajectory::total_reward()  (:391)
    std::vector<float> states;            // [step_limit][D][N], rows past n_states are zero
    std::vector<int32_t> actions;         // [step_limit - 1][N]
    std::vector<float> rewards;           // [step_limit - 1][N]
    std::vector<uint8_t> terminal;        // [N]  the last observation is Observation::Terminal
};
struct Domain { int kind; int64_t n_envs; };
struct MountainCar : Domain { explicit MountainCar(int64_t n = 1) : Domain{RSRL_MOUNTAIN_CAR, n} {} };
struct CartPole : Domain { explicit CartPole(int64_t n = 1) : Domain{RSRL_CART_POLE, n} {} };
struct Acrobot : Domain { explicit Acrobot(int64_t n = 1) : Domain{RSRL_ACROBOT, n} {} };
}  // namespace domains

// ---- rsrl::fa::linear ------------------------------------------------------------------------------
namespace fa { namespace linear {
namespace basis {
struct Fourier {                          // Fourier::from_space(order, space).with_bias()  (q_learning.rs:24)
    int order;
    bool bias = false;
    static Fourier from_space(int order, const domains::Domain&) { return Fourier{order}; }
    Fourier with_bias() const { Fourier f = *this; f.bias = true; return f; }
};
struct TileCoding { int n_tilings, tiles_per_dim; };
}  // namespace basis
namespace optim { struct SGD { double lr; explicit SGD(double l) : lr(l) {} }; }
struct LFA {                              // LFA::vector(basis, SGD(lr), n_actions)  (q_learning.rs:25)
    int basis_kind, order, n_tilings, tiles_per_dim;
    double lr;
    bool shared_weights = false;          // one approximator for all envs (synchronous mini-batch rule)
    static LFA vector(const basis::Fourier& b, optim::SGD o, int /*n_actions*/) {
        if (!b.bias) throw Error(RSRL_HIP_EINVAL, "the device basis always carries the constant feature: call with_bias()");
        return LFA{RSRL_FOURIER, b.order, 0, 0, o.lr};
    }
    static LFA vector(const basis::TileCoding& b, optim::SGD o, int /*n_actions*/) {
        return LFA{RSRL_TILE_CODING, 0, b.n_tilings, b.tiles_per_dim, o.lr};
    }
};
}}  // namespace fa::linear

// ---- rsrl::policies --------------------------------------------------------------------------------
namespace policies {
struct Policy {
    int kind; double epsilon = 0.0, tau = 1.0;
    // the reference drivers' schedule on the pub field: `agent.policy.epsilon *= decay` after every EPISODE of a learner
    // (examples/sarsa_lambda.rs:68); with N learners in one ctx every learner carries its own field.  1.0 = no schedule.
    double epsilon_decay = 1.0, epsilon_min = 0.0;
};
struct Random : Policy { explicit Random(int /*n_actions*/) : Policy{RSRL_RANDOM} {} };
struct Greedy : Policy {
    Shared<fa::linear::LFA> q;
    explicit Greedy(Shared<fa::linear::LFA> q_func) : Policy{RSRL_GREEDY}, q(std::move(q_func)) {}
};
struct EpsilonGreedy : Policy {           // EpsilonGreedy::new(greedy, random, epsilon)  (epsilon_greedy.rs:22-31)
    Shared<fa::linear::LFA> q;
    EpsilonGreedy(const Greedy& g, const Random&, double eps) : Policy{RSRL_EPSILON_GREEDY, eps}, q(g.q) {}
    // the driver's per-episode `policy.epsilon *= decay` (floored at `floor`) as part of the policy object
    EpsilonGreedy& decayed_per_episode(double decay, double floor = 0.0) { epsilon_decay = decay; epsilon_min = floor; return *this; }
};
struct Softmax : Policy {                 // Softmax::new(fa, tau) panics for |tau| < 1e-7 (softmax.rs:63-66)
    Shared<fa::linear::LFA> q;
    Softmax(Shared<fa::linear::LFA> q_func, double tau_) : Policy{RSRL_SOFTMAX, 0.0, tau_}, q(std::move(q_func)) {
        if (tau_ < 1e-7 && tau_ > -1e-7) throw Error(RSRL_HIP_EINVAL, "Tau parameter in Softmax must be non-zero.");
    }
};
}  // namespace policies

// ---- rsrl::control::td -----------------------------------------------------------------------------
namespace control { namespace td {
struct Agent {
    int algo; Shared<fa::linear::LFA> q_func; double gamma; double alpha = 1.0;
    int trace = RSRL_TRACE_ACCUMULATE; double lambda = 0.0;      // SARSALambda / QLambda
    double lr_td = 0.0;                                          // GreedyGQ: SGD rate of fa_td
    // the policy the AGENT owns (SARSA / ExpectedSARSA / SARSALambda / QSigma: pub field `policy`).  Unset: the behaviour
    // policy object itself, as in the reference's examples, which share one policy through make_shared.
    bool owns_policy = false; policies::Policy policy{RSRL_GREEDY};
    double sigma = 0.0; int n_steps = 1;                         // QSigma
    Agent& with_policy(const policies::Policy& p) { owns_policy = true; policy = p; return *this; }
};
struct QLearning : Agent { QLearning(Shared<fa::linear::LFA> q, double gamma) : Agent{RSRL_QLEARNING, std::move(q), gamma} {} };
// SARSA { q_func, policy, gamma }                                                  (control/td/sarsa.rs:35-41)
struct SARSA : Agent {
    SARSA(Shared<fa::linear::LFA> q, double gamma) : Agent{RSRL_SARSA, std::move(q), gamma} {}
    SARSA(Shared<fa::linear::LFA> q, const policies::Policy& p, double gamma) : Agent{RSRL_SARSA, std::move(q), gamma} { with_policy(p); }
};
// ExpectedSARSA { q_func, policy, alpha, gamma }                                   (control/td/expected_sarsa.rs:22-29)
struct ExpectedSARSA : Agent {
    ExpectedSARSA(Shared<fa::linear::LFA> q, double alpha_, double gamma) : Agent{RSRL_EXPECTED_SARSA, std::move(q), gamma, alpha_} {}
    ExpectedSARSA(Shared<fa::linear::LFA> q, const policies::Policy& p, double alpha_, double gamma)
        : Agent{RSRL_EXPECTED_SARSA, std::move(q), gamma, alpha_} { with_policy(p); }
};
// QSigma::new(q_func, policy, alpha, gamma, sigma, n_steps)                         (control/td/q_sigma.rs:94-105)
struct QSigma : Agent {
    QSigma(Shared<fa::linear::LFA> q, const policies::Policy& p, double alpha_, double gamma, double sigma_, int n_steps_)
        : Agent{RSRL_Q_SIGMA, std::move(q), gamma, alpha_} { with_policy(p); sigma = sigma_; n_steps = n_steps_; }
};
// PAL { q_func, alpha, gamma }                                                    (control/td/pal.rs:18-24)
struct PAL : Agent { PAL(Shared<fa::linear::LFA> q, double alpha_, double gamma) : Agent{RSRL_PAL, std::move(q), gamma, alpha_} {} };
}}  // namespace control::td

// ---- rsrl::traces: Trace::{accumulating, replacing, dutch}(dim, gamma, lambda)       (traces.rs:38-70)
namespace traces {
struct Trace {
    int rule; double gamma, lambda;
    static Trace accumulating(double gamma, double lambda) { return Trace{RSRL_TRACE_ACCUMULATE, gamma, lambda}; }
    static Trace replacing(double gamma, double lambda) { return Trace{RSRL_TRACE_SATURATE, gamma, lambda}; }
    static Trace dutch(double gamma, double lambda) { return Trace{RSRL_TRACE_DUTCH, gamma, lambda}; }
};
}  // namespace traces

namespace control { namespace td {
// SARSALambda { q_func, policy, trace, alpha, gamma } / QLambda                   (sarsa_lambda.rs:37-51, q_lambda.rs:37-54)
struct SARSALambda : Agent {
    SARSALambda(Shared<fa::linear::LFA> q, const traces::Trace& tr, double alpha_, double gamma)
        : Agent{RSRL_SARSA_LAMBDA, std::move(q), gamma, alpha_, tr.rule, tr.lambda} {}
    SARSALambda(Shared<fa::linear::LFA> q, const policies::Policy& p, const traces::Trace& tr, double alpha_, double gamma)
        : Agent{RSRL_SARSA_LAMBDA, std::move(q), gamma, alpha_, tr.rule, tr.lambda} { with_policy(p); }
};
struct QLambda : Agent {
    QLambda(Shared<fa::linear::LFA> q, const traces::Trace& tr, double alpha_, double gamma)
        : Agent{RSRL_Q_LAMBDA, std::move(q), gamma, alpha_, tr.rule, tr.lambda} {}
};
// GreedyGQ { fa_q, fa_td, behaviour_policy, gamma }                                (greedy_gq.rs:49-58)
struct GreedyGQ : Agent {
    GreedyGQ(Shared<fa::linear::LFA> fa_q, const fa::linear::LFA& fa_td, double gamma)
        : Agent{RSRL_GREEDY_GQ, std::move(fa_q), gamma, 1.0, RSRL_TRACE_ACCUMULATE, 0.0, fa_td.lr} {}
};
}}  // namespace control::td

// ---- rsrl::prediction::td: TD { v_func, gamma } / TDLambda { fa_theta, trace, gamma }   (td.rs:25-30, td_lambda.rs:25-32)
// The value function is a ScalarLFA (one weight column); drive these with policies::Random.
namespace prediction { namespace td {
struct TD : control::td::Agent {
    TD(Shared<fa::linear::LFA> v_func, double gamma) : control::td::Agent{RSRL_TD, std::move(v_func), gamma} {}
};
struct TDLambda : control::td::Agent {
    TDLambda(Shared<fa::linear::LFA> fa_theta, const traces::Trace& tr, double gamma)
        : control::td::Agent{RSRL_TD_LAMBDA, std::move(fa_theta), gamma, 1.0, tr.rule, tr.lambda} {}
};
}}  // namespace prediction::td

// ---- the bound object graph: env + agent + policy sharing one q_func on one MI355X -------------------
class Session {
public:
    Session(const domains::Domain& env, const control::td::Agent& agent, const policies::Policy& policy,
            uint64_t seed = 0, uint32_t max_episode_steps = 0, int device = 0, int64_t env_offset = 0) {
        rsrl_hip_config cfg;
        check(rsrl_hip_config_init(&cfg));
        cfg.device = device; cfg.domain = env.kind; cfg.n_envs = env.n_envs; cfg.env_offset = env_offset;
        const fa::linear::LFA& q = *agent.q_func;
        cfg.basis = q.basis_kind; cfg.order = q.order; cfg.n_tilings = q.n_tilings; cfg.tiles_per_dim = q.tiles_per_dim;
        cfg.lr = q.lr; cfg.weight_mode = q.shared_weights ? RSRL_W_SHARED : RSRL_W_PER_ENV;
        cfg.algo = agent.algo; cfg.gamma = agent.gamma; cfg.alpha = agent.alpha;
        cfg.trace = agent.trace; cfg.lambda = agent.lambda; cfg.lr_td = agent.lr_td;
        cfg.policy = policy.kind; cfg.epsilon = policy.epsilon; cfg.tau = policy.tau;
        cfg.epsilon_decay = policy.epsilon_decay; cfg.epsilon_min = policy.epsilon_min;
        // an agent-owned policy equal to the behaviour policy IS the shared object of the reference's examples
        if (agent.owns_policy && (agent.policy.kind != policy.kind || agent.policy.epsilon != policy.epsilon || agent.policy.tau != policy.tau)) {
            cfg.agent_policy = agent.policy.kind; cfg.agent_epsilon = agent.policy.epsilon; cfg.agent_tau = agent.policy.tau;
        }
        cfg.sigma = agent.sigma; cfg.n_steps = agent.n_steps;
        cfg.seed = seed; cfg.max_episode_steps = max_episode_steps;
        check(rsrl_hip_create(&cfg, &ctx_));
        D_ = rsrl_hip_state_dim(ctx_); A_ = rsrl_hip_n_actions(ctx_); F_ = rsrl_hip_n_features(ctx_); N_ = env.n_envs;
        O_ = rsrl_hip_n_outputs(ctx_);
    }
    ~Session() { rsrl_hip_destroy(ctx_); }
    Session(const Session&) = delete;
    Session& operator=(const Session&) = delete;

    int state_dim() const { return D_; }
    int n_actions() const { return A_; }              // env.action_space().card()
    int n_features() const { return F_; }
    int64_t n_envs() const { return N_; }

    // per-episode `Domain::default()` + `policy.sample(rng, env.emit().state())`   (q_learning.rs:37-38)
    void reset() { check(rsrl_hip_reset(ctx_)); }
    // Domain::emit().state()
    std::vector<float> emit() { std::vector<float> s((size_t)D_ * N_); check(rsrl_hip_get_states(ctx_, s.data())); return s; }
    // Domain::transition(a)                                                       (lib.rs:436-446)
    domains::Transition transition(const std::vector<int32_t>& a) {
        domains::Transition t;
        t.from.resize((size_t)D_ * N_); t.to.resize((size_t)D_ * N_); t.reward.resize(N_); t.terminal.resize(N_);
        t.action = a;
        check(rsrl_hip_domain_step(ctx_, a.data(), t.from.data(), t.to.data(), t.reward.data(), t.terminal.data()));
        return t;
    }
    // Handler<&Transition>::handle -> Response.error                              (q_learning.rs:51-71)
    std::vector<float> handle(const domains::Transition& t) {
        std::vector<float> td(N_);
        check(rsrl_hip_handle(ctx_, t.from.data(), t.action.data(), t.reward.data(), t.to.data(), t.terminal.data(), N_, td.data()));
        return td;
    }
    // Policy::sample / Policy::mode                                               (policies/mod.rs:65-78)
    std::vector<int32_t> sample(const std::vector<float>& states) {
        std::vector<int32_t> a(N_); check(rsrl_hip_policy_sample(ctx_, states.data(), N_, a.data())); return a;
    }
    std::vector<int32_t> mode(const std::vector<float>& states) {
        std::vector<int32_t> a(N_); check(rsrl_hip_policy_mode(ctx_, states.data(), N_, a.data())); return a;
    }
    // Function<(S,)>::evaluate                                                    (fa/linear.rs:303-311)
    std::vector<float> evaluate(const std::vector<float>& states) {
        std::vector<float> q((size_t)O_ * N_); check(rsrl_hip_q_evaluate(ctx_, states.data(), N_, q.data())); return q;
    }
    // Enumerable::find_max / find_min -> (index, value), ties -> last index        (core.rs:86-105)
    std::pair<std::vector<int32_t>, std::vector<float>> find_max(const std::vector<float>& states) {
        std::vector<int32_t> i(N_); std::vector<float> v(N_); check(rsrl_hip_q_find_max(ctx_, states.data(), N_, i.data(), v.data())); return {i, v};
    }
    std::pair<std::vector<int32_t>, std::vector<float>> find_min(const std::vector<float>& states) {
        std::vector<int32_t> i(N_); std::vector<float> v(N_); check(rsrl_hip_q_find_min(ctx_, states.data(), N_, i.data(), v.data())); return {i, v};
    }
    // Enumerable::expected_value(args, ps)                                        (core.rs:107-116); ps is [A][N]
    std::vector<float> expected_value(const std::vector<float>& states, const std::vector<float>& ps) {
        std::vector<float> e(N_); check(rsrl_hip_q_expected_value(ctx_, states.data(), N_, ps.data(), e.data())); return e;
    }
    // Function<(S,)> / Function<(S, A)> of the policy                              (greedy.rs:30-60, epsilon_greedy.rs:38-63)
    std::vector<float> policy_probs(const std::vector<float>& states) {
        std::vector<float> p((size_t)A_ * N_); check(rsrl_hip_policy_probs(ctx_, states.data(), N_, p.data())); return p;
    }
    std::vector<float> policy_prob(const std::vector<float>& states, const std::vector<int32_t>& actions) {
        std::vector<float> p(N_); check(rsrl_hip_policy_prob(ctx_, states.data(), actions.data(), N_, p.data())); return p;
    }
    void set_epsilon(double eps) { check(rsrl_hip_set_epsilon(ctx_, eps)); }      // pub field EpsilonGreedy.epsilon
    std::vector<float> epsilons() { std::vector<float> e(N_); check(rsrl_hip_get_epsilons(ctx_, e.data())); return e; }   // ... of every learner
    // Parameterised::weights()                                                    (params/mod.rs:118)
    std::vector<float> weights(int64_t env = 0) {
        std::vector<float> w((size_t)F_ * O_); check(rsrl_hip_get_weights(ctx_, env, w.data())); return w;
    }
    // the pub field `trace` of SARSALambda / QLambda                                (sarsa_lambda.rs:41)
    std::vector<float> trace(int64_t env = 0) {
        std::vector<float> z((size_t)F_ * O_); check(rsrl_hip_get_traces(ctx_, env, z.data())); return z;
    }
    // the pub field `fa_td` of GreedyGQ                                             (greedy_gq.rs:52)
    std::vector<float> td_weights(int64_t env = 0) {
        std::vector<float> v((size_t)F_ * A_); check(rsrl_hip_get_td_weights(ctx_, env, v.data())); return v;
    }
    // serde analogue: checkpoint of every learner's approximator(s)                  (rsrl/Cargo.toml:26)
    void save_weights(const std::string& path) { check(rsrl_hip_save_weights(ctx_, path.c_str())); }
    void load_weights(const std::string& path) { check(rsrl_hip_load_weights(ctx_, path.c_str())); }
    // the fused driver loop: n_steps of {transition, handle, sample} for every env, auto-reset
    rsrl_hip_stats train(int64_t n_steps) { rsrl_hip_stats st; check(rsrl_hip_train(ctx_, n_steps, &st)); return st; }
    // Domain::rollout(|s| policy.mode(s), Some(limit)).n_states()                 (lib.rs:448-479, :340)
    std::vector<uint32_t> rollout_n_states(int64_t step_limit) {
        std::vector<uint32_t> n(N_); check(rsrl_hip_rollout_greedy(ctx_, step_limit, n.data(), nullptr)); return n;
    }
    // Domain::rollout(..) as the Trajectory it returns                             (lib.rs:334-409, 448-479)
    domains::Trajectory rollout(int64_t step_limit) {
        domains::Trajectory tr;
        tr.n_states.resize(N_); tr.total_reward.resize(N_); tr.terminal.resize(N_);
        tr.states.assign((size_t)step_limit * D_ * N_, 0.0f);
        tr.actions.assign((size_t)(step_limit - 1) * N_, 0); tr.rewards.assign((size_t)(step_limit - 1) * N_, 0.0f);
        check(rsrl_hip_rollout_trajectory(ctx_, step_limit, N_, tr.n_states.data(), tr.total_reward.data(), tr.states.data(),
                                          step_limit > 1 ? tr.actions.data() : nullptr, step_limit > 1 ? tr.rewards.data() : nullptr, tr.terminal.data()));
        return tr;
    }
    // Domain::rollout(|s| policy.sample(rng, s), Some(limit)) for any policy over this session's Q function   (lib.rs:448-479)
    domains::Trajectory rollout(const policies::Policy& pi, int64_t step_limit) {
        domains::Trajectory tr;
        tr.n_states.resize(N_); tr.total_reward.resize(N_); tr.terminal.resize(N_);
        tr.states.assign((size_t)step_limit * D_ * N_, 0.0f);
        tr.actions.assign((size_t)(step_limit - 1) * N_, 0); tr.rewards.assign((size_t)(step_limit - 1) * N_, 0.0f);
        check(rsrl_hip_rollout_policy(ctx_, pi.kind, pi.epsilon, pi.tau, step_limit, N_, tr.n_states.data(), tr.total_reward.data(), tr.states.data(),
                                      step_limit > 1 ? tr.actions.data() : nullptr, step_limit > 1 ? tr.rewards.data() : nullptr, tr.terminal.data()));
        return tr;
    }
    rsrl_hip_ctx* raw() { return ctx_; }

private:
    rsrl_hip_ctx* ctx_ = nullptr;
    int D_ = 0, A_ = 0, F_ = 0, O_ = 0;      // O_: weight columns (A_, or 1 for the prediction agents)
    int64_t N_ = 0;
};

}  // namespace rsrl
