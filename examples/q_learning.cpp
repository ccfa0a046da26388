// examples/q_learning.cpp -- the reference's rsrl/examples/q_learning.rs on the HIP path: same object graph,
// same hyper-parameters (MountainCar, Fourier(5).with_bias(), LFA::vector(SGD(0.001), 3), Greedy, QLearning gamma 0.9),
// N environments instead of one, the per-step calls going through the trait-shaped C++ mirror (rsrl_amd/host/rsrl.hpp).
//
//   g++ -std=c++17 -O2 examples/q_learning.cpp -Lrsrl_amd/lib -lrsrl_hip -Wl,-rpath,$PWD/rsrl_amd/lib -o q_learning
#include <cstdio>
#include <cstdlib>

#include "../rsrl_amd/host/rsrl.hpp"

using namespace rsrl;

int main(int argc, char** argv) {
    const int64_t n_envs = argc > 1 ? atoll(argv[1]) : 64;
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;

    domains::MountainCar env(n_envs);
    auto basis = fa::linear::basis::Fourier::from_space(5, env).with_bias();
    auto q_func = make_shared(fa::linear::LFA::vector(basis, fa::linear::optim::SGD(0.001), 3));
    policies::Greedy policy(q_func);
    control::td::QLearning ql(q_func, 0.9);

    Session sess(env, ql, policy, /*seed=*/0, /*max_episode_steps=*/1000);

    // the reference's loop (q_learning.rs:34-55), one trait call at a time
    sess.reset();
    std::vector<int32_t> action(n_envs);
    check(rsrl_hip_get_actions(sess.raw(), action.data()));          // initial policy.sample(env.emit().state())
    double abs_td = 0.0;
    for (int i = 0; i < 50; ++i) {
        auto t = sess.transition(action);                            // env.transition(action)
        auto td = sess.handle(t);                                    // ql.handle(&t)
        action = sess.sample(t.to);                                  // policy.sample(&mut rng, t.to.state())
        for (float v : td) abs_td += v < 0 ? -v : v;
    }
    printf("50 per-call steps: mean |td error| = %.6f\n", abs_td / (50.0 * n_envs));

    // the same loop fused on the device
    auto st = sess.train(steps);
    printf("fused: %llu env-steps, %llu episodes finished (%llu truncated), mean |td| %.6f\n",
           (unsigned long long)st.env_steps, (unsigned long long)st.episodes, (unsigned long long)st.episodes_truncated,
           st.sum_abs_td_error / (double)st.env_steps);

    auto n = sess.rollout_n_states(500);                             // rollout(|s| policy.mode(s), Some(500)).n_states()
    double mean = 0; for (auto v : n) mean += v;
    printf("OOS: %.1f states on average...\n", mean / n_envs);

    // the rest of the trait surface on the path: the Trajectory itself, Enumerable::find_min / expected_value,
    // Function<(S, A)> of the policy (lib.rs:334-409, core.rs:86-116, greedy.rs:46-60)
    auto tr = sess.rollout(500);
    bool same = true;
    for (int64_t i = 0; i < n_envs; ++i) same = same && tr.n_states[i] == n[i];
    auto s0 = sess.emit();
    auto mx = sess.find_max(s0); auto mn = sess.find_min(s0);
    auto ev = sess.expected_value(s0, sess.policy_probs(s0));       // under Greedy: the maximum (unless there are ties)
    auto pa = sess.policy_prob(s0, mx.first);
    bool ordered = true, greedy_ev = true;
    for (int64_t i = 0; i < n_envs; ++i) {
        ordered = ordered && mn.second[i] <= mx.second[i];
        greedy_ev = greedy_ev && (pa[i] < 1.0f || ev[i] == mx.second[i]);
    }
    printf("trajectory == rollout: %d, min <= max: %d, E_greedy[Q] == max Q: %d, first trajectory: %u states, total reward %.1f\n",
           (int)same, (int)ordered, (int)greedy_ev, tr.n_states[0], tr.total_reward[0]);
    return (same && ordered && greedy_ev) ? 0 : 1;
}
