// examples/greedy_gq.cpp -- the reference's rsrl/examples/greedy_gq.rs on the HIP path: MountainCar, Fourier(3).with_bias(),
// fa_q = LFA::vector(SGD(0.1)), fa_td = LFA::vector(SGD(0.001)), EpsilonGreedy(0.1), GreedyGQ gamma 0.99, episodes capped at
// 1000 steps -- N environments instead of one.
//
//   g++ -std=c++17 -O2 examples/greedy_gq.cpp -Lrsrl_amd/lib -lrsrl_hip -Wl,-rpath,$PWD/rsrl_amd/lib -o greedy_gq
#include <cstdio>
#include <cstdlib>

#include "../rsrl_amd/host/rsrl.hpp"

using namespace rsrl;

int main(int argc, char** argv) {
    const int64_t n_envs = argc > 1 ? atoll(argv[1]) : 64;
    const int batches = argc > 2 ? atoi(argv[2]) : 10;
    const int steps = argc > 3 ? atoi(argv[3]) : 1000;

    domains::MountainCar env(n_envs);
    auto basis = fa::linear::basis::Fourier::from_space(3, env).with_bias();
    auto fa_q = make_shared(fa::linear::LFA::vector(basis, fa::linear::optim::SGD(0.1), 3));
    auto fa_td = fa::linear::LFA::vector(basis, fa::linear::optim::SGD(0.001), 3);
    policies::EpsilonGreedy policy(policies::Greedy(fa_q), policies::Random(3), 0.1);
    control::td::GreedyGQ agent(fa_q, fa_td, 0.99);

    Session sess(env, agent, policy, /*seed=*/0, /*max_episode_steps=*/1000);
    sess.reset();
    for (int e = 0; e < batches; ++e) {
        auto st = sess.train(steps);
        printf("Batch %d: %llu episodes finished (%llu truncated), mean |td| %.4f\n", e + 1, (unsigned long long)st.episodes,
               (unsigned long long)st.episodes_truncated, st.sum_abs_td_error / (double)st.env_steps);
    }
    auto v = sess.td_weights(0);
    double vmax = 0; for (float x : v) vmax = (x < 0 ? -x : x) > vmax ? (x < 0 ? -x : x) : vmax;
    printf("max |fa_td weight| of learner 0: %.5f\n", vmax);
    auto n = sess.rollout_n_states(500);                             // rollout(|s| agent.fa_q.find_max((s,)).0, Some(500))
    double mean = 0; for (auto x : n) mean += x;
    printf("OOS: %.1f states on average...\n", mean / n_envs);
    return 0;
}
