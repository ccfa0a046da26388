// examples/pal.cpp -- the reference's rsrl/examples/pal.rs on the HIP path: MountainCar, Fourier(5).with_bias(),
// LFA::vector(SGD(1.0), 3), EpsilonGreedy(0.1), PAL { alpha: 0.001, gamma: 0.9 } -- N environments instead of one.
//
//   g++ -std=c++17 -O2 examples/pal.cpp -Lrsrl_amd/lib -lrsrl_hip -Wl,-rpath,$PWD/rsrl_amd/lib -o pal
#include <cstdio>
#include <cstdlib>

#include "../rsrl_amd/host/rsrl.hpp"

using namespace rsrl;

int main(int argc, char** argv) {
    const int64_t n_envs = argc > 1 ? atoll(argv[1]) : 64;
    const int batches = argc > 2 ? atoi(argv[2]) : 10;
    const int steps = argc > 3 ? atoi(argv[3]) : 1000;

    domains::MountainCar env(n_envs);
    auto basis = fa::linear::basis::Fourier::from_space(5, env).with_bias();
    auto q_func = make_shared(fa::linear::LFA::vector(basis, fa::linear::optim::SGD(1.0), 3));
    policies::EpsilonGreedy policy(policies::Greedy(q_func), policies::Random(3), 0.1);
    control::td::PAL ql(q_func, /*alpha=*/0.001, /*gamma=*/0.9);

    Session sess(env, ql, policy, /*seed=*/0, /*max_episode_steps=*/1000);
    sess.reset();
    for (int e = 0; e < batches; ++e) {
        auto st = sess.train(steps);
        printf("Batch %d: %llu episodes finished, mean |residual| %.4f\n", e + 1, (unsigned long long)st.episodes,
               st.sum_abs_td_error / (double)st.env_steps);
    }
    auto n = sess.rollout_n_states(500);                             // rollout(|s| policy.mode(s), Some(500)).n_states()
    double mean = 0; for (auto v : n) mean += v;
    printf("OOS: %.1f states on average...\n", mean / n_envs);
    return 0;
}
