// examples/q_sigma.cpp -- QSigma (rsrl/src/control/td/q_sigma.rs) on the HIP path, composed like the reference's other examples
// (the reference ships no q_sigma example: its QSigma panics at the first full backup, see rsrl_amd/csrc/kernels_qsigma.hpp).
// MountainCar, Fourier(5).with_bias(), LFA::vector(SGD(0.002), 3), EpsilonGreedy(0.1) shared by the driver and the agent,
// QSigma::new(q_func, policy, alpha 1.0, gamma 0.99, sigma 0.5, n_steps 4); then the greedy rollout.
//
//   g++ -std=c++17 -O2 examples/q_sigma.cpp -Lrsrl_amd/lib -lrsrl_hip -Wl,-rpath,$PWD/rsrl_amd/lib -o q_sigma
#include <cstdio>
#include <cstdlib>

#include "../rsrl_amd/host/rsrl.hpp"

using namespace rsrl;

int main(int argc, char** argv) {
    const int64_t n_envs = argc > 1 ? atoll(argv[1]) : 64;
    const int batches = argc > 2 ? atoi(argv[2]) : 10;
    const int steps = argc > 3 ? atoi(argv[3]) : 2000;

    domains::MountainCar env(n_envs);
    auto basis = fa::linear::basis::Fourier::from_space(5, env).with_bias();
    auto q_func = make_shared(fa::linear::LFA::vector(basis, fa::linear::optim::SGD(0.002), 3));
    policies::EpsilonGreedy policy(policies::Greedy(q_func), policies::Random(3), 0.1);
    control::td::QSigma agent(q_func, policy, /*alpha=*/1.0, /*gamma=*/0.99, /*sigma=*/0.5, /*n_steps=*/4);

    Session sess(env, agent, policy, /*seed=*/0, /*max_episode_steps=*/1000);
    sess.reset();
    for (int e = 0; e < batches; ++e) {
        auto st = sess.train(steps);
        printf("Batch %d: %llu episodes finished, mean length %.1f steps, mean |residual| %.4f\n", e + 1, (unsigned long long)st.episodes,
               st.episodes ? (double)st.sum_episode_steps / (double)st.episodes : 0.0, st.sum_abs_td_error / (double)st.env_steps);
    }
    auto n = sess.rollout_n_states(1000);
    double mean = 0; for (auto v : n) mean += v;
    printf("OOS: %.1f states on average...\n", mean / n_envs);
    return 0;
}
