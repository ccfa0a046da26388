// examples/sarsa_lambda.cpp -- the reference's rsrl/examples/sarsa_lambda.rs on the HIP path: MountainCar, Fourier(5).with_bias(),
// LFA::vector(SGD(1.0), 3), EpsilonGreedy(0.2) with `agent.policy.epsilon *= 0.995` after every EPISODE (:68) -- of every one of
// the N learners, each carrying its own epsilon, decayed on the device when its episode ends --, Trace::replacing(gamma 0.99,
// lambda 0.7), SARSALambda alpha 0.01; each "batch" = `steps` fused batch-steps (episodes restart on the device), then the greedy
// rollout of the reference's last two lines.
//
//   g++ -std=c++17 -O2 examples/sarsa_lambda.cpp -Lrsrl_amd/lib -lrsrl_hip -Wl,-rpath,$PWD/rsrl_amd/lib -o sarsa_lambda
#include <cstdio>
#include <cstdlib>

#include "../rsrl_amd/host/rsrl.hpp"

using namespace rsrl;

int main(int argc, char** argv) {
    const int64_t n_envs = argc > 1 ? atoll(argv[1]) : 64;
    const int batches = argc > 2 ? atoi(argv[2]) : 20;
    const int steps = argc > 3 ? atoi(argv[3]) : 500;
    const double ALPHA = 0.01, GAMMA = 0.99, LAMBDA = 0.7;

    domains::MountainCar env(n_envs);
    auto basis = fa::linear::basis::Fourier::from_space(5, env).with_bias();
    auto fa_theta = make_shared(fa::linear::LFA::vector(basis, fa::linear::optim::SGD(1.0), 3));
    policies::EpsilonGreedy policy(policies::Greedy(fa_theta), policies::Random(3), 0.2);
    policy.decayed_per_episode(0.995);                                // agent.policy.epsilon *= 0.995, once per episode of the learner
    auto trace = traces::Trace::replacing(GAMMA, LAMBDA);
    control::td::SARSALambda agent(fa_theta, policy, trace, ALPHA, GAMMA);      // SARSALambda { fa_theta, policy, trace, alpha, gamma }

    Session sess(env, agent, policy, /*seed=*/0, /*max_episode_steps=*/1000);
    sess.reset();
    for (int e = 0; e < batches; ++e) {
        auto st = sess.train(steps);
        auto eps = sess.epsilons();
        double lo = 1.0, hi = 0.0; for (float v : eps) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        printf("Batch %d: %llu episodes finished, mean length %.1f steps, epsilon in [%.4f, %.4f]...\n", e + 1, (unsigned long long)st.episodes,
               st.episodes ? (double)st.sum_episode_steps / (double)st.episodes : 0.0, lo, hi);
    }
    auto z = sess.trace(0);
    double zmax = 0; for (float v : z) zmax = (v < 0 ? -v : v) > zmax ? (v < 0 ? -v : v) : zmax;
    printf("max |trace| of learner 0: %.4f\n", zmax);
    auto n = sess.rollout_n_states(1000);                            // rollout(|s| agent.policy.mode(s), Some(1000)).n_states()
    double mean = 0; for (auto v : n) mean += v;
    printf("OOS: %.1f states on average...\n", mean / n_envs);
    return 0;
}
