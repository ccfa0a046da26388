/*
 * rsrl_oracle.h -- declarations of the CPU oracle (TEST INFRASTRUCTURE ONLY;
 * see the header of rsrl_oracle.c for scope and pinning status).
 */
#ifndef RSRL_ORACLE_H
#define RSRL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MOUNTAIN_CAR = 0, ORC_CART_POLE = 1, ORC_ACROBOT = 2 };
enum { ORC_FOURIER = 0, ORC_TILE = 1 };
enum { ORC_QLEARNING = 0, ORC_SARSA = 1, ORC_EXPECTED_SARSA = 2, ORC_SARSA_LAMBDA = 3, ORC_Q_LAMBDA = 4, ORC_PAL = 5,
       ORC_GREEDY_GQ = 6, ORC_TD = 7, ORC_TD_LAMBDA = 8, ORC_Q_SIGMA = 9 };
#define ORC_IS_LAMBDA(algo) ((algo) == ORC_SARSA_LAMBDA || (algo) == ORC_Q_LAMBDA)
/* agents with a second per-learner matrix of W's shape: the trace Z (lambda agents) or fa_td's weights (GreedyGQ) */
/* prediction agents: ONE weight column (ScalarLFA, the state-value function); the behaviour policy must be Random */
#define ORC_IS_PRED(algo) ((algo) == ORC_TD || (algo) == ORC_TD_LAMBDA)
#define ORC_HAS_AUX(algo) (ORC_IS_LAMBDA(algo) || (algo) == ORC_GREEDY_GQ || (algo) == ORC_TD_LAMBDA)
/* columns of the weight matrix */
#define ORC_N_OUT(ag) (ORC_IS_PRED((ag)->algo) ? 1 : (ag)->n_actions)
/* eligibility-trace update rules (rsrl/src/traces.rs:188-240) */
enum { ORC_TRACE_ACCUMULATE = 0, ORC_TRACE_SATURATE = 1, ORC_TRACE_DUTCH = 2 };
enum { ORC_GREEDY = 0, ORC_EGREEDY = 1, ORC_SOFTMAX = 2, ORC_RANDOM = 3 };
/* RNG draw blocks (counter word 3) */
enum { ORC_BLK_STEP = 0, ORC_BLK_RESET = 1, ORC_BLK_INNER = 2, ORC_BLK_INIT = 3, ORC_BLK_API = 4, ORC_BLK_ROLLOUT = 5 };

#define ORC_MAX_ACTIONS 8
#define ORC_MAX_TILINGS 32

typedef struct {
    int kind, dim, order, n_tilings, tiles_per_dim;
    double lo[8], hi[8];
    float lo_f[8], hi_f[8];
} orc_basis;

typedef struct {
    int domain, algo, policy, shared_w, n_actions;
    orc_basis basis;
    uint64_t seed;
    int64_t env_offset;
    double gamma, lr, alpha, epsilon, tau;
    uint32_t eps_thr;
    uint32_t max_episode_steps;
    double lambda;       /* eligibility traces */
    int trace;
    double lr_td;        /* GreedyGQ: SGD rate of fa_td (examples/greedy_gq.rs:27) */
    /* the policy OWNED BY THE AGENT (SARSA{q_func, policy, gamma} sarsa.rs:35-41; ExpectedSARSA expected_sarsa.rs:22-29;
     * SARSALambda sarsa_lambda.rs:37-44): orc_agent_init copies the behaviour policy (the examples share one object) */
    int apolicy;
    double aepsilon, atau;
    uint32_t aeps_thr;
    /* QSigma{.., sigma, backup: Backup::new(n_steps)}  (q_sigma.rs:80-105) */
    double sigma;
    int n_steps;
    /* the drivers' epsilon schedule (examples/sarsa_lambda.rs:48-75): `agent.policy.epsilon *= eps_decay` (floored at eps_min) once
     * per episode of a learner, after the episode's last handle / sample and before the next episode's initial sample (:68).
     * eps_decay == 1 (or 0, the memset default): no schedule.  apol_same: the agent's policy IS the behaviour policy object (what the
     * examples build with make_shared), so the schedule moves both. */
    int apol_same;
    double eps_decay, eps_min;
} orc_agent;
#define ORC_MAX_NSTEPS 32

typedef struct {
    uint64_t env_steps, episodes, episodes_truncated, sum_episode_steps;
    double sum_abs_td_error, sum_reward;
} orc_stats;

int  orc_domain_dim(int domain);
int  orc_domain_actions(int domain);
void orc_domain_bounds(int domain, double* lo, double* hi);
void orc_basis_init(orc_basis* b, int domain, int kind, int order, int n_tilings, int tiles_per_dim);
int  orc_basis_nfeat(const orc_basis* b);
void orc_tile_indices(const orc_basis* b, const float* s, int* idx);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_draw(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t block, uint32_t out[4]);
uint32_t orc_mulhi(uint32_t x, uint32_t n);
uint32_t orc_eps_threshold(double eps);
void orc_agent_init(orc_agent* ag, int domain, int basis_kind, int order, int n_tilings, int tiles_per_dim,
                    int algo, int policy, int shared_w, uint64_t seed, int64_t env_offset,
                    double gamma, double lr, double alpha, double epsilon, double tau,
                    uint32_t max_episode_steps);

#define ORC_DECLARE(R, S)                                                                               \
    void  orc_domain_reset_##S(int domain, R* s);                                                       \
    int   orc_domain_is_terminal_##S(int domain, const R* s);                                           \
    int   orc_domain_step_##S(int domain, R* s, int a, R* reward);                                      \
    void  orc_fourier_project_##S(int order, int D, const R* lo, const R* hi, const R* s, R* phi);      \
    void  orc_q_evaluate_##S(const orc_basis* b, const R* W, int A, const R* s, R* q);                  \
    R     orc_q_evaluate_index_##S(const orc_basis* b, const R* W, int A, const R* s, int a);           \
    int   orc_find_max_##S(const R* q, int A, R* val);                                                  \
    void  orc_q_update_index_##S(const orc_basis* b, R* W, int A, const R* s, int a, R lr, R error);    \
    int   orc_argmaxima_##S(const R* v, int n, int* ixs, R* maxv);                                      \
    int   orc_argmax_first_##S(const R* v, int n);                                                      \
    void  orc_greedy_probs_##S(const R* q, int A, R* p);                                                \
    void  orc_egreedy_probs_##S(const R* q, int A, R eps, R* p);                                        \
    void  orc_softmax_probs_##S(const R* q, int A, R tau, R* p);                                        \
    void  orc_policy_probs_##S(int policy, const R* q, int A, R eps, R tau, R* p);                      \
    int   orc_policy_sample_##S(int policy, const R* q, int A, uint32_t eps_thr, R tau,                 \
                                const uint32_t x[4]);                                                   \
    int   orc_policy_mode_##S(int policy, const R* q, int A, R tau);                                    \
    R     orc_td_error_##S(const orc_agent* ag, const R* W, const R* s, int a, R r, const R* ns,        \
                           int term, const uint32_t x_inner[4], R* delta_out);                          \
    R     orc_handle_##S(const orc_agent* ag, R* W, const R* s, int a, R r, const R* ns, int term,      \
                         const uint32_t x_inner[4]);                                                    \
    void* orc_run_create_##S(const orc_agent* ag, int64_t n_envs);                                      \
    void  orc_run_destroy_##S(void* h);                                                                 \
    R*    orc_run_state_##S(void* h);                                                                   \
    int32_t*  orc_run_action_##S(void* h);                                                              \
    uint32_t* orc_run_ep_step_##S(void* h);                                                             \
    R*    orc_run_weights_##S(void* h);                                                                 \
    uint64_t  orc_run_t_##S(void* h);                                                                   \
    void  orc_run_set_epsilon_##S(void* h, double eps);                                                 \
    void  orc_run_reset_##S(void* h);                                                                   \
    R     orc_handle_lambda_##S(const orc_agent* ag, R* W, R* Z, const R* s, int a, R r, const R* ns, int term,   \
                                const uint32_t x_inner[4]);                                             \
    R     orc_handle_gq_##S(const orc_agent* ag, R* W, R* V, const R* s, int a, R r, const R* ns, int term);              \
    R     orc_v_evaluate_##S(const orc_basis* b, const R* w, const R* s);                                         \
    R     orc_handle_td_##S(const orc_agent* ag, R* w, R* z, const R* s, R r, const R* ns, int term);             \
    int   orc_run_train_fast_##S(void* h, int64_t n_steps, orc_stats* st);                                        \
    int   orc_run_train_dev_##S(void* h, int64_t n_steps, orc_stats* st);                                         \
    void  orc_run_invalidate_q_##S(void* h);                                                                      \
    int   orc_run_train_wave_##S(void* h, int64_t n_steps, orc_stats* st, int w_bf16);                            \
    void  orc_run_reset_wave_##S(void* h);                                                                        \
    int   orc_run_train_shared_dev_##S(void* h, int64_t n_steps, orc_stats* st);                                  \
    void* orc_qsigma_new_##S(int n_steps);                                                                        \
    void  orc_qsigma_free_##S(void* backup);                                                                      \
    int   orc_qsigma_len_##S(const void* backup);                                                                 \
    R     orc_handle_qsigma_##S(const orc_agent* ag, R* W, void* backup, const R* s, int a, R r, const R* ns,     \
                                int term, const uint32_t x_inner[4]);                                             \
    R*    orc_run_traces_##S(void* h);                                                                  \
    R*    orc_run_eps_##S(void* h);                                                                     \
    void  orc_run_train_##S(void* h, int64_t n_steps, orc_stats* st);                                   \
    int   orc_run_train_sparse_lambda_##S(void* h, int64_t n_steps, orc_stats* st);                   \
    void  orc_run_sparse_trace_##S(void* h, int64_t i, R* out);                                         \
    void  orc_run_teacher_##S(void* h, orc_stats* st, R* from, int32_t* act, R* rew, R* to, uint8_t* term, R* td); \
    int   orc_run_teacher_sparse_lambda_##S(void* h, orc_stats* st, R* from, int32_t* act, R* rew, R* to, uint8_t* term, R* td); \
    void  orc_run_train_hook_##S(void* h, int64_t n_steps, orc_stats* st,                               \
                                 void (*dw_hook)(R* dW, int n, void* user), void* user);                \
    int   orc_run_rollout_greedy_##S(void* h, int64_t step_limit, uint32_t* n_states, R* total_reward); \
    int   orc_run_rollout_greedy_margin_##S(void* h, int64_t step_limit, uint32_t* n_states, R* total_reward, R* min_margin); \
    int   orc_run_rollout_policy_##S(void* h, int policy, double eps, double tau, uint64_t call, int64_t step_limit, uint32_t* n_states, R* total_reward, int32_t* actions);

ORC_DECLARE(double, f64)
ORC_DECLARE(float, f32)
/* f32 with the device's sincos / exp polynomials restated: bitwise comparison with the HIP path (rsrl_oracle_impl.h) */
ORC_DECLARE(float, f32d)

#ifdef __cplusplus
}
#endif
#endif
