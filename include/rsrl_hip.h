/*
 * rsrl_hip.h -- C ABI of librsrl_hip.so: the MI355X (gfx950) implementation of
 * tspooner/rsrl's per-step TD-control hot path
 *
 *     env.transition(a) -> agent.handle(&t) -> policy.sample(s')
 *                                   (rsrl/examples/q_learning.rs:34-55)
 *
 * vectorised over N independent environments.  The reference has no FFI of its
 * own: its extension surface is Rust traits (Domain / Function / Enumerable /
 * Handler / Policy / Parameterised).  Every entry point below stands in for one
 * trait method on the path and cites it (paths relative to the reference
 * repository).  INTEGRATION.md shows the Rust `extern "C"` block and the trait
 * impls a maintainer would write on top of it.
 *
 * Conventions
 *   - every function returns rsrl_hip_status (0 = OK, <0 = error); the message of
 *     the last error on the calling thread is rsrl_hip_last_error().
 *   - a ctx is NOT thread-safe (mirrors `&mut self` + Rc<RefCell>, core.rs:13-15);
 *     distinct ctxs may be used from distinct threads.
 *   - batched arrays are SoA, component-major: states f32[D][M], Q f32[A][M];
 *     actions int32[M]; terminal flags uint8[M]; rewards / td errors f32[M].
 *     A batch of M items addresses learners 0..M-1 of the ctx (M <= n_envs).
 *   - array arguments may be HOST or DEVICE pointers (detected per call with
 *     hipPointerGetAttributes); device pointers are used in place, host pointers
 *     are staged through ctx-owned device buffers.  Nothing is freed by the
 *     other side.
 *   - all kernels of a ctx run on ONE HIP stream (the caller's, if given in the
 *     config, else a ctx-owned one).  Calls taking host output pointers return
 *     after the data has landed; calls with device pointers are asynchronous on
 *     that stream (use rsrl_hip_sync).  On a CALLER-supplied stream every call has
 *     enqueued all of its work when it returns: hipStreamSynchronize / an event on
 *     that stream orders the caller's own work after it (launch coalescing, below,
 *     is for ctx-owned streams only).
 *   - there is no CPU fallback: without a usable HIP device rsrl_hip_create fails.
 */
#ifndef RSRL_HIP_H
#define RSRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSRL_HIP_ABI_VERSION 9

typedef enum {
    RSRL_HIP_OK      = 0,
    RSRL_HIP_EINVAL  = -1,   /* bad argument / unsupported combination              */
    RSRL_HIP_EHIP    = -2,   /* HIP runtime error (incl. no device)                  */
    RSRL_HIP_ENOMEM  = -3,   /* allocation failed                                    */
    RSRL_HIP_ERCCL   = -4,   /* RCCL error                                           */
    RSRL_HIP_ESTATE  = -5    /* call not valid in the ctx's current state            */
} rsrl_hip_status;

/* rsrl_domains::{MountainCar, CartPole, Acrobot}
 *   mountain_car/discrete.rs:8-102, cart_pole.rs:7-121, acrobot.rs:8-152 */
typedef enum { RSRL_MOUNTAIN_CAR = 0, RSRL_CART_POLE = 1, RSRL_ACROBOT = 2 } rsrl_domain;
/* lfa::basis::{Fourier (+with_bias), TileCoding}  (re-exported by rsrl/src/fa/linear.rs:11-14) */
typedef enum { RSRL_FOURIER = 0, RSRL_TILE_CODING = 1 } rsrl_basis;
/* rsrl::control::td::{QLearning, SARSA, ExpectedSARSA}
 *   q_learning.rs:35-71, sarsa.rs:35-75, expected_sarsa.rs:22-66
 * the eligibility-trace agents {SARSALambda, QLambda}
 *   sarsa_lambda.rs:37-98, q_lambda.rs:37-99 (per-learner weights; register-family Fourier bases, every other Fourier order (round 5: W and
 *   the trace in memory, one thread per learner, rsrl_amd/csrc/kernels_lambda_mem.hpp), the order-7 Fourier bases of
 *   the 4-D domains with f32 weights (W and the trace streamed from memory every step), or tile coding with one dense trace
 *   table of W's shape per learner -- the reference's traces are generic over the gradient buffer, traces.rs:6-12;
 *   round 5: weight_mode = RSRL_W_SHARED on tile coding -- ONE shared table, every learner its own SPARSE trace as params/sparse.rs:13-97
 *   offers: at most 512 (entry, value) pairs, kept as one sub-list of 512 / n_tilings per tiling (n_tilings 4, 8 or 16; a tiling's slice of the
 *   table, cells * actions, at most 65 536 entries), the entry with the smallest |value| of a full sub-list making room; the table moves by the
 *   synchronous mini-batch rule W += sum_i alpha * residual_i * z_i in exact 64-bit fixed point.  Stepped by rsrl_hip_train, or by
 *   rsrl_hip_handle with transition i taken as LEARNER i's (its trace is the one that moves; M <= n_envs: ABI unchanged, round 6);
 *   rsrl_hip_get_traces shows a learner's list as the dense (F, A) matrix it stands for; the lists travel with a checkpoint, file version 6)
 * PAL (persistent advantage learning), pal.rs:18-60 -- a drop-in sibling of QLearning (uses `alpha`)
 * and GreedyGQ, greedy_gq.rs:49-142 -- fa_q (SGD(lr)) plus a second approximator fa_td (SGD(lr_td), weights through
 *   rsrl_hip_get/set_td_weights); per-learner weights: register-family Fourier bases, the generic Fourier orders, tile coding, and (round 5)
 *   the order-7 wave family (one wavefront per learner, W and V streamed: kernels_wave_aux.hpp; round 6: W as bf16 with stochastic rounding too -- for every agent of
 *   that family: the trace / fa_td's weights / QSigma's backups stay f32) */
typedef enum { RSRL_QLEARNING = 0, RSRL_SARSA = 1, RSRL_EXPECTED_SARSA = 2, RSRL_SARSA_LAMBDA = 3, RSRL_Q_LAMBDA = 4,
               RSRL_PAL = 5, RSRL_GREEDY_GQ = 6,
               /* prediction (state-value function on a ScalarLFA, ONE weight column; behaviour policy RSRL_RANDOM):
                *   TD prediction/td/td.rs:25-59 (SGD(lr)), TDLambda prediction/td/td_lambda.rs:25-78 (step = the TD error);
                *   per-learner weights: register-family Fourier bases (fused, register-resident), the other Fourier orders (one thread
                *   per learner, w and z in memory), tile coding (one block per learner) or (round 5) the order-7 wave family (f32 weights; bf16 since round 6)
                *   (one wavefront per learner, w and z streamed: kernels_wave_aux.hpp) */
               RSRL_TD = 7, RSRL_TD_LAMBDA = 8,
               /* QSigma, the n-step Q(sigma) agent (control/td/q_sigma.rs:80-202; config.sigma, config.n_steps, alpha, gamma, and the
                * agent's own policy).  The reference panics at its first full backup -- Backup::propagate reads entries[n_steps]
                * (q_sigma.rs:52-53 vs :113-114) -- so the library implements it with that one dead out-of-bounds read removed
                * (rsrl_amd/csrc/kernels_qsigma.hpp).  Per-learner f32 weights on every basis: register-family and generic Fourier orders, tile
                * coding, and (round 5) the order-7 wave family (one wavefront per learner, kernels_wave_aux.hpp k_wave_qsigma). */
               RSRL_Q_SIGMA = 9 } rsrl_algo;
/* rsrl::traces::{Accumulate, Saturate (Trace::replacing), Dutch}      traces.rs:188-240 */
typedef enum { RSRL_TRACE_ACCUMULATE = 0, RSRL_TRACE_SATURATE = 1, RSRL_TRACE_DUTCH = 2 } rsrl_trace;
/* rsrl::policies::{Greedy, EpsilonGreedy, Softmax, Random}
 *   greedy.rs:16-84, epsilon_greedy.rs:14-83, softmax.rs:55-143, random.rs:13-48 */
typedef enum { RSRL_GREEDY = 0, RSRL_EPSILON_GREEDY = 1, RSRL_SOFTMAX = 2, RSRL_RANDOM = 3 } rsrl_policy;
/* one LFA per learner (independent replicas) or one LFA shared by all learners
 * (synchronous mini-batch rule; N=1 == the reference rule) */
typedef enum { RSRL_W_PER_ENV = 0, RSRL_W_SHARED = 1 } rsrl_weight_mode;
typedef enum { RSRL_W_F32 = 0, RSRL_W_BF16 = 1 } rsrl_weight_dtype;
/* multi-rank shared-W: the per-batch-step exchange of the (F x A) weight delta (no reference counterpart)
 *   RCCL : ncclAllReduce(sum) on the ctx's stream (one process per GPU, any topology)
 *   PEER : one-hop peer-write -- every rank stores its delta into a slot of every other rank's receive buffer
 *          (hipIpc-mapped memory: xGMI stores across GPUs) and each rank sums the slots in rank order: deterministic,
 *          ring-free, one fabric hop (single node)
 *   AUTO : (the default) decided when the exchange is attached: rsrl_hip_comm_init makes it RCCL, rsrl_hip_peer_export makes it PEER,
 *          rsrl_hip_group_create takes PEER whenever every device of the group can access every other one's memory (one hop over xGMI
 *          beats a ring of latency-bound all-reduces at 432 B; the persistent kernel exchanges inside its one launch) and RCCL, the
 *          any-topology fallback, otherwise.  rsrl_hip_can_access_peer lets a multi-process host make the same choice. */
typedef enum { RSRL_EXCHANGE_RCCL = 0, RSRL_EXCHANGE_PEER = 1, RSRL_EXCHANGE_AUTO = 2 } rsrl_exchange;

typedef struct rsrl_hip_ctx rsrl_hip_ctx;

/* Everything the reference spreads over its constructors
 *   MountainCar::default(), Fourier::from_space(n, space).with_bias(),
 *   LFA::vector(basis, SGD(lr), n_actions), make_shared, Greedy::new /
 *   EpsilonGreedy::new / Softmax::new, QLearning{q_func, gamma} ...
 *   (rsrl/examples/q_learning.rs:19-32)                                      */
typedef struct {
    uint32_t struct_size;        /* = sizeof(rsrl_hip_config); checked                      */
    int32_t  device;             /* HIP device ordinal                                       */
    int32_t  domain;             /* rsrl_domain                                              */
    int32_t  basis;              /* rsrl_basis                                               */
    int32_t  order;              /* Fourier order n: F = (n+1)^D                             */
    int32_t  n_tilings;          /* tile coding: T tilings ...                               */
    int32_t  tiles_per_dim;      /*   ... x B^D cells: F = T*B^D                             */
    int32_t  algo;               /* rsrl_algo                                                */
    int32_t  policy;             /* rsrl_policy (behaviour policy; also SARSA's / ExpectedSARSA's) */
    int32_t  weight_mode;        /* rsrl_weight_mode                                         */
    int32_t  weight_dtype;       /* rsrl_weight_dtype (storage; arithmetic is always f32)    */
    uint32_t max_episode_steps;  /* 0 = unbounded (examples/q_learning.rs:40); else cap + auto-reset
                                    (examples/greedy_gq.rs:46 uses 1000)                     */
    int64_t  n_envs;             /* learners owned by this ctx                                */
    int64_t  env_offset;         /* global id of local learner 0: RNG streams are keyed by the
                                    GLOBAL id, so results do not depend on the sharding       */
    uint64_t seed;               /* StdRng::seed_from_u64 analogue (q_learning.rs:22)        */
    double   gamma;              /* QLearning.gamma etc.                                      */
    double   lr;                 /* SGD(lr)                                                   */
    double   alpha;              /* ExpectedSARSA.alpha (expected_sarsa.rs:26,64)             */
    double   epsilon;            /* EpsilonGreedy.epsilon (pub field, epsilon_greedy.rs:19)  */
    double   tau;                /* Softmax.tau (softmax.rs:52); |tau| < 1e-7 is rejected (:63-66) */
    uint32_t steps_per_launch;   /* fuse depth of rsrl_hip_train (0 = library default: 4096 for the register-resident loops,
                                    256 for the memory-resident and wave-family ones). 1 = one batch-step per launch:
                                    the ctx then keeps W learner-major and streams it once per step (the 608 B/env-step
                                    formulation), replayed as a hipGraph; results are bit-identical for every depth */
    int32_t  trace;              /* rsrl_trace (lambda agents)                                */
    void*    stream;             /* hipStream_t to run on; NULL = ctx-owned stream           */
    double   lambda;             /* Trace::{accumulating,replacing,dutch}(dim, gamma, lambda) (examples/sarsa_lambda.rs:37);
                                    the lambda agents step with `alpha` and bypass SGD(lr) (fa/linear.rs:184-196) */
    double   lr_td;              /* GreedyGQ: SGD rate of fa_td (examples/greedy_gq.rs:27 uses 0.001 next to SGD(0.1) for fa_q) */
    /* ---- ABI 4 (struct_size-versioned: a caller built against ABI 3 passes the shorter struct and gets the defaults) ---- */
    int32_t  agent_policy;       /* the policy OWNED BY THE AGENT: SARSA{q_func, policy, gamma} draws its inner a' from it
                                    (sarsa.rs:35-41,61), ExpectedSARSA{.., policy, ..} takes its expectation under it
                                    (expected_sarsa.rs:22-29,52-58), SARSALambda likewise (sarsa_lambda.rs:37-44).
                                    -1 (default) = the behaviour policy object itself (the examples share one policy through
                                    make_shared); otherwise an rsrl_policy with its own parameters below, e.g. a Greedy
                                    target under an EpsilonGreedy behaviour = off-policy ExpectedSARSA                   */
    int32_t  exchange;           /* rsrl_exchange: how ranks exchange the shared-W delta (rsrl_hip_comm_init)            */
    double   agent_epsilon;      /* EpsilonGreedy.epsilon of the agent's policy                                          */
    double   agent_tau;          /* Softmax.tau of the agent's policy                                                    */
    double   sigma;              /* QSigma.sigma in [0, 1]: 1 = SARSA-like sampling, 0 = tree backup (q_sigma.rs:66-72)          */
    int32_t  n_steps;            /* QSigma: Backup::new(n_steps), 1..32 (q_sigma.rs:94-104)                                      */
    int32_t  peer_timeout_ms;    /* ABI 5 (was reserved0): bound of every in-kernel wait for a peer / block of the shared-W exchange, in
                                    milliseconds; 0 = RSRL_PEER_TIMEOUT_MS from the environment, else 4000.  Make it longer than the
                                    longest time one rank may spend away from the others (a rollout, a checkpoint) */
    /* ---- ABI 6 ---- */
    double   epsilon_decay;      /* the reference drivers' epsilon schedule, `agent.policy.epsilon *= 0.995` once per EPISODE of the learner
                                    (examples/sarsa_lambda.rs:48-75, :68; the pub field epsilon_greedy.rs:19).  1.0 (default) = no
                                    schedule: one epsilon for the whole ctx (rsrl_hip_set_epsilon).  In (0, 1): EpsilonGreedy.epsilon is
                                    a field of every LEARNER; whenever an episode of learner i ends (terminal transition or step cap),
                                    after that episode's last handle and before the next episode's initial sample,
                                    eps_i <- max(eps_i * epsilon_decay, epsilon_min).  One learner reproduces the reference loop's
                                    schedule step for step.  The agent's policy follows when it IS the behaviour policy object
                                    (agent_policy = -1, as in the example).  Needs policy = RSRL_EPSILON_GREEDY, per-learner weights and a
                                    fused driver loop: the one-step agents and SARSALambda / QLambda on a register-family Fourier basis,
                                    or any one-step agent on tile coding / a generic Fourier order.
                                    Precision: the per-learner field is fp32 and eps * decay is rounded to fp32 once per episode, where the
                                    reference decays an f64 field: gen_bool's threshold (eps * 2^24, truncated) can differ from the f64
                                    schedule's by a few units after many episodes (relative 6e-8 per episode at most; ~1e-4 of a threshold of
                                    1.7e6 after 1 000 episodes) -- bit parity of the schedule is against the oracle's f32 instantiation, the f64
                                    oracle is matched to that tolerance (tests/test_gpu_round4.py). */
    double   epsilon_min;        /* floor of the schedule (0 = none) */
} rsrl_hip_config;
/* size of the ABI 3 struct: the oldest layout rsrl_hip_create accepts */
#define RSRL_HIP_CONFIG_SIZE_V3 ((uint32_t)offsetof(rsrl_hip_config, agent_policy))

/* per-call statistics of rsrl_hip_train (the println! / Response{error} of the
 * reference drivers, examples/q_learning.rs:54, control/td/q_learning.rs:17-20) */
typedef struct {
    uint64_t env_steps;            /* n_steps * n_envs                                        */
    uint64_t episodes;             /* episodes finished (terminal or step cap)                */
    uint64_t episodes_truncated;   /*   ... of which ended by the step cap                    */
    uint64_t sum_episode_steps;    /* sum of lengths of the finished episodes                 */
    double   sum_abs_td_error;     /* sum |delta|                                             */
    double   sum_reward;
} rsrl_hip_stats;

int         rsrl_hip_abi_version(void);
/* number of visible HIP devices (>= 0) or a negative status */
int         rsrl_hip_device_count(void);
const char* rsrl_hip_last_error(void);
/* README example values: MountainCar, Fourier(5), QLearning, gamma 0.9, SGD(0.001), Greedy */
int rsrl_hip_config_init(rsrl_hip_config* cfg);

int rsrl_hip_create(const rsrl_hip_config* cfg, rsrl_hip_ctx** out);
int rsrl_hip_destroy(rsrl_hip_ctx* ctx);
int rsrl_hip_sync(rsrl_hip_ctx* ctx);

/* Space queries: Domain::state_space().dim(), action_space().card(), basis.n_features()
 *   (examples/q_learning.rs:20; Parameterised::weights_dim, params/mod.rs:128) */
int rsrl_hip_state_dim(const rsrl_hip_ctx* ctx);
int rsrl_hip_n_actions(const rsrl_hip_ctx* ctx);
/* columns of the weight matrix: n_actions for the control agents (VectorLFA), 1 for TD / TDLambda (ScalarLFA,
 * Parameterised::weights_dim = (F, 1), fa/linear.rs:201-203) */
int rsrl_hip_n_outputs(const rsrl_hip_ctx* ctx);
int rsrl_hip_n_features(const rsrl_hip_ctx* ctx);
int64_t rsrl_hip_n_envs(const rsrl_hip_ctx* ctx);
/* state_space() bounds: mountain_car/discrete.rs:97-99, cart_pole.rs:112-118, acrobot.rs:143-149 */
int rsrl_hip_state_bounds(const rsrl_hip_ctx* ctx, double* lo /*[D]*/, double* hi /*[D]*/);

/* per-episode `Domain::default()` + initial `policy.sample(rng, env.emit().state())`
 *   examples/q_learning.rs:37-38 -- for every learner of the ctx */
int rsrl_hip_reset(rsrl_hip_ctx* ctx);

/* Domain::emit (state part)                         rsrl_domains/src/lib.rs:430 */
int rsrl_hip_get_states(rsrl_hip_ctx* ctx, float* states /*[D][N]*/);
int rsrl_hip_set_states(rsrl_hip_ctx* ctx, const float* states /*[D][N]*/);
/*   (set_states: the array must hold finite values within 1000 widths of each dimension's bounds, else EINVAL and the ctx is untouched -- the reference's
 *    wrap! macro, rsrl_domains/src/macros.rs:14-24, loops without end on an infinite angle; a HOST array is checked on the host, a DEVICE array on the
 *    device by the same rule (ABI 9: it used to be clamped silently, NaN passing)) */
int rsrl_hip_get_actions(rsrl_hip_ctx* ctx, int32_t* actions /*[N]*/);
int rsrl_hip_set_actions(rsrl_hip_ctx* ctx, const int32_t* actions /*[N]*/);
/* (ABI 8) The rest of a learner's state between two driver calls, so that a run can be carried into another ctx EXACTLY (with the checkpoint of
 * rsrl_hip_save_weights, the states and the actions above):
 *   - the steps every learner's current episode has taken (what the driver loop's `for j in 0..step_limit` counter would hold,
 *     rsrl/src/lib.rs:82-88 as the examples drive it): without it a resumed run counts its step cap from zero;
 *   - the register-family loops (one-step agents on MountainCar Fourier 1-5, CartPole / Acrobot Fourier 1) CARRY Q(s,.) of the current state from
 *     launch to launch as the fused loop left it (the pre-update value plus the rank-1 term of the last update).  rsrl_hip_set_states / _set_weights /
 *     _load_weights drop it and the next launch evaluates Q(s,.) from the weights: equal in exact arithmetic, not always in the last bit -- enough to
 *     part a Softmax / ExpectedSARSA run from the uninterrupted one.  get: *valid = 1 and q filled while something is carried, 0 (q untouched)
 *     otherwise -- always 0 for the families that evaluate Q from the weights every step; set (AFTER the states and weights are in place): the next
 *     launch continues from it.  EINVAL for a ctx whose kernels carry nothing. */
int rsrl_hip_get_episode_steps(rsrl_hip_ctx* ctx, uint32_t* steps /*[N]*/);
int rsrl_hip_set_episode_steps(rsrl_hip_ctx* ctx, const uint32_t* steps /*[N]*/);
int rsrl_hip_get_q_carry(rsrl_hip_ctx* ctx, float* q /*[A][N]*/, int32_t* valid);
int rsrl_hip_set_q_carry(rsrl_hip_ctx* ctx, const float* q /*[A][N]*/);

/* ---- The trait-granular loop (examples/q_learning.rs:40-52), one call per trait method, every learner of the ctx:
 *
 *        rsrl_hip_domain_step(ctx, act, from, to, rew, term);          t  = env.transition(a)
 *        rsrl_hip_handle(ctx, from, act, rew, to, term, n_envs, td);        agent.handle(&t)
 *        rsrl_hip_domain_reset(ctx, term);                                  terminal -> a new episode (q_learning.rs:37, :47-51)
 *        rsrl_hip_policy_sample(ctx, NULL, n_envs, act);               a' = policy.sample(rng, env.emit().state())
 *
 *   reproduces the reference's call pattern -- every action value evaluated afresh from the weights -- bit for bit against the CPU oracle's
 *   reference-order loop (tests/test_gpu_trait_loop.py), on every basis / agent with a handle.  A ctx created with steps_per_launch = 1 on a
 *   register-family Fourier basis (MountainCar order 1 / 3 / 5, CartPole / Acrobot order 1) with per-learner f32 weights and a one-step agent
 *   (QLearning, SARSA, ExpectedSARSA, PAL) keeps W learner-major and serves the loop as an HBM stream (rsrl_amd/csrc/kernels_trait.hpp): handle
 *   makes ONE pass over the learners' weights and hands Q(s',.) under the updated weights over to the sample that follows (a cache keyed by the
 *   state itself: a hit returns the bits a fresh evaluation returns).  With DEVICE arrays on a CTX-OWNED stream such a ctx ACCEPTS the four
 *   calls above and launches them as one kernel when they arrive in this order on the same arrays; any other call launches the accepted ones
 *   first, one kernel each -- same results, same order; as with rsrl_hip_train's coalescing, acceptance is not completion: rsrl_hip_sync (or
 *   any call that returns data to the host) completes them.  On a caller-supplied stream every call enqueues its own kernel before it returns.
 *   RSRL_NO_TRAIT_DEFER=1 / RSRL_NO_TRAIT_FAST=1 in the environment switch the deferral / the fast kernels off (A/B runs).  (ABI 9)
 *
 * Domain::transition                                rsrl_domains/src/lib.rs:436-446
 * Steps every env of the ctx with `actions` (NULL: the ctx's pending actions).  Outputs are
 * optional (NULL to skip).  The ctx's env state becomes s'; no auto-reset. */
int rsrl_hip_domain_step(rsrl_hip_ctx* ctx, const int32_t* actions,
                         float* from_states, float* next_states, float* rewards, uint8_t* terminal);
/* `MountainCar::default()` for the envs whose mask byte is non-zero (NULL: all) */
int rsrl_hip_domain_reset(rsrl_hip_ctx* ctx, const uint8_t* mask);

/* Function<(S,)>::evaluate for VectorLFA: Q(s,.) = W^T phi(s)      rsrl/src/fa/linear.rs:303-311
 * state m is evaluated with learner m's weights (or the shared weights). */
int rsrl_hip_q_evaluate(rsrl_hip_ctx* ctx, const float* states /*[D][M]*/, int64_t M, float* q_out /*[A][M]*/);
/*   prediction agents: the same call is Function<(S,)>::evaluate of the ScalarLFA (fa/linear.rs:213-221), q_out = V f32[1][M];
 *   find_max / policy_mode / policy_probs / policy_sample / rollout_greedy return RSRL_HIP_ESTATE for them (no Q function). */
/* Enumerable::find_max (ties -> last index)                         rsrl/src/core.rs:96-105 */
int rsrl_hip_q_find_max(rsrl_hip_ctx* ctx, const float* states, int64_t M, int32_t* idx_out, float* val_out);
/* Enumerable::find_min (the minimum; ties -> last index)            rsrl/src/core.rs:86-94 */
int rsrl_hip_q_find_min(rsrl_hip_ctx* ctx, const float* states, int64_t M, int32_t* idx_out, float* val_out);
/* Enumerable::expected_value(args, ps) = fold(0.0, acc + Q(s,a)*p_a) rsrl/src/core.rs:107-116; probs f32[A][M], out f32[M] */
int rsrl_hip_q_expected_value(rsrl_hip_ctx* ctx, const float* states, int64_t M, const float* probs, float* out);
/* basis.project(s): dense features phi f32[F][M] (Fourier) -- lfa Basis::project */
int rsrl_hip_project(rsrl_hip_ctx* ctx, const float* states, int64_t M, float* phi_out /*[F][M]*/);
/* tile coding: active indices int32[T][M] */
int rsrl_hip_tile_indices(rsrl_hip_ctx* ctx, const float* states, int64_t M, int32_t* idx_out /*[T][M]*/);

/* Handler<&Transition>::handle for QLearning / SARSA / ExpectedSARSA
 *   control/td/q_learning.rs:51-71, sarsa.rs:53-75, expected_sarsa.rs:45-66
 * td_error_out (optional) = Response.error (q_learning.rs:17-20).  Each call counts as one batch-step: it advances
 * rsrl_hip_step_count, the counter that addresses the agent-side random draws (SARSA's inner policy sample,
 * sarsa.rs:61; bf16 stochastic rounding), exactly as one step of rsrl_hip_train does.
 * With device arrays only the call is asynchronous on the ctx's stream (ABI 9; it used to synchronise); host arrays are staged and the call
 * returns when they have been consumed / filled. */
int rsrl_hip_handle(rsrl_hip_ctx* ctx, const float* from_states, const int32_t* actions,
                    const float* rewards, const float* to_states, const uint8_t* terminal,
                    int64_t M, float* td_error_out);

/* Policy::sample / Policy::mode / Function<(S,)> of the policy (action probabilities)
 *   policies/mod.rs:65-78; greedy.rs:30-44,77-83; epsilon_greedy.rs:38-45,74-82;
 *   softmax.rs:74-82,131-143; random.rs:19-48 (mode of Random: RSRL_HIP_EINVAL, random.rs:47 panics) */
/* states == NULL (ABI 9; M must be n_envs): policy.sample(rng, env.emit().state()) for the ctx's OWN envs -- the driver loop's behaviour sample
 *   (examples/q_learning.rs:38, :45).  It draws what batch-step rsrl_hip_step_count() - 1 of rsrl_hip_train draws (before the first handle: the initial
 *   sample's stream, as rsrl_hip_reset), and the actions also become the ctx's pending ones.  With explicit states the draws are a stream of their own,
 *   addressed by the number of such calls made on the ctx. */
int rsrl_hip_policy_sample(rsrl_hip_ctx* ctx, const float* states, int64_t M, int32_t* actions_out);
int rsrl_hip_policy_mode(rsrl_hip_ctx* ctx, const float* states, int64_t M, int32_t* actions_out);
int rsrl_hip_policy_probs(rsrl_hip_ctx* ctx, const float* states, int64_t M, float* probs_out /*[A][M]*/);
/* Function<(S, A)> of the policy: the probability of action a in state s for Greedy / EpsilonGreedy / Random
 *   (greedy.rs:46-60, epsilon_greedy.rs:49-63, random.rs:28-32) and -- faithfully -- the raw action value Q(s, a) for Softmax
 *   (softmax.rs:84-92 forwards to the approximator).  actions int32[M] in [0, A), prob_out f32[M]. */
int rsrl_hip_policy_prob(rsrl_hip_ctx* ctx, const float* states, const int32_t* actions, int64_t M, float* prob_out);
/* the pub field EpsilonGreedy.epsilon (decayed by drivers, examples/sarsa_lambda.rs:68); with config.epsilon_decay it sets every
 * learner's field */
int rsrl_hip_set_epsilon(rsrl_hip_ctx* ctx, double epsilon);
/* every learner's current epsilon, f32[N] (config.epsilon_decay: each learner's own field; otherwise N copies of the ctx's) */
int rsrl_hip_get_epsilons(rsrl_hip_ctx* ctx, float* eps_out /*[N]*/);

/* Parameterised::weights / weights_view_mut               rsrl/src/params/mod.rs:116-134
 * w is row-major f32[F][A] (ndarray Array2 (F, A), fa/linear.rs:293-301); env_index is
 * ignored in shared mode.  Also the checkpoint hook. */
int rsrl_hip_get_weights(rsrl_hip_ctx* ctx, int64_t env_index, float* w /*[F][A]*/);
int rsrl_hip_set_weights(rsrl_hip_ctx* ctx, int64_t env_index, const float* w /*[F][A]*/);
/* the pub field `trace` of SARSALambda / QLambda (sarsa_lambda.rs:41, q_lambda.rs:40): one learner's eligibility trace,
 * row-major f32[F][A] like the weights */
int rsrl_hip_get_traces(rsrl_hip_ctx* ctx, int64_t env_index, float* z /*[F][A]*/);
int rsrl_hip_set_traces(rsrl_hip_ctx* ctx, int64_t env_index, const float* z /*[F][A]*/);
/* the pub field `fa_td` of GreedyGQ (greedy_gq.rs:52): one learner's second approximator, row-major f32[F][A] */
int rsrl_hip_get_td_weights(rsrl_hip_ctx* ctx, int64_t env_index, float* v /*[F][A]*/);
int rsrl_hip_set_td_weights(rsrl_hip_ctx* ctx, int64_t env_index, const float* v /*[F][A]*/);
/* Checkpoint of the approximator(s) (the reference's only persistence story is the optional serde
 * derive on the agents, rsrl/Cargo.toml:26).  File format version 2 (3 for files that carry QSigma's backups, 5 for sparse traces), little-endian,
 * serialised field by field (no padding):
 *   offset  0  char magic[8] = "RSRLHIPW"
 *           8  u32  version = 2 (3 iff aux_kind = 3, 5 iff aux_kind = 4, 4 with the epsilon schedule)
 *          12  i32  domain, basis, order, n_tilings, tiles_per_dim, weight_mode, F, A (weight columns),
 *                   algo, weight_dtype, aux_kind (0 none, 1 eligibility traces, 2 GreedyGQ's fa_td weights,
 *                   3 QSigma's n-step backups, 4 sparse traces over a shared table)                             [11 x i32]
 *          56  i64  n_learners (1 in shared mode)
 *          64  u64  step_count
 *          72  n_learners x f32[F][A] weights in the reference's row-major (F, A) order (Parameterised::weights,
 *              params/mod.rs:118), independent of the device layout and storage dtype;
 *              then, if aux_kind is 1 or 2, n_learners x f32[F][A] of the auxiliary matrix (traces / fa_td);
 *              if aux_kind is 3 (file version 3): u32 head[N], u32 len[N], f32 entries[D + 5][n_steps][N] -- every learner's
 *              Backup ring {s, a, q, residual, pi, mu} (q_sigma.rs:30-63), so that a QSigma run with n_steps > 1 resumes
 *              bit-identically too.  Files of version 2 (no aux_kind 3) are still read.
 *              if aux_kind is 4 (file version 6; SARSALambda / QLambda over ONE shared tile-coded table): u64 n_envs, u64 env_offset (whose
 *              learners the lists belong to: a file of another shard is refused as a different configuration), u32 len[N], then for every
 *              learner in turn u32 key[len] (= feature index * A + action) and f32 value[len] -- its sparse trace (params/sparse.rs:13-97),
 *              the tilings' sub-lists one after the other, each in slot order, so that the run resumes bit-identically (the slot order decides
 *              which entry a full sub-list overwrites).  Version 5 (round 5: no n_envs / env_offset, one list per learner) is still read; its
 *              entries go to the sub-lists of their keys' tilings, and a file with more than 512 / n_tilings entries of one tiling is refused.
 *              A ctx with config.epsilon_decay writes file version 4: everything above, then f32 eps[N], every learner's current
 *              epsilon (the schedule's state), so that a resumed run continues the schedule.
 * load refuses a file whose header does not match the ctx's configuration or whose size is not exactly what the header
 * implies, and stages the data: a failing load leaves the ctx's weights untouched.  A loaded run's LEARNING resumes bit-identically (weights, traces,
 * backups, epsilons, step counter -- with the env states restored through rsrl_hip_set_states / _set_actions, the episodes' step counts through
 * rsrl_hip_set_episode_steps and, for the register-family loops, the carried Q(s,.) through rsrl_hip_set_q_carry: ABI 8); the evaluation-rollout draw
 * counter of rsrl_hip_rollout_policy is not part of the file (see there). */
int rsrl_hip_save_weights(rsrl_hip_ctx* ctx, const char* path);
int rsrl_hip_load_weights(rsrl_hip_ctx* ctx, const char* path);
/* same weights broadcast to every learner (per-env mode) */
int rsrl_hip_set_weights_all(rsrl_hip_ctx* ctx, const float* w /*[F][A]*/);

/* The fused driver loop (examples/q_learning.rs:40-52) x n_envs x n_steps with auto-reset
 * on terminal / step cap.  stats_out is a HOST pointer (optional). */
int rsrl_hip_train(rsrl_hip_ctx* ctx, int64_t n_steps, rsrl_hip_stats* stats_out);
/*   Without stats_out the call is ASYNCHRONOUS: it returns once the work is enqueued.  On a CTX-OWNED stream (config.stream
 *   NULL) short calls (a driver loop's 20 batch-steps) that arrive while the stream is still busy are coalesced -- held back and
 *   launched fuse-depth (4096 batch-steps) at a time, when anything observes or changes the ctx (every other entry point,
 *   rsrl_hip_sync included, flushes first), or when a call finds the stream idle.  Results are bit-identical to one launch per
 *   call (the fused loop carries Q(s,.) between launches and addresses the RNG by the batch-step); RSRL_NO_COALESCE=1 in the
 *   environment disables it.  On a caller-supplied stream nothing is ever held back.
 *   Shared dense weights (RSRL_W_SHARED on a register-family Fourier basis, at most one 512-learner block per CU): the whole
 *   call is ONE persistent launch; the ranks of a peer group exchange inside it (RSRL_NO_PERSIST=1: one launch per batch-step).
 *   That kernel needs every block of its grid -- and of its peers' grids -- resident at once.  The library guarantees it: the grid
 *   is checked against the device's occupancy (and once per ctx by a cooperative launch); a peer group decides COLLECTIVELY in
 *   rsrl_hip_peer_connect (ranks that share a device must fit together; RSRL_NO_PERSIST on any rank counts for all), so that no
 *   two ranks ever take different paths; unrelated persistent ctxs of one process take turns on a device.  Whenever the guarantee
 *   cannot be given the per-step path runs instead -- same results, bit for bit.
 *   Launch coalescing contract: steps a call has ACCEPTED but not launched (rsrl_hip_pending_steps) are launched by the next call
 *   on the ctx, whichever it is; a host that stops calling and wants them to run calls rsrl_hip_sync (as with any asynchronous
 *   queue, acceptance is not completion). */
/* batch-steps executed so far by rsrl_hip_train and rsrl_hip_handle (the RNG counter) */
uint64_t rsrl_hip_step_count(const rsrl_hip_ctx* ctx);
/* batch-steps rsrl_hip_train has accepted but not enqueued yet (launch coalescing); always 0 on a caller-supplied stream */
int64_t rsrl_hip_pending_steps(const rsrl_hip_ctx* ctx);

/* Domain::rollout with the closure s -> policy.mode(s) and Some(step_limit), + n_states
 *   rsrl_domains/src/lib.rs:448-479, :340; one fresh default env per learner, the ctx's
 *   training envs are untouched.  step_limit >= 1.
 *   step_limit == 0 (all three rollout calls): Domain::rollout(.., None), lib.rs:469-476 -- no limit, the trajectory ends at the first
 *   Terminal observation.  A device loop needs a bound, so config.max_episode_steps (> 0, else RSRL_HIP_EINVAL) caps it at that many
 *   transitions: outputs are sized as for step_limit = max_episode_steps + 1, and a trajectory that terminates within the cap
 *   (terminal_out = 1) is exactly the reference's unbounded one.  (ABI 7) */
int rsrl_hip_rollout_greedy(rsrl_hip_ctx* ctx, int64_t step_limit,
                            uint32_t* n_states_out /*[N]*/, float* total_reward_out /*[N]*/);
/* The same rollout for learners 0..M-1 with the Trajectory itself (rsrl_domains/src/lib.rs:334-409): every output but
 * n_states_out is optional.
 *   states_out   f32[step_limit][D][M]     row 0 = Trajectory.start, row k = the observation of steps[k-1]
 *   actions_out  i32[step_limit-1][M]      steps[k].1        rewards_out f32[step_limit-1][M]   steps[k].2
 *   terminal_out u8[M]                     the last observation is Observation::Terminal
 * Rows past a learner's n_states are zero (host buffers) / untouched (device buffers).  Trajectory::total_reward (:391) =
 * total_reward_out, n_states (:340) = n_states_out, n_transitions = n_states - 1. */
int rsrl_hip_rollout_trajectory(rsrl_hip_ctx* ctx, int64_t step_limit, int64_t M, uint32_t* n_states_out, float* total_reward_out,
                                float* states_out, int32_t* actions_out, float* rewards_out, uint8_t* terminal_out);

/* Domain::rollout with ANY of the four policies as the closure, s -> policy.sample(rng, s)  (lib.rs:448-479 takes any
 * FnMut(&S) -> A; policies/mod.rs:65-78): an epsilon-greedy or softmax evaluation run next to the greedy one.  `policy` is an
 * rsrl_policy over the ctx's Q function with its own parameters (epsilon for RSRL_EPSILON_GREEDY, tau for RSRL_SOFTMAX; the ctx's
 * behaviour policy is not touched).  Outputs as rsrl_hip_rollout_trajectory (every one but n_states_out optional).  The draws
 * are a stream of their own, addressed by (number of rollout_policy calls made on the ctx so far, action selection k, global
 * learner id): a call is reproducible, successive calls are independent samples.  The call counter belongs to the ctx OBJECT: it starts
 * at 0 when the ctx is created and is neither saved by rsrl_hip_save_weights nor changed by rsrl_hip_load_weights / rsrl_hip_reset -- the n-th
 * evaluation rollout of a process that restored a checkpoint draws what the n-th rollout of any fresh ctx draws (training draws are keyed by
 * the checkpointed step counter; evaluation draws are not part of the learning state). */
int rsrl_hip_rollout_policy(rsrl_hip_ctx* ctx, int policy, double epsilon, double tau, int64_t step_limit, int64_t M,
                            uint32_t* n_states_out, float* total_reward_out, float* states_out, int32_t* actions_out,
                            float* rewards_out, uint8_t* terminal_out);

/* Order-independent 64-bit checksums of the ctx's device state (sum of the 32-bit words, each multiplied by an odd
 * function of its index): out[0] weights (+traces), out[1] env states/actions/episode counters.  For determinism /
 * sharding / fusion-invariance checks at sizes where copying the weights out is not practical. */
int rsrl_hip_checksum(rsrl_hip_ctx* ctx, uint64_t out[2]);

/* Shared weights: every cross-learner sum is 64-bit fixed point (lsb = 2^(floor(log2 lr) - 28)); a term or block sum beyond
 * +-2^42 lsb (|lr*e*phi| > 16384 * 2^floor(log2 lr): a diverged learner) is CLAMPED, and counted here -- terms clamped so far by
 * the shared-W kernels on this ctx's device, all ctxs of the process included (0 in every healthy run: the update then is
 * exactly W += fl(sum of the quantised terms)). */
int rsrl_hip_fx_saturations(rsrl_hip_ctx* ctx, uint64_t* count_out);

/* ---- multi-GPU (one process per GPU; no reference counterpart) -------------------------
 * Shared-W mode across ranks: every batch-step all-reduces the (F x A) f32 weight delta
 * over RCCL.  id_bytes is an ncclUniqueId (128 bytes) produced on rank 0 and distributed
 * by the caller's control plane (torch.distributed / MPI / files). */
int rsrl_hip_comm_unique_id(uint8_t* id_bytes /*[128]*/);
int rsrl_hip_comm_init(rsrl_hip_ctx* ctx, const uint8_t* id_bytes, int world_size, int rank);
/* A communicator of size 1 is valid and runs the SAME finalize -> all-reduce -> apply sequence as world_size > 1.
 *
 * RSRL_EXCHANGE_PEER (config.exchange): the one-hop peer-write exchange instead of RCCL.  Every rank calls
 * rsrl_hip_peer_export (allocates its receive buffer for world_size ranks and describes it in a 128-byte handle: an
 * hipIpcMemHandle plus the owner's pid), the caller's control plane all-gathers the handles, then every rank calls
 * rsrl_hip_peer_connect with all of them in rank order.  Ranks may live in different processes (one per GPU; the
 * buffers are mapped with hipIpcOpenMemHandle) or in one process (several ctxs, any devices: peer access between the
 * devices is enabled by the call, RSRL_HIP_EINVAL if the hardware cannot).  A rank that waits more than 4 s
 * (RSRL_PEER_TIMEOUT_MS in the environment) for a peer's delta gives up: that update is NOT applied as a partial sum (the
 * persistent kernel skips it and ends; the per-step exchange kernels poison the weights with NaN), and the next call that
 * synchronises (rsrl_hip_sync, get_weights / get_states / checksum into host memory, save_weights) returns RSRL_HIP_ERCCL.
 * Slot parity and granule tags follow the number of exchanges performed, not the batch-step counter: a restored checkpoint
 * (rsrl_hip_load_weights sets the counter back) cannot make a stale slot look current. */
#define RSRL_HIP_PEER_HANDLE_BYTES 128
int rsrl_hip_peer_export(rsrl_hip_ctx* ctx, int world_size, uint8_t* handle_out /*[128]*/);
/* 1 if `device` can read and write `peer_device`'s memory directly (hipDeviceCanAccessPeer; a device can always access itself), 0 if not,
 * a negative status on error: what a host needs to choose RSRL_EXCHANGE_PEER for its ranks (every pair must say 1) */
int rsrl_hip_can_access_peer(int device, int peer_device);
/* Which PHYSICAL device is ordinal `device` of this process: PCI domain << 32 | bus << 16 | device (bit 63 set), the same number in every
 * process of a node whatever HIP_VISIBLE_DEVICES it runs under.  With the host's identity it is what a multi-process launcher all-gathers to
 * decide RSRL_EXCHANGE_AUTO the same way on every rank: the peer exchange only if all ranks share ONE host and every rank sees, and can access,
 * every other rank's device (rsrl_amd/distributed.py: choose_exchange); RCCL otherwise.  (ABI 7) */
int rsrl_hip_device_identity(int device, uint64_t* identity_out);
int rsrl_hip_peer_connect(rsrl_hip_ctx* ctx, const uint8_t* handles /*[world_size][128]*/, int world_size, int rank);

/* All ranks in ONE process (a single-threaded host, like the reference's Rc<RefCell> owner graph, rsrl/src/core.rs:13-15):
 * attaches the exchange configured in the ctxs (all alike) to ctxs[0..n), rank = index.  PEER: export + connect of every
 * ctx; RCCL: ncclCommInitAll over the ctxs' devices (one device per rank) plus a grouped warm-up all-reduce, so that no
 * later call blocks on a rank the same thread has not driven yet.  Afterwards the host calls rsrl_hip_train(ctxs[i], ..)
 * for each i in turn (the calls only enqueue) and rsrl_hip_sync(ctxs[i]) at the end. */
int rsrl_hip_group_create(rsrl_hip_ctx* const* ctxs, int n);
/* Step every rank of such a group from that one thread: n_steps batch-steps on ctxs[0..n) (exactly the group's ranks, in rank
 * order); results are those of rsrl_hip_train on every rank from a thread of its own, bit for bit.
 *   RCCL: REQUIRED for n > 1 -- a thread driving several communicators must issue each collective for all of them inside one
 *         ncclGroupStart / End, so the ranks advance in lock-step here and rsrl_hip_train / rsrl_hip_handle on a rank of such a group
 *         return RSRL_HIP_ESTATE.
 *   PEER: feeds the ranks in turns of at most 32 batch-steps, so that no rank's launch queue fills up in front of a peer whose
 *         kernels the same thread has not enqueued yet (calling rsrl_hip_train rank by rank stays valid for short calls). */
int rsrl_hip_group_train(rsrl_hip_ctx* const* ctxs, int n, int64_t n_steps);
/* what is attached: world size and rank as the exchange itself reports them (ncclCommCount / ncclCommUserRank for RCCL),
 * exchange = -1 (none), RSRL_EXCHANGE_RCCL or RSRL_EXCHANGE_PEER.  Any output pointer may be NULL. */
int rsrl_hip_comm_info(rsrl_hip_ctx* ctx, int* world_size, int* rank, int* exchange);

/* ---- measurement hooks (bench.py) --------------------------------------------------------
 * HIP-event timing of the kernels launched by train since the last reset, on the ctx's
 * stream.  ms_total / launches = average launch duration of the dominant kernel. */
int rsrl_hip_timing_enable(rsrl_hip_ctx* ctx, int enable);
int rsrl_hip_timing_read(rsrl_hip_ctx* ctx, double* ms_total, uint64_t* launches, const char** kernel_name);
/* (ABI 9) what this box's memory system delivers: a float4 device-to-device copy of `bytes` bytes (rounded down to 16), `reps` times, by HIP events ->
 * GB/s counting read + write.  No ctx: allocates and frees its own two buffers.  bench.py quotes every HBM fraction against it next to the published peak. */
int rsrl_hip_measure_copy(int device, size_t bytes, int reps, double* gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* RSRL_HIP_H */
