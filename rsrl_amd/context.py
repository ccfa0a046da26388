"""`Context`: one (domain, basis, algo, policy, N, W-mode) instance on one MI355X -- a thin numpy-facing
wrapper over the C ABI (include/rsrl_hip.h).  All compute happens in librsrl_hip.so's HIP kernels."""
import ctypes as C

import numpy as np

from . import _abi

MOUNTAIN_CAR, CART_POLE, ACROBOT = 0, 1, 2
FOURIER, TILE_CODING = 0, 1
QLEARNING, SARSA, EXPECTED_SARSA, SARSA_LAMBDA, Q_LAMBDA, PAL, GREEDY_GQ, TD, TD_LAMBDA, Q_SIGMA = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
TRACE_ACCUMULATE, TRACE_SATURATE, TRACE_DUTCH = 0, 1, 2
GREEDY, EPSILON_GREEDY, SOFTMAX, RANDOM = 0, 1, 2, 3
W_PER_ENV, W_SHARED = 0, 1
W_F32, W_BF16 = 0, 1
EXCHANGE_RCCL, EXCHANGE_PEER, EXCHANGE_AUTO = 0, 1, 2
PEER_HANDLE_BYTES = 128


def _p(a):
    """numpy array / raw device pointer (int) / None -> void*"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _in(a, dtype, shape=None):
    if a is None or isinstance(a, int):
        return a
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def can_access_peer(device, peer_device):
    """True if `device` reads and writes `peer_device`'s memory directly (what the one-hop peer exchange needs of every pair of ranks)"""
    rc = _abi.lib().rsrl_hip_can_access_peer(int(device), int(peer_device))
    if rc < 0:
        _abi.check(rc)
    return rc == 1


def device_identity(device):
    """which PHYSICAL device ordinal `device` is (PCI domain : bus : device): the same number in every process of the node, whatever
    HIP_VISIBLE_DEVICES each runs under"""
    out = C.c_uint64()
    _abi.check(_abi.lib().rsrl_hip_device_identity(int(device), C.byref(out)))
    return int(out.value)


def device_count():
    """visible HIP devices (raises RsrlHipError when the runtime finds none)"""
    n = _abi.lib().rsrl_hip_device_count()
    if n < 0:
        _abi.check(n)
    return n


class Context:
    """One object graph of the reference on one GPU, times `n_envs` independent environments in lock-step:

        env    = MountainCar | CartPole | Acrobot ::default()                      (domain)
        basis  = Fourier::from_space(order, ..).with_bias() | TileCoding           (basis, order | n_tilings, tiles_per_dim)
        q_func = make_shared(LFA::vector(basis, SGD(lr), n_actions))               (lr; weight_mode: one per env or one shared)
        policy = Greedy | EpsilonGreedy(epsilon) | Softmax(tau) | Random           (policy, epsilon, tau)
        agent  = QLearning{gamma} | SARSA | ExpectedSARSA{alpha} | PAL{alpha}      (algo, gamma, alpha)
                 | SARSALambda / QLambda {trace(gamma, lam), alpha}                (lam, trace)
                 | GreedyGQ {fa_td = LFA(SGD(lr_td))} | TD | TDLambda              (lr_td; TD/TDLambda need policy=RANDOM)
                 | QSigma {alpha, gamma, sigma, n_steps}                            (sigma, n_steps)
                 SARSA / ExpectedSARSA / SARSALambda own a policy of their own       (agent_policy, agent_epsilon, agent_tau;
                                                                                     None = the behaviour policy object)

    (rsrl/examples/q_learning.rs:19-32 and the other examples).  `seed` keys the per-env Philox streams by GLOBAL env id
    (`env_offset` + local index), so a sharded run reproduces the unsharded one bit for bit.  `steps_per_launch`: fuse
    depth of `train` (0 = library default: 4096 for the register-resident loops, 256 otherwise; 1 = one batch-step per launch).  Arrays are SoA: states `(D, M)`, actions `(M,)`; weights
    `(F, n_out)` row-major like the reference's `Parameterised::weights()`.  Every method raises `RsrlHipError` with the
    ABI's message on failure; there is no CPU fallback."""

    def __init__(self, domain=MOUNTAIN_CAR, basis=FOURIER, order=5, n_tilings=8, tiles_per_dim=8,
                 algo=QLEARNING, policy=GREEDY, weight_mode=W_PER_ENV, weight_dtype=W_F32,
                 n_envs=1, env_offset=0, seed=0, gamma=0.9, lr=0.001, alpha=1.0, epsilon=0.1, tau=1.0,
                 max_episode_steps=0, steps_per_launch=0, device=0, stream=None, lam=0.0, trace=TRACE_ACCUMULATE, lr_td=0.0,
                 agent_policy=None, agent_epsilon=0.1, agent_tau=1.0, exchange=EXCHANGE_AUTO, sigma=0.0, n_steps=1, peer_timeout_ms=0,
                 epsilon_decay=1.0, epsilon_min=0.0):
        self._L = _abi.lib()
        cfg = _abi.Config()
        _abi.check(self._L.rsrl_hip_config_init(C.byref(cfg)))
        cfg.device, cfg.domain, cfg.basis, cfg.order = device, domain, basis, order
        cfg.n_tilings, cfg.tiles_per_dim = n_tilings, tiles_per_dim
        cfg.algo, cfg.policy, cfg.weight_mode, cfg.weight_dtype = algo, policy, weight_mode, weight_dtype
        cfg.max_episode_steps, cfg.n_envs, cfg.env_offset, cfg.seed = max_episode_steps, n_envs, env_offset, seed
        cfg.gamma, cfg.lr, cfg.alpha, cfg.epsilon, cfg.tau = gamma, lr, alpha, epsilon, tau
        cfg.steps_per_launch = steps_per_launch
        cfg.lam, cfg.trace, cfg.lr_td = lam, trace, lr_td
        cfg.stream = stream
        # the policy owned by the agent (SARSA's inner draw, ExpectedSARSA's expectation); None = the behaviour policy itself
        cfg.agent_policy = -1 if agent_policy is None else int(agent_policy)
        cfg.agent_epsilon, cfg.agent_tau, cfg.exchange = agent_epsilon, agent_tau, exchange
        cfg.sigma, cfg.n_steps = sigma, n_steps                   # QSigma{sigma, Backup::new(n_steps)}
        cfg.peer_timeout_ms = int(peer_timeout_ms)                # bound of the in-kernel waits of the shared-W exchange (0 = default)
        # the reference drivers' schedule `agent.policy.epsilon *= 0.995` once per episode of a learner (examples/sarsa_lambda.rs:68)
        cfg.epsilon_decay, cfg.epsilon_min = float(epsilon_decay), float(epsilon_min)
        self.cfg = cfg
        self._h = C.c_void_p()
        _abi.check(self._L.rsrl_hip_create(C.byref(cfg), C.byref(self._h)))
        self.N = int(n_envs)
        self.D = self._L.rsrl_hip_state_dim(self._h)
        self.A = self._L.rsrl_hip_n_actions(self._h)
        self.n_out = self._L.rsrl_hip_n_outputs(self._h)      # weight columns: A, or 1 for TD / TDLambda
        self.F = self._L.rsrl_hip_n_features(self._h)
        self.shared = weight_mode == W_SHARED

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.rsrl_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        _abi.check(self._L.rsrl_hip_sync(self._h))

    # ---- spaces
    def state_bounds(self):
        lo, hi = (C.c_double * self.D)(), (C.c_double * self.D)()
        _abi.check(self._L.rsrl_hip_state_bounds(self._h, lo, hi))
        return np.array(lo[:]), np.array(hi[:])

    # ---- env state
    def reset(self):
        _abi.check(self._L.rsrl_hip_reset(self._h))

    @property
    def states(self):
        out = np.empty((self.D, self.N), dtype=np.float32)
        _abi.check(self._L.rsrl_hip_get_states(self._h, _p(out)))
        return out

    @states.setter
    def states(self, v):
        _abi.check(self._L.rsrl_hip_set_states(self._h, _p(_in(v, np.float32, (self.D, self.N)))))

    @property
    def actions(self):
        out = np.empty(self.N, dtype=np.int32)
        _abi.check(self._L.rsrl_hip_get_actions(self._h, _p(out)))
        return out

    @actions.setter
    def actions(self, v):
        _abi.check(self._L.rsrl_hip_set_actions(self._h, _p(_in(v, np.int32, (self.N,)))))

    @property
    def episode_steps(self):
        """steps every learner's current episode has taken (ABI 8)"""
        out = np.empty(self.N, dtype=np.uint32)
        _abi.check(self._L.rsrl_hip_get_episode_steps(self._h, _p(out)))
        return out

    @episode_steps.setter
    def episode_steps(self, v):
        _abi.check(self._L.rsrl_hip_set_episode_steps(self._h, _p(_in(v, np.uint32, (self.N,)))))

    @property
    def q_carry(self):
        """Q(s,.) as the register-family loops carry it from launch to launch, (A, N) -- None while nothing is carried (ABI 8)"""
        out = np.empty((self.A, self.N), dtype=np.float32)
        valid = C.c_int32(0)
        _abi.check(self._L.rsrl_hip_get_q_carry(self._h, _p(out), C.byref(valid)))
        return out if valid.value else None

    @q_carry.setter
    def q_carry(self, v):
        _abi.check(self._L.rsrl_hip_set_q_carry(self._h, _p(_in(v, np.float32, (self.A, self.N)))))

    def domain_step(self, actions=None):
        """Domain::transition for every env -> (from, next, reward, terminal)"""
        frm = np.empty((self.D, self.N), dtype=np.float32)
        nxt = np.empty((self.D, self.N), dtype=np.float32)
        rew = np.empty(self.N, dtype=np.float32)
        term = np.empty(self.N, dtype=np.uint8)
        _abi.check(self._L.rsrl_hip_domain_step(self._h, _p(_in(actions, np.int32, (self.N,))), _p(frm), _p(nxt),
                                                _p(rew), _p(term)))
        return frm, nxt, rew, term

    def domain_reset(self, mask=None):
        _abi.check(self._L.rsrl_hip_domain_reset(self._h, _p(_in(mask, np.uint8, (self.N,)))))

    # ---- Q function
    def _batch(self, states):
        states = _in(states, np.float32)
        if states.ndim != 2 or states.shape[0] != self.D:
            raise ValueError(f"states must be [D={self.D}][M]")
        return states, states.shape[1]

    def q_evaluate(self, states):
        states, M = self._batch(states)
        out = np.empty((self.n_out, M), dtype=np.float32)
        _abi.check(self._L.rsrl_hip_q_evaluate(self._h, _p(states), M, _p(out)))
        return out

    def q_find_max(self, states):
        states, M = self._batch(states)
        idx, val = np.empty(M, dtype=np.int32), np.empty(M, dtype=np.float32)
        _abi.check(self._L.rsrl_hip_q_find_max(self._h, _p(states), M, _p(idx), _p(val)))
        return idx, val

    def q_find_min(self, states):
        """Enumerable::find_min (core.rs:86-94): (index, value), ties -> last index"""
        states, M = self._batch(states)
        idx, val = np.empty(M, dtype=np.int32), np.empty(M, dtype=np.float32)
        _abi.check(self._L.rsrl_hip_q_find_min(self._h, _p(states), M, _p(idx), _p(val)))
        return idx, val

    def q_expected_value(self, states, probs):
        """Enumerable::expected_value (core.rs:107-116): sum_a Q(s, a) * probs[a]; probs (A, M)"""
        states, M = self._batch(states)
        out = np.empty(M, dtype=np.float32)
        _abi.check(self._L.rsrl_hip_q_expected_value(self._h, _p(states), M, _p(_in(probs, np.float32, (self.A, M))), _p(out)))
        return out

    def project(self, states):
        states, M = self._batch(states)
        out = np.empty((self.F, M), dtype=np.float32)
        _abi.check(self._L.rsrl_hip_project(self._h, _p(states), M, _p(out)))
        return out

    def tile_indices(self, states):
        states, M = self._batch(states)
        out = np.empty((self.cfg.n_tilings, M), dtype=np.int32)
        _abi.check(self._L.rsrl_hip_tile_indices(self._h, _p(states), M, _p(out)))
        return out

    # ---- agent / policy
    def handle(self, from_states, actions, rewards, to_states, terminal):
        from_states, M = self._batch(from_states)
        to_states, _ = self._batch(to_states)
        td = np.empty(M, dtype=np.float32)
        _abi.check(self._L.rsrl_hip_handle(self._h, _p(from_states), _p(_in(actions, np.int32, (M,))),
                                           _p(_in(rewards, np.float32, (M,))), _p(to_states),
                                           _p(_in(terminal, np.uint8, (M,))), M, _p(td)))
        return td

    def policy_sample(self, states=None):
        """Policy::sample.  states = None: the ctx's OWN envs (env.emit().state()) -- the driver loop's behaviour sample of the current batch-step
        (what rsrl_hip_train draws); the actions also become the ctx's pending ones"""
        if states is None:
            out = np.empty(self.N, dtype=np.int32)
            _abi.check(self._L.rsrl_hip_policy_sample(self._h, None, self.N, _p(out)))
            return out
        states, M = self._batch(states)
        out = np.empty(M, dtype=np.int32)
        _abi.check(self._L.rsrl_hip_policy_sample(self._h, _p(states), M, _p(out)))
        return out

    def policy_mode(self, states):
        states, M = self._batch(states)
        out = np.empty(M, dtype=np.int32)
        _abi.check(self._L.rsrl_hip_policy_mode(self._h, _p(states), M, _p(out)))
        return out

    def policy_probs(self, states):
        states, M = self._batch(states)
        out = np.empty((self.A, M), dtype=np.float32)
        _abi.check(self._L.rsrl_hip_policy_probs(self._h, _p(states), M, _p(out)))
        return out

    def policy_prob(self, states, actions):
        """Function<(S, A)> of the policy: P(a | s) (Softmax: the raw Q(s, a), softmax.rs:84-92)"""
        states, M = self._batch(states)
        out = np.empty(M, dtype=np.float32)
        _abi.check(self._L.rsrl_hip_policy_prob(self._h, _p(states), _p(_in(actions, np.int32, (M,))), M, _p(out)))
        return out

    def set_epsilon(self, eps):
        _abi.check(self._L.rsrl_hip_set_epsilon(self._h, float(eps)))

    @property
    def epsilons(self):
        """every learner's current EpsilonGreedy.epsilon (N,): its own field under `epsilon_decay`, else N copies of the ctx's"""
        out = np.empty(self.N, dtype=np.float32)
        _abi.check(self._L.rsrl_hip_get_epsilons(self._h, _p(out)))
        return out

    # ---- Parameterised
    def get_weights(self, env_index=0):
        out = np.empty((self.F, self.n_out), dtype=np.float32)
        _abi.check(self._L.rsrl_hip_get_weights(self._h, int(env_index), _p(out)))
        return out

    def set_weights(self, w, env_index=0):
        _abi.check(self._L.rsrl_hip_set_weights(self._h, int(env_index), _p(_in(w, np.float32, (self.F, self.n_out)))))

    def get_traces(self, env_index=0):
        out = np.empty((self.F, self.n_out), dtype=np.float32)
        _abi.check(self._L.rsrl_hip_get_traces(self._h, int(env_index), _p(out)))
        return out

    def set_traces(self, z, env_index=0):
        _abi.check(self._L.rsrl_hip_set_traces(self._h, int(env_index), _p(_in(z, np.float32, (self.F, self.n_out)))))

    def get_td_weights(self, env_index=0):
        """GreedyGQ.fa_td's weights of one learner (greedy_gq.rs:52)"""
        out = np.empty((self.F, self.A), dtype=np.float32)
        _abi.check(self._L.rsrl_hip_get_td_weights(self._h, int(env_index), _p(out)))
        return out

    def set_td_weights(self, v, env_index=0):
        _abi.check(self._L.rsrl_hip_set_td_weights(self._h, int(env_index), _p(_in(v, np.float32, (self.F, self.A)))))

    def save_weights(self, path):
        _abi.check(self._L.rsrl_hip_save_weights(self._h, str(path).encode()))

    def load_weights(self, path):
        _abi.check(self._L.rsrl_hip_load_weights(self._h, str(path).encode()))

    def set_weights_all(self, w):
        _abi.check(self._L.rsrl_hip_set_weights_all(self._h, _p(_in(w, np.float32, (self.F, self.n_out)))))

    # ---- driver loop
    def train(self, n_steps, want_stats=True):
        st = _abi.Stats()
        _abi.check(self._L.rsrl_hip_train(self._h, int(n_steps), C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    @property
    def step_count(self):
        return int(self._L.rsrl_hip_step_count(self._h))

    @property
    def pending_steps(self):
        """batch-steps accepted by train() but not enqueued yet (launch coalescing on a ctx-owned stream)"""
        return int(self._L.rsrl_hip_pending_steps(self._h))

    def _rollout_rows(self, step_limit):
        """rows of a rollout's state output: step_limit, or -- step_limit 0 / None = Domain::rollout(.., None), bounded by the ctx's
        max_episode_steps -- max_episode_steps + 1"""
        return int(step_limit) if step_limit else int(self.cfg.max_episode_steps) + 1

    def rollout_greedy(self, step_limit):
        n_states = np.empty(self.N, dtype=np.uint32)
        tot = np.empty(self.N, dtype=np.float32)
        _abi.check(self._L.rsrl_hip_rollout_greedy(self._h, int(step_limit or 0), _p(n_states), _p(tot)))
        return n_states, tot

    def rollout_trajectory(self, step_limit, M=None):
        """Domain::rollout with the Trajectory (lib.rs:334-409) of learners 0..M-1 -> dict(n_states, total_reward, states
        (step_limit, D, M), actions / rewards (step_limit - 1, M), terminal (M,)); rows past n_states are zero"""
        M = self.N if M is None else int(M)
        L = self._rollout_rows(step_limit)
        out = dict(n_states=np.empty(M, dtype=np.uint32), total_reward=np.empty(M, dtype=np.float32),
                   states=np.zeros((L, self.D, M), dtype=np.float32), actions=np.zeros((max(L - 1, 0), M), dtype=np.int32),
                   rewards=np.zeros((max(L - 1, 0), M), dtype=np.float32), terminal=np.empty(M, dtype=np.uint8))
        _abi.check(self._L.rsrl_hip_rollout_trajectory(self._h, int(step_limit or 0), M, _p(out["n_states"]), _p(out["total_reward"]), _p(out["states"]),
                                                       _p(out["actions"]) if L > 1 else None, _p(out["rewards"]) if L > 1 else None,
                                                       _p(out["terminal"])))
        return out

    def rollout_policy(self, policy, step_limit, M=None, epsilon=0.1, tau=1.0):
        """Domain::rollout with the closure s -> policy.sample(rng, s) for ANY of the four policies over the ctx's Q function
        (lib.rs:448-479 takes any FnMut(&S) -> A) -> the same dict as rollout_trajectory"""
        M = self.N if M is None else int(M)
        L = self._rollout_rows(step_limit)
        out = dict(n_states=np.empty(M, dtype=np.uint32), total_reward=np.empty(M, dtype=np.float32),
                   states=np.zeros((L, self.D, M), dtype=np.float32), actions=np.zeros((max(L - 1, 0), M), dtype=np.int32),
                   rewards=np.zeros((max(L - 1, 0), M), dtype=np.float32), terminal=np.empty(M, dtype=np.uint8))
        _abi.check(self._L.rsrl_hip_rollout_policy(self._h, int(policy), float(epsilon), float(tau), int(step_limit or 0), M, _p(out["n_states"]),
                                                   _p(out["total_reward"]), _p(out["states"]), _p(out["actions"]) if L > 1 else None,
                                                   _p(out["rewards"]) if L > 1 else None, _p(out["terminal"])))
        return out

    def checksum(self):
        """(weights(+traces), env state) 64-bit checksums computed on the device"""
        out = (C.c_uint64 * 2)()
        _abi.check(self._L.rsrl_hip_checksum(self._h, out))
        return int(out[0]), int(out[1])

    def fx_saturations(self):
        """terms the shared-W fixed-point sums had to clamp so far on this device (0 in a healthy run)"""
        n = C.c_uint64()
        _abi.check(self._L.rsrl_hip_fx_saturations(self._h, C.byref(n)))
        return int(n.value)

    # ---- multi-GPU (shared weights): RCCL communicator, one process per GPU
    @staticmethod
    def comm_unique_id():
        """ncclUniqueId (128 bytes) -- call on rank 0 and distribute through the control plane"""
        buf = (C.c_uint8 * 128)()
        _abi.check(_abi.lib().rsrl_hip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, id_bytes, world_size, rank):
        buf = (C.c_uint8 * 128)(*id_bytes)
        _abi.check(self._L.rsrl_hip_comm_init(self._h, buf, int(world_size), int(rank)))

    # ---- multi-GPU (shared weights): one-hop peer-write exchange (exchange=EXCHANGE_PEER)
    def peer_export(self, world_size):
        """this rank's receive buffer described in a 128-byte handle -- all-gather the handles through the control plane"""
        buf = (C.c_uint8 * PEER_HANDLE_BYTES)()
        _abi.check(self._L.rsrl_hip_peer_export(self._h, int(world_size), buf))
        return bytes(buf)

    def peer_connect(self, handles, rank):
        """handles: the world_size handles in rank order"""
        blob = b"".join(handles)
        buf = (C.c_uint8 * len(blob))(*blob)
        _abi.check(self._L.rsrl_hip_peer_connect(self._h, buf, len(handles), int(rank)))

    @staticmethod
    def group_create(ctxs):
        """all ranks in this process (rank = position in `ctxs`): attach the exchange the ctxs were configured with"""
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        _abi.check(_abi.lib().rsrl_hip_group_create(arr, len(ctxs)))

    @staticmethod
    def group_train(ctxs, n_steps):
        """one thread stepping every rank of a group made by group_create (required for RCCL groups of more than one rank:
        each batch-step's all-reduces are issued for all ranks inside one ncclGroupStart / End)"""
        arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        _abi.check(_abi.lib().rsrl_hip_group_train(arr, len(ctxs), int(n_steps)))

    def comm_info(self):
        """(world_size, rank, exchange) as the attached exchange reports them; exchange -1 = none"""
        w, r, e = C.c_int(), C.c_int(), C.c_int()
        _abi.check(self._L.rsrl_hip_comm_info(self._h, C.byref(w), C.byref(r), C.byref(e)))
        return w.value, r.value, e.value

    # ---- measurement
    def timing_enable(self, on=True):
        _abi.check(self._L.rsrl_hip_timing_enable(self._h, int(bool(on))))

    def timing_read(self):
        ms, n, name = C.c_double(), C.c_uint64(), C.c_char_p()
        _abi.check(self._L.rsrl_hip_timing_read(self._h, C.byref(ms), C.byref(n), C.byref(name)))
        return ms.value, int(n.value), (name.value or b"").decode()
