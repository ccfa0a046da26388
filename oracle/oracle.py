"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Precisions: "f64" (the reference's), "f32" (the device's arithmetic type and op order, libm transcendentals rounded
once) and "f32d" (f32 with the device's own sincos / exp polynomials restated: bitwise comparison with the HIP path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  See oracle/rsrl_oracle.c for scope and pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MOUNTAIN_CAR, CART_POLE, ACROBOT = 0, 1, 2
FOURIER, TILE = 0, 1
QLEARNING, SARSA, EXPECTED_SARSA, SARSA_LAMBDA, Q_LAMBDA, PAL, GREEDY_GQ, TD, TD_LAMBDA, Q_SIGMA = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
TRACE_ACCUMULATE, TRACE_SATURATE, TRACE_DUTCH = 0, 1, 2
GREEDY, EGREEDY, SOFTMAX, RANDOM = 0, 1, 2, 3
BLK_STEP, BLK_RESET, BLK_INNER, BLK_INIT, BLK_API = 0, 1, 2, 3, 4


class Basis(C.Structure):
    _fields_ = [("kind", C.c_int), ("dim", C.c_int), ("order", C.c_int),
                ("n_tilings", C.c_int), ("tiles_per_dim", C.c_int),
                ("lo", C.c_double * 8), ("hi", C.c_double * 8),
                ("lo_f", C.c_float * 8), ("hi_f", C.c_float * 8)]


class Agent(C.Structure):
    _fields_ = [("domain", C.c_int), ("algo", C.c_int), ("policy", C.c_int),
                ("shared_w", C.c_int), ("n_actions", C.c_int), ("basis", Basis),
                ("seed", C.c_uint64), ("env_offset", C.c_int64),
                ("gamma", C.c_double), ("lr", C.c_double), ("alpha", C.c_double),
                ("epsilon", C.c_double), ("tau", C.c_double),
                ("eps_thr", C.c_uint32), ("max_episode_steps", C.c_uint32),
                ("lam", C.c_double), ("trace", C.c_int), ("lr_td", C.c_double),
                ("apolicy", C.c_int), ("aepsilon", C.c_double), ("atau", C.c_double), ("aeps_thr", C.c_uint32),
                ("sigma", C.c_double), ("n_steps", C.c_int),
                ("apol_same", C.c_int), ("eps_decay", C.c_double), ("eps_min", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("episodes", C.c_uint64),
                ("episodes_truncated", C.c_uint64), ("sum_episode_steps", C.c_uint64),
                ("sum_abs_td_error", C.c_double), ("sum_reward", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    src = [os.path.join(_HERE, f) for f in ("rsrl_oracle.c", "rsrl_oracle_impl.h", "rsrl_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    u32p = C.POINTER(C.c_uint32)
    L.orc_domain_dim.restype = C.c_int
    L.orc_domain_actions.restype = C.c_int
    L.orc_basis_nfeat.restype = C.c_int
    L.orc_basis_nfeat.argtypes = [C.POINTER(Basis)]
    L.orc_mulhi.restype = C.c_uint32
    L.orc_mulhi.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_eps_threshold.restype = C.c_uint32
    L.orc_eps_threshold.argtypes = [C.c_double]
    L.orc_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, u32p]
    L.orc_agent_init.argtypes = [C.POINTER(Agent)] + [C.c_int] * 8 + [C.c_uint64, C.c_int64] + \
        [C.c_double] * 5 + [C.c_uint32]
    for S, R in (("f64", C.c_double), ("f32", C.c_float), ("f32d", C.c_float)):
        Rp = C.POINTER(R)
        g = lambda n: getattr(L, f"{n}_{S}")
        g("orc_domain_step").restype = C.c_int
        g("orc_domain_step").argtypes = [C.c_int, Rp, C.c_int, Rp]
        g("orc_domain_is_terminal").restype = C.c_int
        g("orc_domain_is_terminal").argtypes = [C.c_int, Rp]
        g("orc_domain_reset").argtypes = [C.c_int, Rp]
        g("orc_fourier_project").argtypes = [C.c_int, C.c_int, Rp, Rp, Rp, Rp]
        g("orc_q_evaluate").argtypes = [C.POINTER(Basis), Rp, C.c_int, Rp, Rp]
        g("orc_q_evaluate_index").restype = R
        g("orc_q_evaluate_index").argtypes = [C.POINTER(Basis), Rp, C.c_int, Rp, C.c_int]
        g("orc_find_max").restype = C.c_int
        g("orc_find_max").argtypes = [Rp, C.c_int, Rp]
        g("orc_q_update_index").argtypes = [C.POINTER(Basis), Rp, C.c_int, Rp, C.c_int, R, R]
        g("orc_argmaxima").restype = C.c_int
        g("orc_argmaxima").argtypes = [Rp, C.c_int, C.POINTER(C.c_int), Rp]
        g("orc_argmax_first").restype = C.c_int
        g("orc_argmax_first").argtypes = [Rp, C.c_int]
        g("orc_greedy_probs").argtypes = [Rp, C.c_int, Rp]
        g("orc_egreedy_probs").argtypes = [Rp, C.c_int, R, Rp]
        g("orc_softmax_probs").argtypes = [Rp, C.c_int, R, Rp]
        g("orc_policy_probs").argtypes = [C.c_int, Rp, C.c_int, R, R, Rp]
        g("orc_policy_sample").restype = C.c_int
        g("orc_policy_sample").argtypes = [C.c_int, Rp, C.c_int, C.c_uint32, R, u32p]
        g("orc_policy_mode").restype = C.c_int
        g("orc_policy_mode").argtypes = [C.c_int, Rp, C.c_int, R]
        g("orc_td_error").restype = R
        g("orc_td_error").argtypes = [C.POINTER(Agent), Rp, Rp, C.c_int, R, Rp, C.c_int, u32p, Rp]
        g("orc_handle").restype = R
        g("orc_handle").argtypes = [C.POINTER(Agent), Rp, Rp, C.c_int, R, Rp, C.c_int, u32p]
        g("orc_run_create").restype = C.c_void_p
        g("orc_run_create").argtypes = [C.POINTER(Agent), C.c_int64]
        g("orc_run_destroy").argtypes = [C.c_void_p]
        g("orc_run_state").restype = Rp
        g("orc_run_state").argtypes = [C.c_void_p]
        g("orc_run_action").restype = C.POINTER(C.c_int32)
        g("orc_run_action").argtypes = [C.c_void_p]
        g("orc_run_ep_step").restype = u32p
        g("orc_run_ep_step").argtypes = [C.c_void_p]
        g("orc_run_weights").restype = Rp
        g("orc_run_weights").argtypes = [C.c_void_p]
        g("orc_run_t").restype = C.c_uint64
        g("orc_run_t").argtypes = [C.c_void_p]
        g("orc_run_set_epsilon").argtypes = [C.c_void_p, C.c_double]
        g("orc_run_reset").argtypes = [C.c_void_p]
        g("orc_run_train").argtypes = [C.c_void_p, C.c_int64, C.POINTER(Stats)]
        g("orc_run_train_sparse_lambda").restype = C.c_int
        g("orc_run_train_sparse_lambda").argtypes = [C.c_void_p, C.c_int64, C.POINTER(Stats)]
        g("orc_run_sparse_trace").argtypes = [C.c_void_p, C.c_int64, Rp]
        g("orc_run_teacher").argtypes = [C.c_void_p, C.POINTER(Stats), Rp, C.POINTER(C.c_int32), Rp, Rp, C.POINTER(C.c_uint8), Rp]
        g("orc_run_teacher_sparse_lambda").argtypes = [C.c_void_p, C.POINTER(Stats), Rp, C.POINTER(C.c_int32), Rp, Rp, C.POINTER(C.c_uint8), Rp]
        g("orc_run_teacher_sparse_lambda").restype = C.c_int
        g("orc_run_eps").restype = Rp
        g("orc_run_eps").argtypes = [C.c_void_p]
        g("orc_handle_lambda").restype = R
        g("orc_handle_lambda").argtypes = [C.POINTER(Agent), Rp, Rp, Rp, C.c_int, R, Rp, C.c_int, u32p]
        g("orc_handle_gq").restype = R
        g("orc_handle_gq").argtypes = [C.POINTER(Agent), Rp, Rp, Rp, C.c_int, R, Rp, C.c_int]
        g("orc_v_evaluate").restype = R
        g("orc_v_evaluate").argtypes = [C.POINTER(Basis), Rp, Rp]
        g("orc_handle_td").restype = R
        g("orc_handle_td").argtypes = [C.POINTER(Agent), Rp, Rp, Rp, R, Rp, C.c_int]
        g("orc_run_train_fast").restype = C.c_int
        g("orc_run_train_fast").argtypes = [C.c_void_p, C.c_int64, C.POINTER(Stats)]
        g("orc_run_train_dev").restype = C.c_int
        g("orc_run_train_dev").argtypes = [C.c_void_p, C.c_int64, C.POINTER(Stats)]
        g("orc_run_invalidate_q").argtypes = [C.c_void_p]
        g("orc_run_train_wave").restype = C.c_int
        g("orc_run_train_wave").argtypes = [C.c_void_p, C.c_int64, C.POINTER(Stats), C.c_int]
        g("orc_run_reset_wave").argtypes = [C.c_void_p]
        g("orc_run_train_shared_dev").restype = C.c_int
        g("orc_run_train_shared_dev").argtypes = [C.c_void_p, C.c_int64, C.POINTER(Stats)]
        g("orc_qsigma_new").restype = C.c_void_p
        g("orc_qsigma_new").argtypes = [C.c_int]
        g("orc_qsigma_free").argtypes = [C.c_void_p]
        g("orc_qsigma_len").restype = C.c_int
        g("orc_qsigma_len").argtypes = [C.c_void_p]
        g("orc_handle_qsigma").restype = R
        g("orc_handle_qsigma").argtypes = [C.POINTER(Agent), Rp, C.c_void_p, Rp, C.c_int, R, Rp, C.c_int, u32p]
        g("orc_run_traces").restype = Rp
        g("orc_run_traces").argtypes = [C.c_void_p]
        g("orc_run_train_hook").argtypes = [C.c_void_p, C.c_int64, C.POINTER(Stats), C.c_void_p, C.c_void_p]
        g("orc_run_rollout_greedy").restype = C.c_int
        g("orc_run_rollout_greedy").argtypes = [C.c_void_p, C.c_int64, u32p, Rp]
        g("orc_run_rollout_greedy_margin").restype = C.c_int
        g("orc_run_rollout_greedy_margin").argtypes = [C.c_void_p, C.c_int64, u32p, Rp, Rp]
        g("orc_run_rollout_policy").restype = C.c_int
        g("orc_run_rollout_policy").argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_int64, u32p, Rp, C.POINTER(C.c_int32)]


def _np_dtype(prec):
    return np.float64 if prec == "f64" else np.float32


def _ct(prec):
    return C.c_double if prec == "f64" else C.c_float


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def make_agent(domain=MOUNTAIN_CAR, basis=FOURIER, order=5, n_tilings=8, tiles_per_dim=8,
               algo=QLEARNING, policy=EGREEDY, shared_w=False, seed=0, env_offset=0,
               gamma=0.9, lr=0.001, alpha=1.0, epsilon=0.1, tau=1.0, max_episode_steps=1000, lam=0.0,
               trace=TRACE_ACCUMULATE, lr_td=0.0, agent_policy=None, agent_epsilon=0.1, agent_tau=1.0, sigma=0.0, n_steps=1,
               epsilon_decay=1.0, epsilon_min=0.0):
    ag = Agent()
    lib().orc_agent_init(C.byref(ag), domain, basis, order, n_tilings, tiles_per_dim, algo, policy,
                         int(bool(shared_w)), seed, env_offset, gamma, lr, alpha, epsilon, tau,
                         max_episode_steps)
    ag.lam, ag.trace, ag.lr_td = lam, trace, lr_td
    ag.sigma, ag.n_steps = sigma, n_steps
    if agent_policy is not None:      # the agent's own policy object (sarsa.rs:35-41, expected_sarsa.rs:22-29); default: the behaviour policy
        ag.apolicy, ag.aepsilon, ag.atau = agent_policy, agent_epsilon, agent_tau
        ag.aeps_thr = lib().orc_eps_threshold(agent_epsilon)
        ag.apol_same = 0
    # the drivers' schedule `agent.policy.epsilon *= decay` once per episode of a learner (examples/sarsa_lambda.rs:68)
    ag.eps_decay, ag.eps_min = epsilon_decay, epsilon_min
    return ag


def draw(seed, env_id, t, block):
    out = (C.c_uint32 * 4)()
    lib().orc_draw(seed, env_id, t, block, out)
    return np.array(out[:], dtype=np.uint32)


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return [int(v) for v in out]


def domain_step(domain, s, a, prec="f64"):
    """One Domain::step from state s with action a -> (s', reward, terminal)."""
    dt, ct = _np_dtype(prec), _ct(prec)
    ns = np.array(s, dtype=dt).copy()
    r = ct(0)
    term = getattr(lib(), f"orc_domain_step_{prec}")(domain, _ptr(ns, ct), int(a), C.byref(r))
    return ns, float(r.value), bool(term)


def domain_reset(domain, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    s = np.zeros(lib().orc_domain_dim(domain), dtype=dt)
    getattr(lib(), f"orc_domain_reset_{prec}")(domain, _ptr(s, ct))
    return s


def domain_is_terminal(domain, s, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    s = np.array(s, dtype=dt)
    return bool(getattr(lib(), f"orc_domain_is_terminal_{prec}")(domain, _ptr(s, ct)))


def domain_bounds(domain):
    lo = (C.c_double * 8)()
    hi = (C.c_double * 8)()
    lib().orc_domain_bounds(domain, lo, hi)
    d = lib().orc_domain_dim(domain)
    return np.array(lo[:d]), np.array(hi[:d])


def fourier_project(domain, order, s, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    lo, hi = domain_bounds(domain)
    lo, hi = lo.astype(dt), hi.astype(dt)
    s = np.array(s, dtype=dt)
    D = len(s)
    phi = np.zeros((order + 1) ** D, dtype=dt)
    getattr(lib(), f"orc_fourier_project_{prec}")(order, D, _ptr(lo, ct), _ptr(hi, ct), _ptr(s, ct),
                                                  _ptr(phi, ct))
    return phi


def tile_indices(ag_or_basis, s):
    b = ag_or_basis.basis if isinstance(ag_or_basis, Agent) else ag_or_basis
    s = np.array(s, dtype=np.float32)
    idx = (C.c_int * b.n_tilings)()
    lib().orc_tile_indices(C.byref(b), _ptr(s, C.c_float), idx)
    return np.array(idx[:], dtype=np.int32)


def n_features(ag):
    return lib().orc_basis_nfeat(C.byref(ag.basis))


def q_evaluate(ag, W, s, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    W = np.ascontiguousarray(W, dtype=dt)
    s = np.array(s, dtype=dt)
    q = np.zeros(ag.n_actions, dtype=dt)
    getattr(lib(), f"orc_q_evaluate_{prec}")(C.byref(ag.basis), _ptr(W, ct), ag.n_actions, _ptr(s, ct),
                                             _ptr(q, ct))
    return q


def argmaxima(v, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    v = np.array(v, dtype=dt)
    ixs = (C.c_int * max(1, len(v)))()
    mx = ct(0)
    n = getattr(lib(), f"orc_argmaxima_{prec}")(_ptr(v, ct), len(v), ixs, C.byref(mx))
    return list(ixs[:n]), float(mx.value)


def argmax_first(v, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    v = np.array(v, dtype=dt)
    return getattr(lib(), f"orc_argmax_first_{prec}")(_ptr(v, ct), len(v))


def find_max(v, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    v = np.array(v, dtype=dt)
    val = ct(0)
    i = getattr(lib(), f"orc_find_max_{prec}")(_ptr(v, ct), len(v), C.byref(val))
    return i, float(val.value)


def policy_probs(policy, q, eps=0.0, tau=1.0, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    q = np.array(q, dtype=dt)
    p = np.zeros(len(q), dtype=dt)
    getattr(lib(), f"orc_policy_probs_{prec}")(policy, _ptr(q, ct), len(q), ct(eps), ct(tau), _ptr(p, ct))
    return p


def policy_sample(policy, q, x, eps=0.0, tau=1.0, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    q = np.array(q, dtype=dt)
    xx = (C.c_uint32 * 4)(*[int(v) for v in x])
    return getattr(lib(), f"orc_policy_sample_{prec}")(policy, _ptr(q, ct), len(q),
                                                       lib().orc_eps_threshold(eps), ct(tau), xx)


def policy_mode(policy, q, tau=1.0, prec="f64"):
    dt, ct = _np_dtype(prec), _ct(prec)
    q = np.array(q, dtype=dt)
    return getattr(lib(), f"orc_policy_mode_{prec}")(policy, _ptr(q, ct), len(q), ct(tau))


def handle(ag, W, s, a, r, ns, term, x_inner=(0, 0, 0, 0), prec="f64"):
    """Agent handle on one transition; W (F,A) is updated in place; returns the TD error."""
    dt, ct = _np_dtype(prec), _ct(prec)
    assert W.dtype == dt and W.flags.c_contiguous
    s = np.array(s, dtype=dt)
    ns = np.array(ns, dtype=dt)
    xx = (C.c_uint32 * 4)(*[int(v) for v in x_inner])
    return float(getattr(lib(), f"orc_handle_{prec}")(C.byref(ag), _ptr(W, ct), _ptr(s, ct), int(a), ct(r),
                                                      _ptr(ns, ct), int(term), xx))


def handle_lambda(ag, W, Z, s, a, r, ns, term, x_inner=(0, 0, 0, 0), prec="f64"):
    """SARSA(lambda) / Q(lambda) handle on one transition; W and Z (F,A) are updated in place; returns the TD error."""
    dt, ct = _np_dtype(prec), _ct(prec)
    assert W.dtype == dt and Z.dtype == dt and W.flags.c_contiguous and Z.flags.c_contiguous
    s = np.array(s, dtype=dt)
    ns = np.array(ns, dtype=dt)
    xx = (C.c_uint32 * 4)(*[int(v) for v in x_inner])
    return float(getattr(lib(), f"orc_handle_lambda_{prec}")(C.byref(ag), _ptr(W, ct), _ptr(Z, ct), _ptr(s, ct), int(a),
                                                             ct(r), _ptr(ns, ct), int(term), xx))


def v_evaluate(ag, w, s, prec="f64"):
    """V(s) = <phi(s), w> of a prediction agent (ScalarLFA); w has F entries"""
    dt, ct = _np_dtype(prec), _ct(prec)
    w = np.ascontiguousarray(w, dtype=dt).reshape(-1)
    s = np.array(s, dtype=dt)
    return float(getattr(lib(), f"orc_v_evaluate_{prec}")(C.byref(ag.basis), _ptr(w, ct), _ptr(s, ct)))


def handle_td(ag, w, z, s, r, ns, term, prec="f64"):
    """TD / TDLambda handle on one transition; w (and the trace z, TDLambda only) with F entries are updated in place;
    returns the TD error."""
    dt, ct = _np_dtype(prec), _ct(prec)
    assert w.dtype == dt and w.flags.c_contiguous and (z is None or (z.dtype == dt and z.flags.c_contiguous))
    s = np.array(s, dtype=dt)
    ns = np.array(ns, dtype=dt)
    zp = _ptr(z, ct) if z is not None else None
    return float(getattr(lib(), f"orc_handle_td_{prec}")(C.byref(ag), _ptr(w, ct), zp, _ptr(s, ct), ct(r), _ptr(ns, ct), int(term)))


def handle_gq(ag, W, V, s, a, r, ns, term, prec="f64"):
    """GreedyGQ handle on one transition; W (fa_q) and V (fa_td), both (F,A), are updated in place; returns td_error."""
    dt, ct = _np_dtype(prec), _ct(prec)
    assert W.dtype == dt and V.dtype == dt and W.flags.c_contiguous and V.flags.c_contiguous
    s = np.array(s, dtype=dt)
    ns = np.array(ns, dtype=dt)
    return float(getattr(lib(), f"orc_handle_gq_{prec}")(C.byref(ag), _ptr(W, ct), _ptr(V, ct), _ptr(s, ct), int(a),
                                                         ct(r), _ptr(ns, ct), int(term)))


class QSigmaBackup:
    """One QSigma agent's n-step backup (q_sigma.rs:26-64) for single-transition tests: handle() updates W in place when the
    backup is full and returns the residual it pushed."""

    def __init__(self, n_steps, prec="f64"):
        self.prec, self._L = prec, lib()
        self._h = C.c_void_p(getattr(self._L, f"orc_qsigma_new_{prec}")(int(n_steps)))

    def __len__(self):
        return int(getattr(self._L, f"orc_qsigma_len_{self.prec}")(self._h))

    def handle(self, ag, W, s, a, r, ns, term, x_inner=(0, 0, 0, 0)):
        dt, ct = _np_dtype(self.prec), _ct(self.prec)
        assert W.dtype == dt and W.flags.c_contiguous
        s, ns = np.array(s, dtype=dt), np.array(ns, dtype=dt)
        xx = (C.c_uint32 * 4)(*[int(v) for v in x_inner])
        return float(getattr(self._L, f"orc_handle_qsigma_{self.prec}")(C.byref(ag), _ptr(W, ct), self._h, _ptr(s, ct), int(a), ct(r),
                                                                         _ptr(ns, ct), int(term), xx))

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(self._L, f"orc_qsigma_free_{self.prec}")(self._h)
            self._h = None


class Run:
    """N independent (or shared-W) learners stepped by the oracle's driver loop."""

    def __init__(self, ag, n_envs, prec="f64"):
        self.ag, self.n, self.prec = ag, int(n_envs), prec
        self._dt, self._ct = _np_dtype(prec), _ct(prec)
        self._L = lib()
        self._h = C.c_void_p(getattr(self._L, f"orc_run_create_{prec}")(C.byref(ag), self.n))
        self.D = ag.basis.dim
        self.A = 1 if ag.algo in (TD, TD_LAMBDA) else ag.n_actions       # columns of the weight matrix
        self.F = n_features(ag)

    def _f(self, name):
        return getattr(self._L, f"{name}_{self.prec}")

    def close(self):
        if self._h:
            self._f("orc_run_destroy")(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def state(self):      # (N, D) view
        return np.ctypeslib.as_array(self._f("orc_run_state")(self._h), shape=(self.n, self.D))

    @property
    def action(self):
        return np.ctypeslib.as_array(self._f("orc_run_action")(self._h), shape=(self.n,))

    @property
    def ep_step(self):
        return np.ctypeslib.as_array(self._f("orc_run_ep_step")(self._h), shape=(self.n,))

    @property
    def weights(self):    # per-env (N, F, A) or shared (F, A) view
        shape = (self.F, self.A) if self.ag.shared_w else (self.n, self.F, self.A)
        return np.ctypeslib.as_array(self._f("orc_run_weights")(self._h), shape=shape)

    @property
    def traces(self):     # per-env (N, F, A) view (lambda agents only)
        return np.ctypeslib.as_array(self._f("orc_run_traces")(self._h), shape=(self.n, self.F, self.A))

    @property
    def eps(self):        # (N,) view: every learner's EpsilonGreedy.epsilon
        return np.ctypeslib.as_array(self._f("orc_run_eps")(self._h), shape=(self.n,))

    @property
    def t(self):
        return int(self._f("orc_run_t")(self._h))

    def set_epsilon(self, eps):
        self._f("orc_run_set_epsilon")(self._h, float(eps))

    def reset(self):
        self._f("orc_run_reset")(self._h)

    def train(self, n_steps):
        st = Stats()
        self._f("orc_run_train")(self._h, int(n_steps), C.byref(st))
        return st.as_dict()

    def train_sparse_lambda(self, n_steps):
        """SARSALambda / QLambda over ONE shared tile-coded table with a sparse trace per learner (the device's rule and order,
        rsrl_amd/csrc/kernels_sparse_lambda.hpp): float precisions are bit-identical to the HIP path, "f64" is the same rule in the reference's
        precision."""
        st = Stats()
        if self._f("orc_run_train_sparse_lambda")(self._h, int(n_steps), C.byref(st)) != 0:
            raise ValueError("train_sparse_lambda: SARSALambda / QLambda, tile coding, shared weights")
        return st.as_dict()

    def sparse_trace(self, i):
        """learner i's sparse trace as the dense (F, A) matrix it stands for"""
        out = np.zeros((self.F, self.A), dtype=self._dt)
        self._f("orc_run_sparse_trace")(self._h, int(i), _ptr(out, self._ct))
        return out

    def teacher_step(self):
        """ONE batch-step of train() as a teacher: successor states rounded to fp32, the handled transitions returned ->
        dict(from (N, D), action (N,), reward (N,), to (N, D), terminal (N,) uint8, td (N,), stats).  Replayed through the device's
        Handler::handle (its k-th call draws what this run's k-th batch-step drew) both sides learn from identical inputs."""
        st = Stats()
        out = dict(frm=np.empty((self.n, self.D), dtype=self._dt), action=np.empty(self.n, dtype=np.int32),
                   reward=np.empty(self.n, dtype=self._dt), to=np.empty((self.n, self.D), dtype=self._dt),
                   terminal=np.empty(self.n, dtype=np.uint8), td=np.empty(self.n, dtype=self._dt))
        self._f("orc_run_teacher")(self._h, C.byref(st), _ptr(out["frm"], self._ct), out["action"].ctypes.data_as(C.POINTER(C.c_int32)),
                                   _ptr(out["reward"], self._ct), _ptr(out["to"], self._ct),
                                   out["terminal"].ctypes.data_as(C.POINTER(C.c_uint8)), _ptr(out["td"], self._ct))
        out["stats"] = st.as_dict()
        return out

    def teacher_step_sparse_lambda(self):
        """teacher_step() for SARSALambda / QLambda over ONE shared tile table with sparse per-learner traces (orc_run_train_sparse_lambda's rule)"""
        st = Stats()
        out = dict(frm=np.empty((self.n, self.D), dtype=self._dt), action=np.empty(self.n, dtype=np.int32),
                   reward=np.empty(self.n, dtype=self._dt), to=np.empty((self.n, self.D), dtype=self._dt),
                   terminal=np.empty(self.n, dtype=np.uint8), td=np.empty(self.n, dtype=self._dt))
        if self._f("orc_run_teacher_sparse_lambda")(self._h, C.byref(st), _ptr(out["frm"], self._ct), out["action"].ctypes.data_as(C.POINTER(C.c_int32)),
                                                    _ptr(out["reward"], self._ct), _ptr(out["to"], self._ct),
                                                    out["terminal"].ctypes.data_as(C.POINTER(C.c_uint8)), _ptr(out["td"], self._ct)) != 0:
            raise ValueError("teacher_step_sparse_lambda: SARSALambda / QLambda on tile coding with shared weights only")
        out["stats"] = st.as_dict()
        return out

    def train_fast(self, n_steps):
        """Same results as train() with the repeated projections / heap traffic of the reference's call pattern removed
        (QLearning + Fourier + per-env W only): the 'optimised CPU' baseline."""
        st = Stats()
        if self._f("orc_run_train_fast")(self._h, int(n_steps), C.byref(st)) != 0:
            raise ValueError("train_fast: QLearning on a Fourier basis with per-env weights only")
        return st.as_dict()

    def train_dev(self, n_steps):
        """The driver loop in the DEVICE's evaluation order (carried phi / Q, rank-1 post-update Q): with prec="f32d" this is
        what the HIP path must reproduce bit for bit.  QLearning / SARSA / ExpectedSARSA / PAL, Fourier, per-env W."""
        st = Stats()
        if self._f("orc_run_train_dev")(self._h, int(n_steps), C.byref(st)) != 0:
            raise ValueError("train_dev: one-step control agents on a Fourier basis with per-env weights only")
        return st.as_dict()

    def train_wave(self, n_steps, bf16=False):
        """The driver loop in the evaluation order of the device's wave family (Fourier order 7 on a 4-D domain: lane
        partials + the DPP ladder; bf16: stochastic rounding of every updated weight) -- with prec="f32d" bit-identical to
        k_train_wave."""
        st = Stats()
        if self._f("orc_run_train_wave")(self._h, int(n_steps), C.byref(st), int(bool(bf16))) != 0:
            raise ValueError("train_wave: one-step control agents (f32 / bf16), lambda agents, GreedyGQ, TD, TDLambda (f32); Fourier order 7 on CartPole / Acrobot, per-env weights")
        return st.as_dict()

    def train_shared_dev(self, n_steps):
        """Shared-W training (dense basis) in the device's evaluation order: 512-learner block sums as four 128-long fma chains,
        rows reduced by lane partials + the DPP ladder, phase C of step t-1 fused with phase A of step t -- with prec="f32d"
        bit-identical to the HIP path's k_shared_step."""
        st = Stats()
        if self._f("orc_run_train_shared_dev")(self._h, int(n_steps), C.byref(st)) != 0:
            raise ValueError("train_shared_dev: one-step control agents, Fourier basis, shared weights, n_steps >= 1")
        return st.as_dict()

    def reset_wave(self):
        """reset() with the initial Q(s0,.) evaluated in the wave family's summation order"""
        self._f("orc_run_reset_wave")(self._h)

    def invalidate_q(self):
        """Q(s,.) carried between train_dev calls is stale (weights / states were written from outside)"""
        self._f("orc_run_invalidate_q")(self._h)

    def train_with_dw_hook(self, n_steps, hook):
        """Shared-W training; hook(dW: np.ndarray view of the local delta) runs once per batch-step before the
        delta is applied (in-place edits are kept) -- where a multi-rank run all-reduces it."""
        st = Stats()
        CB = C.CFUNCTYPE(None, C.POINTER(self._ct), C.c_int, C.c_void_p)

        def _cb(ptr, n, _user):
            hook(np.ctypeslib.as_array(ptr, shape=(n,)))
        cb = CB(_cb)
        self._f("orc_run_train_hook")(self._h, int(n_steps), C.byref(st), C.cast(cb, C.c_void_p), None)
        return st.as_dict()

    def rollout_policy(self, policy, step_limit, epsilon=0.1, tau=1.0, call=0):
        """Domain::rollout with the closure s -> policy.sample(rng, s) (any of the four policies); `call` = how many such rollouts
        came before (the draws' stream) -> (n_states, total_reward, actions (step_limit - 1, N))"""
        n_states = np.zeros(self.n, dtype=np.uint32)
        tot = np.zeros(self.n, dtype=self._dt)
        acts = np.zeros((max(int(step_limit) - 1, 0), self.n), dtype=np.int32)
        rc = self._f("orc_run_rollout_policy")(self._h, int(policy), float(epsilon), float(tau), int(call), int(step_limit),
                                               n_states.ctypes.data_as(C.POINTER(C.c_uint32)), _ptr(tot, self._ct),
                                               acts.ctypes.data_as(C.POINTER(C.c_int32)) if acts.size else None)
        if rc != 0:
            raise ValueError("rollout_policy: invalid step_limit / policy")
        return n_states, tot, acts

    def rollout_greedy(self, step_limit):
        n_states = np.zeros(self.n, dtype=np.uint32)
        tot = np.zeros(self.n, dtype=self._dt)
        rc = self._f("orc_run_rollout_greedy")(self._h, int(step_limit),
                                               n_states.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               _ptr(tot, self._ct))
        if rc != 0:
            raise ValueError("rollout_greedy: invalid step_limit or policy has no mode")
        return n_states, tot

    def rollout_greedy_margin(self, step_limit):
        """rollout_greedy + every learner's smallest argmax margin (largest minus second largest action value) over its action selections"""
        n_states = np.zeros(self.n, dtype=np.uint32)
        tot = np.zeros(self.n, dtype=self._dt)
        mm = np.zeros(self.n, dtype=self._dt)
        rc = self._f("orc_run_rollout_greedy_margin")(self._h, int(step_limit), n_states.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                      _ptr(tot, self._ct), _ptr(mm, self._ct))
        if rc != 0:
            raise ValueError("rollout_greedy_margin: invalid step_limit or policy has no mode")
        return n_states, tot, mm
