/*
 * rsrl_oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by rsrl_oracle.c: once with R = double (suffix _f64, the
 * reference-faithful precision: every number in tspooner/rsrl is f64,
 * rsrl_domains/src/lib.rs:49) and once with R = float (suffix _f32, same
 * algorithm in the device's arithmetic type, transcendentals evaluated in
 * double and rounded once, dot products / AXPY as explicit fma chains -- the
 * op order the HIP kernels use).
 *
 * Every function cites the reference file:line it restates.  Paths are
 * relative to /root/reference.  Nothing here is copied: the reference is
 * Rust, this is a C restatement of its arithmetic.
 */

#ifndef R
#error "define R, FN(), RC() before including"
#endif

/* ------------------------------------------------------------------ */
/* transcendental helpers: evaluate in double, round once to R         */
/* ------------------------------------------------------------------ */
#ifdef ORC_DEVTRIG
/* Third instantiation (suffix _f32d): the f32 body with the DEVICE's own branch-free polynomials restated operation by
 * operation (rsrl_amd/csrc/device_core.hpp: sincospi01, sincos_cw, exp_dev), so that the HIP path can be checked BITWISE
 * against a CPU run for arbitrarily many steps instead of "97 % of the trajectories".  Everything else (fma chains, op
 * order, integer decision logic, Philox) is already shared with the _f32 instantiation.  The polynomials themselves are
 * checked against libm in tests/test_oracle_golden.py (<= 2 ulp on the ranges the path uses). */
static inline void FN(sincospi01_dev)(R x, R* sn, R* cs) {       /* sin(pi x), cos(pi x), x in [0, 1] */
    const R q = rintf(x * 2.0f);
    const R r = fmaf(q, -0.5f, x);
    const R u = r * r;
    R ps = 0.08100174367427826f, pc = 0.23132924735546112f, s, c;
    int qi;
    ps = fmaf(ps, u, -0.5992020964622498f);
    ps = fmaf(ps, u, 2.5501625537872314f);
    ps = fmaf(ps, u, -5.167712688446045f);
    ps = fmaf(ps, u, 3.1415927410125732f);
    s = ps * r;
    pc = fmaf(pc, u, -1.335044503211975f);
    pc = fmaf(pc, u, 4.058707237243652f);
    pc = fmaf(pc, u, -4.934802055358887f);
    c = fmaf(pc, u, 1.0f);
    qi = (int)q;
    *sn = (qi == 1) ? c : ((qi == 2) ? -s : s);
    *cs = (qi == 1) ? -s : ((qi == 2) ? -c : c);
}
static inline void FN(sincos_cw_dev)(R x, R* sn, R* cs) {        /* sin(x), cos(x), |x| <= 100 */
    const R n = rintf(x * 0.6366197466850281f);
    R r = fmaf(n, -1.5707963705062866f, x), u, ps, pc, s, c;
    int q;
    r = fmaf(n, 4.371138828673793e-08f, r);
    u = r * r;
    ps = 2.715809387154877e-06f;
    ps = fmaf(ps, u, -0.00019839033484458923f);
    ps = fmaf(ps, u, 0.008333328180015087f);
    ps = fmaf(ps, u, -0.1666666716337204f);
    ps = fmaf(ps, u, 1.0f);
    s = ps * r;
    pc = 2.4362980184378102e-05f;
    pc = fmaf(pc, u, -0.001388643286190927f);
    pc = fmaf(pc, u, 0.04166661202907562f);
    pc = fmaf(pc, u, -0.5f);
    c = fmaf(pc, u, 1.0f);
    q = ((int)n) & 3;
    *sn = (q == 0) ? s : ((q == 1) ? c : ((q == 2) ? -s : -c));
    *cs = (q == 0) ? c : ((q == 1) ? -s : ((q == 2) ? -c : s));
}
static inline R FN(cos_)(R x) { R s, c; FN(sincos_cw_dev)(x, &s, &c); return c; }
static inline R FN(sin_)(R x) { R s, c; FN(sincos_cw_dev)(x, &s, &c); return s; }
/* exp_dev: n = rint(x log2 e), two-term Cody-Waite reduction, degree-5 polynomial in r on top of 1 + r, scaled by 2^n;
 * below -87 the result is 0 (no denormals), above 88.5 +inf */
static inline R FN(exp_)(R x) {
    const R n = rintf(x * 1.4426950216293335f);
    R r = fmaf(n, -0.693145751953125f, x), p;
    r = fmaf(n, -1.4286067653302337e-06f, r);
    p = 1.9875691e-4f;
    p = fmaf(p, r, 1.3981999e-3f);
    p = fmaf(p, r, 8.3334519e-3f);
    p = fmaf(p, r, 4.1665795e-2f);
    p = fmaf(p, r, 1.6666665e-1f);
    p = fmaf(p, r, 5.0000001e-1f);
    p = fmaf(p, r * r, r);
    p = p + 1.0f;
    if (x < -87.0f) return 0.0f;
    if (x > 88.5f) return INFINITY;
    return ldexpf(p, (int)n);
}
#else
static inline R FN(cos_)(R x) { return (R)cos((double)x); }
static inline R FN(sin_)(R x) { return (R)sin((double)x); }
static inline R FN(exp_)(R x) { return (R)exp((double)x); }
#endif
/* cos(pi * x): reference computes (PI * cx).cos() in f64 (lfa Fourier, recalled).
 * For R = double this is literally that; for R = float it is the correctly
 * rounded cospi of the f32 argument, i.e. what a good cospif approximates. */
static inline R FN(cospi_)(R x) { return (R)cos(M_PI * (double)x); }
static inline R FN(fma_)(R a, R b, R c) { return (R)RC(fma)(a, b, c); }

/* clip!(lb, x, ub) = lb.max(ub.min(x))   rsrl_domains/src/macros.rs:20-24 */
static inline R FN(clip_)(R lb, R x, R ub) {
    R m = (ub < x) ? ub : x;          /* ub.min(x) (NaN-free inputs) */
    return (lb > m) ? lb : m;         /* lb.max(..) */
}
/* wrap!(lb, x, ub)                       rsrl_domains/src/macros.rs:3-18 */
static inline R FN(wrap_)(R lb, R x, R ub) {
    R nx = x, diff = ub - lb;
    while (nx > ub) nx -= diff;
    while (nx < lb) nx += diff;
    return nx;
}

/* ------------------------------------------------------------------ */
/* Domains                                                             */
/* ------------------------------------------------------------------ */

/* MountainCar::update_state + dv      mountain_car/discrete.rs:58-65
 * constants                            mountain_car/discrete.rs:8-22 */
static void FN(mc_update_state)(R* s, int a) {
    const R X_MIN = (R)-1.2, X_MAX = (R)0.6, V_MIN = (R)-0.07, V_MAX = (R)0.07;
    const R FORCE_G = (R)-0.0025, FORCE_CAR = (R)0.001, HILL_FREQ = (R)3.0;
    const R act = (R)(a - 1);                           /* ALL_ACTIONS = [-1,0,1] */
    R x = s[0], v = s[1];
    R dv = FORCE_CAR * act + FORCE_G * FN(cos_)(HILL_FREQ * x);
    v = FN(clip_)(V_MIN, v + dv, V_MAX);
    x = FN(clip_)(X_MIN, x + v, X_MAX);
    s[0] = x; s[1] = v;
}
static int FN(mc_is_terminal)(const R* s) { return s[0] >= (R)0.6; }   /* discrete.rs:76-82 */

/* runge_kutta4                         rsrl_domains/src/ode.rs:1-43
 * f ignores the time argument; dim fixed at 4 for both users. */
typedef void (*FN(grad_fn))(R ctl, const R* y, R* out);
static void FN(rk4)(FN(grad_fn) f, R ctl, R* y, R dx) {
    R k1[4], k2[4], k3[4], k4[4], t[4];
    int i;
    f(ctl, y, k1); for (i = 0; i < 4; i++) k1[i] *= dx;
    for (i = 0; i < 4; i++) t[i] = y[i] + k1[i] / (R)2.0;
    f(ctl, t, k2); for (i = 0; i < 4; i++) k2[i] *= dx;
    for (i = 0; i < 4; i++) t[i] = y[i] + k2[i] / (R)2.0;
    f(ctl, t, k3); for (i = 0; i < 4; i++) k3[i] *= dx;
    for (i = 0; i < 4; i++) t[i] = y[i] + k3[i];
    f(ctl, t, k4); for (i = 0; i < 4; i++) k4[i] *= dx;
    for (i = 0; i < 4; i++)
        y[i] += (k1[i] + (R)2.0 * k2[i] + (R)2.0 * k3[i] + k4[i]) / (R)6.0;
}

/* CartPole::grad                        cart_pole.rs:52-72; consts cart_pole.rs:7-26, consts.rs:4-10 */
static void FN(cp_grad)(R force, const R* y, R* out) {
    const R G = (R)9.8, FOUR_THIRDS = (R)(4.0 / 3.0);
    const R POLE_COM = (R)0.5, POLE_MASS = (R)0.1, CART_MASS = (R)1.0;
    const R POLE_MOMENT = POLE_COM * POLE_MASS, TOTAL_MASS = CART_MASS + POLE_MASS;
    R dx = y[1], theta = y[2], dtheta = y[3];
    R cos_t = FN(cos_)(theta), sin_t = FN(sin_)(theta);
    R z = (force + POLE_MOMENT * dtheta * dtheta * sin_t) / TOTAL_MASS;
    R numer = G * sin_t - cos_t * z;
    R denom = FOUR_THIRDS * POLE_COM - POLE_MOMENT * cos_t * cos_t;
    R ddtheta = numer / denom;
    out[0] = dx;
    out[2] = dtheta;
    out[3] = ddtheta;
    out[1] = z - POLE_COM * ddtheta * cos_t;
}
/* CartPole::update_state                cart_pole.rs:39-50 */
static void FN(cp_update_state)(R* s, int a) {
    const R TWELVE_DEG = (R)(M_PI / 15.0);
    const R force = (a == 0) ? (R)-10.0 : (R)10.0;      /* ALL_ACTIONS cart_pole.rs:26 */
    R ns[4] = { s[0], s[1], s[2], s[3] };
    FN(rk4)(FN(cp_grad), force, ns, (R)0.02);
    s[0] = FN(clip_)((R)-2.4, ns[0], (R)2.4);
    s[1] = FN(clip_)((R)-6.0, ns[1], (R)6.0);
    s[2] = FN(clip_)(-TWELVE_DEG, ns[2], TWELVE_DEG);
    s[3] = FN(clip_)((R)-2.0, ns[3], (R)2.0);
}
/* CartPole::emit terminal predicate     cart_pole.rs:83-97 */
static int FN(cp_is_terminal)(const R* s) {
    const R TWELVE_DEG = (R)(M_PI / 15.0);
    return s[0] <= (R)-2.4 || s[0] >= (R)2.4 || s[2] <= -TWELVE_DEG || s[2] >= TWELVE_DEG;
}

/* Acrobot::grad                         acrobot.rs:81-108; consts acrobot.rs:8-36 */
static void FN(ac_grad)(R torque, const R* y, R* out) {
    const R M1 = 1, M2 = 1, L1 = 1, LC1 = (R)0.5, LC2 = (R)0.5, I1 = 1, I2 = 1, G = (R)9.8;
    const R PI_OVER_2 = (R)(M_PI / 2.0);
    R theta1 = y[0], theta2 = y[1], dtheta1 = y[2], dtheta2 = y[3];
    R sin_t2 = FN(sin_)(theta2), cos_t2 = FN(cos_)(theta2);
    R d1 = M1 * LC1 * LC1 + M2 * (L1 * L1 + LC2 * LC2 + (R)2.0 * L1 * LC2 * cos_t2) + I1 + I2;
    R d2 = M2 * (LC2 * LC2 + L1 * LC2 * cos_t2) + I2;
    R phi2 = M2 * LC2 * G * FN(cos_)(theta1 + theta2 - PI_OVER_2);
    R phi1 = (R)-1.0 * L1 * LC2 * dtheta2 * dtheta2 * sin_t2
           - (R)2.0 * M2 * L1 * LC2 * dtheta2 * dtheta1 * sin_t2
           + (M1 * LC1 + M2 * L1) * G * FN(cos_)(theta1 - PI_OVER_2)
           + phi2;
    R dd1 = (torque + d2 / d1 * phi1 - M2 * L1 * LC2 * dtheta1 * dtheta1 * sin_t2 - phi2)
          / (M2 * LC2 * LC2 + I2 - d2 * d2 / d1);
    out[0] = dtheta1;
    out[1] = dtheta2;
    out[2] = dd1;
    out[3] = -(d2 * dd1 + phi1) / d1;
}
/* Acrobot::update_state                 acrobot.rs:60-79 */
static void FN(ac_update_state)(R* s, int a) {
    const R PI_ = (R)M_PI;
    const R torque = (R)(a - 1);                        /* ALL_ACTIONS acrobot.rs:35-36 */
    R ns[4] = { s[0], s[1], s[2], s[3] };
    FN(rk4)(FN(ac_grad), torque, ns, (R)0.2);
    s[0] = FN(wrap_)(-PI_, ns[0], PI_);
    s[1] = FN(wrap_)(-PI_, ns[1], PI_);
    s[2] = FN(clip_)((R)-4.0 * PI_, ns[2], (R)4.0 * PI_);
    s[3] = FN(clip_)((R)-9.0 * PI_, ns[3], (R)9.0 * PI_);
}
/* Acrobot::is_terminal                  acrobot.rs:56-58 */
static int FN(ac_is_terminal)(const R* s) {
    return FN(cos_)(s[0]) + FN(cos_)(s[0] + s[1]) < (R)-1.0;
}

/* Default::default()  discrete.rs:68-70, cart_pole.rs:75-77, acrobot.rs:111-113 */
void FN(orc_domain_reset)(int domain, R* s) {
    int i, d = orc_domain_dim(domain);
    for (i = 0; i < d; i++) s[i] = 0;
    if (domain == ORC_MOUNTAIN_CAR) { s[0] = (R)-0.5; s[1] = 0; }
}
int FN(orc_domain_is_terminal)(int domain, const R* s) {
    switch (domain) {
    case ORC_MOUNTAIN_CAR: return FN(mc_is_terminal)(s);
    case ORC_CART_POLE:    return FN(cp_is_terminal)(s);
    default:               return FN(ac_is_terminal)(s);
    }
}
/* Domain::step          discrete.rs:84-95, cart_pole.rs:99-110, acrobot.rs:130-141
 * s is advanced in place; returns terminal flag of the new state. */
int FN(orc_domain_step)(int domain, R* s, int a, R* reward) {
    int term;
    switch (domain) {
    case ORC_MOUNTAIN_CAR:
        FN(mc_update_state)(s, a); term = FN(mc_is_terminal)(s);
        *reward = term ? (R)0.0 : (R)-1.0; break;       /* REWARD_GOAL / REWARD_STEP */
    case ORC_CART_POLE:
        FN(cp_update_state)(s, a); term = FN(cp_is_terminal)(s);
        *reward = term ? (R)-1.0 : (R)0.0; break;       /* REWARD_TERMINAL / REWARD_STEP */
    default:
        FN(ac_update_state)(s, a); term = FN(ac_is_terminal)(s);
        *reward = term ? (R)0.0 : (R)-1.0; break;
    }
    return term;
}

/* ------------------------------------------------------------------ */
/* Bases (crate lfa 0.15 -- NOT in /root/reference: parity unpinned)   */
/* ------------------------------------------------------------------ */

/* Fourier::project + with_bias()  (lfa 0.15, recalled; call site
 * rsrl/examples/q_learning.rs:24).  Coefficient vectors: {0..=order}^D in
 * lexicographic order (last dimension fastest), all-zero vector skipped;
 * constant 1.0 feature stacked LAST.  F = (order+1)^D.
 * s~_i = (s_i - lo_i)/(hi_i - lo_i); phi_k = cos(pi * sum_i c_ki s~_i).
 *
 * f64 instantiation: literally the formula above ((PI * cx).cos()).
 * f32 instantiation (ORC_SEPARABLE): the device's evaluation order -- the basis is
 * separable, cos(pi(sum_i c_i s~_i)) = Re prod_i e^{i pi c_i s~_i}: per dimension one
 * correctly-rounded sincospi(s~_i), multiples by the angle-addition chain, then a complex
 * product over dimensions (fma forms written out).  In fp32 this is CLOSER to the f64
 * value than cos(pi * fl(sum c_i s~_i)) (5e-7 vs 3e-6 worst case at order 5). */
void FN(orc_fourier_project)(int order, int D, const R* lo, const R* hi, const R* s, R* phi) {
    int n1 = order + 1, F = 1, i, k, c[8];
    R sc[8];
    for (i = 0; i < D; i++) {
        F *= n1;
#ifdef ORC_SEPARABLE
        sc[i] = (s[i] - lo[i]) * ((R)1.0 / (hi[i] - lo[i]));      /* device op order: multiply by the rounded reciprocal */
#else
        sc[i] = (s[i] - lo[i]) / (hi[i] - lo[i]);
#endif
    }
#ifdef ORC_SEPARABLE
    {
        R ct[8][16], st[8][16];
        int n;
        for (i = 0; i < D; i++) {
            ct[i][0] = (R)1.0; st[i][0] = (R)0.0;
#ifdef ORC_DEVTRIG
            FN(sincospi01_dev)(sc[i], &st[i][1], &ct[i][1]);
#else
            ct[i][1] = (R)cos(M_PI * (double)sc[i]); st[i][1] = (R)sin(M_PI * (double)sc[i]);
#endif
            for (n = 2; n <= order; n++) {
                ct[i][n] = FN(fma_)(-st[i][n - 1], st[i][1], ct[i][n - 1] * ct[i][1]);
                st[i][n] = FN(fma_)(ct[i][n - 1], st[i][1], st[i][n - 1] * ct[i][1]);
            }
        }
        for (k = 1; k < F; k++) {
            int rem = k;
            R re, im;
            for (i = D - 1; i >= 0; i--) { c[i] = rem % n1; rem /= n1; }
            re = ct[0][c[0]]; im = st[0][c[0]];
            for (i = 1; i < D; i++) {
                R cr = ct[i][c[i]], sr = st[i][c[i]];
                R nre = FN(fma_)(-im, sr, re * cr);
                R nim = FN(fma_)(re, sr, im * cr);
                re = nre; im = nim;
            }
            phi[k - 1] = re;
        }
    }
#else
    for (k = 1; k < F; k++) {
        int rem = k;
        R cx = 0;
        for (i = D - 1; i >= 0; i--) { c[i] = rem % n1; rem /= n1; }
        for (i = 0; i < D; i++) cx = cx + (R)c[i] * sc[i];      /* fold(0.0, acc + c*v) */
        phi[k - 1] = FN(cospi_)(cx);
    }
#endif
    phi[F - 1] = (R)1.0;
}

/* ------------------------------------------------------------------ */
/* Q function: VectorLFA over dense or sparse features                 */
/* W is row-major (F, A) exactly like ndarray Array2 zeros((F,A))      */
/*   (fa/linear.rs:293-301, weights.ncols() = A at :358)               */
/* ------------------------------------------------------------------ */

/* q[a] = <phi, W[:,a]> for every column of the row-major (F, A) matrix */
static void FN(dot_columns)(const R* phi, const R* W, int A, int F, R* q) {
    int a, f;
    for (a = 0; a < A; a++) {
#ifdef ORC_SEPARABLE
        /* device order: 4 interleaved partial sums, q = (acc0 + acc1) + (acc2 + acc3) */
        R acc[4] = { 0, 0, 0, 0 };
        for (f = 0; f < F; f++) acc[f & 3] = FN(fma_)(phi[f], W[(size_t)f * A + a], acc[f & 3]);
        q[a] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#else
        R acc = 0;
        for (f = 0; f < F; f++) acc = FN(fma_)(phi[f], W[(size_t)f * A + a], acc);
        q[a] = acc;
#endif
    }
}
/* Function<(S,)>::evaluate -> Q(s,.) = W^T phi      fa/linear.rs:303-311 */
/* Wave-order switch (set by orc_run_train_wave around an agent's handle, e.g. QSigma's): Q(s,.) and the column update of an order-7 Fourier
 * basis on a 4-D domain go through the wave family's projection and summation order (wave_project / wave_dot below), so that agents written
 * against orc_q_evaluate / orc_q_update_index are bit-comparable with the device's one-wavefront-per-learner kernels. */
static int FN(g_wave_order) = 0;
static void FN(wave_project)(const orc_basis* b, const R* s, R* phi);
static R FN(wave_dot)(const R* phi, const R* W, int A, int a);
static inline size_t FN(wave_row)(int l, int j, int v);
static int FN(use_wave_order)(const orc_basis* b) { return FN(g_wave_order) && b->kind == ORC_FOURIER && b->order == 7 && b->dim == 4; }
void FN(orc_q_evaluate)(const orc_basis* b, const R* W, int A, const R* s, R* q) {
    int F = orc_basis_nfeat(b), a;
    if (FN(use_wave_order)(b)) {
        R* phi = (R*)malloc(sizeof(R) * 4096);
        FN(wave_project)(b, s, phi);
        for (a = 0; a < A; a++) q[a] = FN(wave_dot)(phi, W, A, a);
        free(phi);
    } else if (b->kind == ORC_FOURIER) {
        R* phi = (R*)malloc(sizeof(R) * (size_t)F);             /* reference allocs a feature array per call */
        FN(orc_fourier_project)(b->order, b->dim, FN(basis_lo)(b), FN(basis_hi)(b), s, phi);
        FN(dot_columns)(phi, W, A, F, q);
        free(phi);
    } else {
        int idx[ORC_MAX_TILINGS], t;
        float sf[8]; for (t = 0; t < b->dim; t++) sf[t] = (float)s[t];
        orc_tile_indices(b, sf, idx);
        for (a = 0; a < A; a++) {
            R acc = 0;
            for (t = 0; t < b->n_tilings; t++) acc = acc + W[(size_t)idx[t] * A + a];
            q[a] = acc;
        }
    }
}
/* Enumerable::evaluate_index -> Q(s,a)              fa/linear.rs:360-362 */
R FN(orc_q_evaluate_index)(const orc_basis* b, const R* W, int A, const R* s, int a) {
    R q[ORC_MAX_ACTIONS];
    FN(orc_q_evaluate)(b, W, A, s, q);      /* same arithmetic per column; the column is independent */
    return q[a];
}
/* Enumerable::find_max: fold (i,x): if acc.1 > x {acc} else {(i,x)} => ties -> LAST   core.rs:96-105 */
int FN(orc_find_max)(const R* q, int A, R* val) {
    int i, bi = 0; R bv = q[0];
    for (i = 1; i < A; i++) { if (bv > q[i]) { } else { bi = i; bv = q[i]; } }
    if (val) *val = bv;
    return bi;
}
/* Handler<StateActionUpdate>: W[:,a] += lr * error * phi(s)   fa/linear.rs:379-391 -> lfa update_index -> SGD (recalled) */
void FN(orc_q_update_index)(const orc_basis* b, R* W, int A, const R* s, int a, R lr, R error) {
    int F = orc_basis_nfeat(b), f;
    R scale = lr * error;
    if (FN(use_wave_order)(b)) {
        R* phi = (R*)malloc(sizeof(R) * 4096); int l, j, v;
        FN(wave_project)(b, s, phi);
        for (l = 0; l < 64; l++) for (j = 0; j < 8; j++) for (v = 0; v < 8; v++) {
            const size_t at = FN(wave_row)(l, j, v) * A + a;
            W[at] = FN(fma_)(scale, phi[(l * 8 + j) * 8 + v], W[at]);
        }
        free(phi);
    } else if (b->kind == ORC_FOURIER) {
        R* phi = (R*)malloc(sizeof(R) * (size_t)F);
        FN(orc_fourier_project)(b->order, b->dim, FN(basis_lo)(b), FN(basis_hi)(b), s, phi);
        for (f = 0; f < F; f++) W[(size_t)f * A + a] = FN(fma_)(scale, phi[f], W[(size_t)f * A + a]);
        free(phi);
    } else {
        int idx[ORC_MAX_TILINGS], t;
        float sf[8]; for (t = 0; t < b->dim; t++) sf[t] = (float)s[t];
        orc_tile_indices(b, sf, idx);
        for (t = 0; t < b->n_tilings; t++) W[(size_t)idx[t] * A + a] += scale;
    }
}

/* ------------------------------------------------------------------ */
/* utils.rs argmax helpers                                             */
/* ------------------------------------------------------------------ */

/* argmaxima           rsrl/src/utils.rs:6-21 (tolerance test first, max not raised by near-ties) */
int FN(orc_argmaxima)(const R* v, int n, int* ixs, R* maxv) {
    R max = -RMAX; int cnt = 0, i;
    for (i = 0; i < n; i++) {
        R d = v[i] - max; if (d < 0) d = -d;
        if (d < (R)1e-7) { ixs[cnt++] = i; }
        else if (v[i] > max) { max = v[i]; cnt = 0; ixs[cnt++] = i; }
    }
    if (maxv) *maxv = max;
    return cnt;
}
/* argmax_first        rsrl/src/utils.rs:23-34 */
int FN(orc_argmax_first)(const R* v, int n) {
    int bi = 0, j; R bx = -RMAX;
    for (j = 0; j < n; j++) if (v[j] - bx > (R)1e-7) { bi = j; bx = v[j]; }
    return bi;
}

/* ------------------------------------------------------------------ */
/* Policies                                                            */
/* ------------------------------------------------------------------ */

/* Greedy  Function<(S,)>: 1/|M| on argmaxima         policies/greedy.rs:30-44 */
void FN(orc_greedy_probs)(const R* q, int A, R* p) {
    int ixs[ORC_MAX_ACTIONS], n, i;
    n = FN(orc_argmaxima)(q, A, ixs, NULL);
    for (i = 0; i < A; i++) p[i] = 0;
    for (i = 0; i < n; i++) p[ixs[i]] = (R)1.0 / (R)n;
}
/* EpsilonGreedy Function<(S,)>: eps/A + p*(1-eps)    policies/epsilon_greedy.rs:38-45 */
void FN(orc_egreedy_probs)(const R* q, int A, R eps, R* p) {
    int i; R pr = eps / (R)A;
    FN(orc_greedy_probs)(q, A, p);
    for (i = 0; i < A; i++) p[i] = pr + p[i] * ((R)1.0 - eps);
}
/* softmax_stable + softmax                            policies/softmax.rs:15-37 */
void FN(orc_softmax_probs)(const R* q, int A, R tau, R* p) {
    int i; R m = q[0], z = 0;
    for (i = 1; i < A; i++) if (q[i] > m) m = q[i];             /* f64::max fold seeded with NaN */
    for (i = 0; i < A; i++) { p[i] = FN(exp_)((q[i] - m) / tau); z += p[i]; }
    for (i = 0; i < A; i++) { R v = p[i] / z; p[i] = (v < RMAX) ? v : RMAX; }
}
/* policy probabilities by kind                        Function<(S,)> of each policy */
void FN(orc_policy_probs)(int policy, const R* q, int A, R eps, R tau, R* p) {
    int i;
    switch (policy) {
    case ORC_GREEDY:  FN(orc_greedy_probs)(q, A, p); break;
    case ORC_EGREEDY: FN(orc_egreedy_probs)(q, A, eps, p); break;
    case ORC_SOFTMAX: FN(orc_softmax_probs)(q, A, tau, p); break;
    default: for (i = 0; i < A; i++) p[i] = (R)1.0 / (R)A; break;   /* Random  random.rs:19-26 */
    }
}
/* Greedy::sample -> argmax_choose_rng                 greedy.rs:77-81, utils.rs:63-79
 * x_tie is the u32 draw used when |M| > 1 (slice.choose -> uniform index). */
static int FN(greedy_sample)(const R* q, int A, uint32_t x_tie) {
    int ixs[ORC_MAX_ACTIONS], n;
    n = FN(orc_argmaxima)(q, A, ixs, NULL);
    if (n == 1) return ixs[0];
    if (n == 0) return (int)orc_mulhi(x_tie, (uint32_t)A);   /* all Q NaN / -inf: the reference panics ("No valid maxima",
                                                                utils.rs:70-76); the build keeps the action in [0, A) */
    return ixs[orc_mulhi(x_tie, (uint32_t)n)];
}
/* sample_probs_with_rng                               policies/mod.rs:45-61 */
static int FN(sample_probs)(const R* p, int A, uint32_t x) {
    R r = (R)(x >> 8) * (R)(1.0 / 16777216.0), acc = 0; int i;
    for (i = 0; i < A; i++) { acc = acc + p[i]; if (acc > r) return i; }
    return A - 1;
}
/* Policy::sample for the four policies; x[0]=explore draw, x[1]=random action, x[2]=tie/softmax u.
 *   EpsilonGreedy::sample   epsilon_greedy.rs:74-80  (gen_bool(eps) -> Random else Greedy)
 *   Random::sample          random.rs:43-45          (Uniform(0,A))
 *   Softmax::sample         softmax.rs:131-139
 * eps_thr = (uint32)(eps * 2^24): explore iff (x0 >> 8) < eps_thr  (integer, identical on device). */
int FN(orc_policy_sample)(int policy, const R* q, int A, uint32_t eps_thr, R tau, const uint32_t x[4]) {
    R p[ORC_MAX_ACTIONS];
    switch (policy) {
    case ORC_GREEDY:  return FN(greedy_sample)(q, A, x[2]);
    case ORC_EGREEDY:
        if ((x[0] >> 8) < eps_thr) return (int)orc_mulhi(x[1], (uint32_t)A);
        return FN(greedy_sample)(q, A, x[2]);
    case ORC_SOFTMAX:
        FN(orc_softmax_probs)(q, A, tau, p);
        return FN(sample_probs)(p, A, x[2]);
    default: return (int)orc_mulhi(x[1], (uint32_t)A);
    }
}
/* Policy::mode  greedy.rs:83 (find_max), epsilon_greedy.rs:82, softmax.rs:141-143 (argmax_first of probs),
 * random.rs:47 panics -> -1 */
int FN(orc_policy_mode)(int policy, const R* q, int A, R tau) {
    R p[ORC_MAX_ACTIONS];
    switch (policy) {
    case ORC_GREEDY: case ORC_EGREEDY: return FN(orc_find_max)(q, A, NULL);
    case ORC_SOFTMAX: FN(orc_softmax_probs)(q, A, tau, p); return FN(orc_argmax_first)(p, A);
    default: return -1;
    }
}

/* ------------------------------------------------------------------ */
/* TD control agents: Handler<&Transition>::handle                     */
/* ------------------------------------------------------------------ */

/* Computes the TD error delta for one transition with the PRE-update W and returns the
 * error sent to the approximator (delta, or alpha*delta for ExpectedSARSA).
 *   QLearning::handle      control/td/q_learning.rs:51-71
 *   SARSA::handle          control/td/sarsa.rs:53-75  (inner policy.sample with its own draws x_inner)
 *   ExpectedSARSA::handle  control/td/expected_sarsa.rs:45-66 */
R FN(orc_td_error)(const orc_agent* ag, const R* W, const R* s, int a, R r, const R* ns, int term,
                   const uint32_t x_inner[4], R* delta_out) {
    const orc_basis* b = &ag->basis; int A = ag->n_actions;
    R qsa = FN(orc_q_evaluate_index)(b, W, A, s, a);                   /* projection #1 */
    R delta;
    if (term) {
        delta = r - qsa;
    } else if (ag->algo == ORC_QLEARNING) {
        R q[ORC_MAX_ACTIONS], m;
        FN(orc_q_evaluate)(b, W, A, ns, q);                            /* projection #2 */
        FN(orc_find_max)(q, A, &m);
        delta = r + (R)ag->gamma * m - qsa;
    } else if (ag->algo == ORC_SARSA) {
        R q[ORC_MAX_ACTIONS]; int na;
        FN(orc_q_evaluate)(b, W, A, ns, q);
        na = FN(orc_policy_sample)(ag->apolicy, q, A, ag->aeps_thr, (R)ag->atau, x_inner);
        delta = r + (R)ag->gamma * q[na] - qsa;
    } else if (ag->algo == ORC_PAL) {
        /* PAL::handle  control/td/pal.rs:34-60: persistent advantage learning, the max of the advantage-learning error at
         * s and at s'; a* / na* are argmax_first (utils.rs:23-34); the error sent on is alpha * residual (:57). */
        R qs[ORC_MAX_ACTIONS], nqs[ORC_MAX_ACTIONS], td, al, alt; int as, nas;
        FN(orc_q_evaluate)(b, W, A, s, qs);
        FN(orc_q_evaluate)(b, W, A, ns, nqs);
        as = FN(orc_argmax_first)(qs, A); nas = FN(orc_argmax_first)(nqs, A);
        td = r + (R)ag->gamma * nqs[as] - qs[a];
        al = td - (R)ag->alpha * (qs[as] - qs[a]);
        alt = td - (R)ag->alpha * (nqs[nas] - nqs[a]);
        delta = (al > alt) ? al : alt;                                   /* f64::max */
    } else {
        R q[ORC_MAX_ACTIONS], p[ORC_MAX_ACTIONS], ev = 0; int i;
        FN(orc_q_evaluate)(b, W, A, ns, q);
        FN(orc_policy_probs)(ag->apolicy, q, A, (R)ag->aepsilon, (R)ag->atau, p);
        for (i = 0; i < A; i++) ev = ev + q[i] * p[i];                 /* fold(0.0, acc + q*p) */
        delta = r + (R)ag->gamma * ev - qsa;
    }
    if (delta_out) *delta_out = delta;
    return (ag->algo == ORC_EXPECTED_SARSA || ag->algo == ORC_PAL) ? (R)ag->alpha * delta : delta;
}
/* Full handle on per-env weights: error with pre-update W, then the column AXPY (projection #3). */
R FN(orc_handle)(const orc_agent* ag, R* W, const R* s, int a, R r, const R* ns, int term,
                 const uint32_t x_inner[4]) {
    R delta, e;
    e = FN(orc_td_error)(ag, W, s, a, r, ns, term, x_inner, &delta);
    FN(orc_q_update_index)(&ag->basis, W, ag->n_actions, s, a, (R)ag->lr, e);
    return delta;
}

/* Eligibility-trace agents on per-env weights.  Z has the shape of W ((F, A) row-major).
 *   SARSALambda::handle   control/td/sarsa_lambda.rs:53-98
 *   QLambda::handle       control/td/q_lambda.rs:56-99   (Watkins: trace reset when the action taken was not
 *                                                         argmax_first of Q(s,.), utils.rs:23-34)
 *   trace update rules    traces.rs:188-240  Accumulate: z = gl*z + g;  Saturate: clip(gl*z + g, -1, 1);
 *                                            Dutch: z = gl*(1-alpha)*z + g      (g = phi(s) in column a)
 *   weight update         Handler<ScaledGradientUpdate>: W += (alpha*residual) * Z  -- bypasses the optimiser
 *                         (fa/linear.rs:184-196)
 * Fourier bases: g = phi(s) in column a.  Tile coding (round 3; the reference's traces are generic over the gradient buffer,
 * traces.rs:6-12): g = 1.0 at the T active entries of column a, 0 elsewhere -- a dense trace table of W's shape per learner.
 * Returns the TD error. */
R FN(orc_handle_lambda)(const orc_agent* ag, R* W, R* Z, const R* s, int a, R r, const R* ns, int term,
                        const uint32_t x_inner[4]) {
    const orc_basis* b = &ag->basis; int A = ag->n_actions, F = orc_basis_nfeat(b), f, c;
    R qs[ORC_MAX_ACTIONS], qsa, residual, rate, scale;
    R* phi = (R*)calloc((size_t)F, sizeof(R));
    FN(orc_q_evaluate)(b, W, A, s, qs);
    qsa = qs[a];
    if (ag->algo == ORC_Q_LAMBDA && a != FN(orc_argmax_first)(qs, A)) memset(Z, 0, sizeof(R) * (size_t)F * A);
    if (b->kind == ORC_FOURIER) FN(orc_fourier_project)(b->order, b->dim, FN(basis_lo)(b), FN(basis_hi)(b), s, phi);
    else {                                                               /* unit activations at the T active indices */
        int idx[ORC_MAX_TILINGS], t; float sf[8];
        for (t = 0; t < b->dim; t++) sf[t] = (float)s[t];
        orc_tile_indices(b, sf, idx);
        for (t = 0; t < b->n_tilings; t++) phi[idx[t]] = (R)1.0;
    }
    rate = (R)orc_trace_rate(ag);
    for (f = 0; f < F; f++)
        for (c = 0; c < A; c++) {
            R g = (c == a) ? phi[f] : (R)0.0;
            R z = FN(fma_)(rate, Z[(size_t)f * A + c], g);                     /* rate * x + y */
            if (ag->trace == ORC_TRACE_SATURATE) { z = (z < (R)1.0) ? z : (R)1.0; z = (z > (R)-1.0) ? z : (R)-1.0; }
            Z[(size_t)f * A + c] = z;
        }
    free(phi);
    if (term) {
        residual = r - qsa;
    } else if (ag->algo == ORC_SARSA_LAMBDA) {
        R qn[ORC_MAX_ACTIONS]; int na;
        FN(orc_q_evaluate)(b, W, A, ns, qn);
        na = FN(orc_policy_sample)(ag->apolicy, qn, A, ag->aeps_thr, (R)ag->atau, x_inner);
        residual = r + (R)ag->gamma * qn[na] - qsa;
    } else {
        R qn[ORC_MAX_ACTIONS], m;
        FN(orc_q_evaluate)(b, W, A, ns, qn);
        FN(orc_find_max)(qn, A, &m);
        residual = r + (R)ag->gamma * m - qsa;
    }
    scale = (R)ag->alpha * residual;
    for (f = 0; f < F * A; f++) W[f] = FN(fma_)(scale, Z[f], W[f]);
    if (term) memset(Z, 0, sizeof(R) * (size_t)F * A);                        /* trace.reset() */
    return residual;
}

/* GreedyGQ::handle   control/td/greedy_gq.rs:73-141.  W = fa_q's weights (SGD(lr)), V = fa_td's weights (SGD(lr_td)),
 * both (F, A) row-major.  Order of the reference: qsa and td_est with the pre-update matrices; (na, max) =
 * fa_q.find_max(s') BEFORE any update; fa_q: column a moves by lr*td_error*phi(s), THEN column na by
 * lr*(-gamma*td_est)*phi(s') (non-terminal only); fa_td: column a moves by lr_td*(td_error - td_est)*phi(s).
 * Returns td_error. */
R FN(orc_handle_gq)(const orc_agent* ag, R* W, R* V, const R* s, int a, R r, const R* ns, int term) {
    const orc_basis* b = &ag->basis; int A = ag->n_actions;
    R qsa = FN(orc_q_evaluate_index)(b, W, A, s, a);
    R td_est = FN(orc_q_evaluate_index)(b, V, A, s, a);
    R td_error;
    if (term) {
        td_error = r - qsa;
        FN(orc_q_update_index)(b, W, A, s, a, (R)ag->lr, td_error);
    } else {
        R q[ORC_MAX_ACTIONS], m; int na;
        FN(orc_q_evaluate)(b, W, A, ns, q);
        na = FN(orc_find_max)(q, A, &m);
        td_error = r + (R)ag->gamma * m - qsa;
        FN(orc_q_update_index)(b, W, A, s, a, (R)ag->lr, td_error);
        FN(orc_q_update_index)(b, W, A, ns, na, (R)ag->lr, -(R)ag->gamma * td_est);
    }
    FN(orc_q_update_index)(b, V, A, s, a, (R)ag->lr_td, td_error - td_est);
    return td_error;
}

/* Prediction: TD::handle (prediction/td/td.rs:31-59) and TDLambda::handle (prediction/td/td_lambda.rs:41-78) on a
 * ScalarLFA (fa/linear.rs:201-251: V(s) = <phi(s), w>, grad = phi(s), StateUpdate -> SGD: w += lr*error*phi(s)).
 *   TD        : td = r + gamma*V(s') - V(s)  (terminal: r - V(s));  w += lr * td * phi(s)
 *   TDLambda  : trace.update(phi(s)) FIRST (traces.rs:188-240, same rules as the control agents), then
 *               w += td * trace  -- `ScaledGradientUpdate { alpha: td_error, jacobian: &trace }` (td_lambda.rs:59-62, :71-74):
 *               the step is the TD error itself, no learning rate (fa/linear.rs:184-196 bypasses the optimiser);
 *               a terminal transition then resets the trace (:64).
 * Fourier bases: phi dense.  Tile coding (round 3): V(s) = the sum of the T active weights in tiling order, grad = 1.0 at the T
 * active entries.  w, z: F values.  Returns the TD error. */
static R FN(v_tile)(const orc_basis* b, const R* w, const R* s, int* idx) {
    int t; float sf[8]; R acc = 0;
    for (t = 0; t < b->dim; t++) sf[t] = (float)s[t];
    orc_tile_indices(b, sf, idx);
    for (t = 0; t < b->n_tilings; t++) acc = acc + w[idx[t]];
    return acc;
}
R FN(orc_v_evaluate)(const orc_basis* b, const R* w, const R* s) {
    int F = orc_basis_nfeat(b); R v;
    R* phi;
    if (b->kind != ORC_FOURIER) { int idx[ORC_MAX_TILINGS]; return FN(v_tile)(b, w, s, idx); }
    phi = (R*)malloc(sizeof(R) * (size_t)F);
    FN(orc_fourier_project)(b->order, b->dim, FN(basis_lo)(b), FN(basis_hi)(b), s, phi);
    FN(dot_columns)(phi, w, 1, F, &v);
    free(phi);
    return v;
}
R FN(orc_handle_td)(const orc_agent* ag, R* w, R* z, const R* s, R r, const R* ns, int term) {
    const orc_basis* b = &ag->basis; int F = orc_basis_nfeat(b), f;
    R pred, td, rate;
    R* phi = (R*)calloc((size_t)F, sizeof(R));
    if (b->kind == ORC_FOURIER) {
        FN(orc_fourier_project)(b->order, b->dim, FN(basis_lo)(b), FN(basis_hi)(b), s, phi);
        FN(dot_columns)(phi, w, 1, F, &pred);
    } else {
        int idx[ORC_MAX_TILINGS], t;
        pred = FN(v_tile)(b, w, s, idx);
        for (t = 0; t < b->n_tilings; t++) phi[idx[t]] = (R)1.0;
    }
    if (ag->algo == ORC_TD_LAMBDA) {
        rate = (R)orc_trace_rate(ag);
        for (f = 0; f < F; f++) {
            R v = FN(fma_)(rate, z[f], phi[f]);
            if (ag->trace == ORC_TRACE_SATURATE) { v = (v < (R)1.0) ? v : (R)1.0; v = (v > (R)-1.0) ? v : (R)-1.0; }
            z[f] = v;
        }
    }
    td = term ? r - pred : r + (R)ag->gamma * FN(orc_v_evaluate)(b, w, ns) - pred;
    if (ag->algo == ORC_TD_LAMBDA) {
        for (f = 0; f < F; f++) w[f] = FN(fma_)(td, z[f], w[f]);
        if (term) memset(z, 0, sizeof(R) * (size_t)F);
    } else {
        R scale = (R)ag->lr * td;
        for (f = 0; f < F; f++) w[f] = FN(fma_)(scale, phi[f], w[f]);
    }
    free(phi);
    return td;
}

/* QSigma: the n-step Q(sigma) agent (De Asis et al. 2017).
 *   QSigma::handle         control/td/q_sigma.rs:138-201
 *   QSigma::update_backup  control/td/q_sigma.rs:107-128
 *   Backup::propagate      control/td/q_sigma.rs:46-63
 * DEVIATION (the only one): propagate's loop `for k in 0..n_steps` reads entries[k + 1] (:52-53) although update_backup calls
 * it with exactly n_steps entries (:113-114): the last iteration indexes out of bounds and the reference PANICS at the first
 * full backup.  That iteration needs entries[k + 1] only to update z (:56), a value never used again; the restatement keeps
 * every in-bounds operation (g += z*residual_k, the isr factor of all n entries) and drops that dead z update.
 *   pi = 1/|maxima| if a' is a maximum of Q(s',.) else 0 (:162-166); exp_nqs is argmaxima's running maximum (:161)
 *   mu = policy.evaluate((s', a')) (:167): greedy.rs:46-60 / epsilon_greedy.rs:49-63 probabilities, random.rs:28-32, and
 *        for Softmax the raw action value (softmax.rs:84-92 -- its Function<(S, A)> is the Q-value, not a probability). */
typedef struct { R s[8]; int a; R q, residual, pi, mu; } FN(qs_entry);
typedef struct { int n_steps, head, len; FN(qs_entry) e[ORC_MAX_NSTEPS]; } FN(qs_backup);
void* FN(orc_qsigma_new)(int n_steps) {
    FN(qs_backup)* b = (FN(qs_backup)*)calloc(1, sizeof(*b));
    b->n_steps = n_steps < 1 ? 1 : (n_steps > ORC_MAX_NSTEPS ? ORC_MAX_NSTEPS : n_steps);
    return b;
}
void FN(orc_qsigma_free)(void* backup) { free(backup); }
int FN(orc_qsigma_len)(const void* backup) { return ((const FN(qs_backup)*)backup)->len; }
static R FN(policy_eval_sa)(int policy, const R* q, int A, R eps, int a) {
    int ixs[ORC_MAX_ACTIONS], n, i, in = 0; R pg;
    if (policy == ORC_SOFTMAX) return q[a];
    if (policy == ORC_RANDOM) return (R)1.0 / (R)A;
    n = FN(orc_argmaxima)(q, A, ixs, NULL);
    for (i = 0; i < n; i++) if (ixs[i] == a) in = 1;
    pg = in ? (R)1.0 / (R)(n < 1 ? 1 : n) : (R)0.0;
    if (policy == ORC_GREEDY) return pg;
    return eps / (R)A + ((R)1.0 - eps) * pg;
}
R FN(orc_handle_qsigma)(const orc_agent* ag, R* W, void* backup, const R* s, int a, R r, const R* ns, int term,
                        const uint32_t x_inner[4]) {
    FN(qs_backup)* bk = (FN(qs_backup)*)backup; const orc_basis* b = &ag->basis;
    int A = ag->n_actions, D = b->dim, n = bk->n_steps, d, k;
    const R sigma = (R)ag->sigma, gamma = (R)ag->gamma;
    R qa = FN(orc_q_evaluate_index)(b, W, A, s, a);
    R residual, pi, mu;
    FN(qs_entry)* e;
    if (term) {
        residual = r - qa; pi = (R)0.0; mu = (R)1.0;
    } else {
        R nqs[ORC_MAX_ACTIONS], exp_nqs; int ixs[ORC_MAX_ACTIONS], nmax, na, in = 0, i;
        FN(orc_q_evaluate)(b, W, A, ns, nqs);
        na = FN(orc_policy_sample)(ag->apolicy, nqs, A, ag->aeps_thr, (R)ag->atau, x_inner);
        nmax = FN(orc_argmaxima)(nqs, A, ixs, &exp_nqs);
        for (i = 0; i < nmax; i++) if (ixs[i] == na) in = 1;
        pi = in ? (R)1.0 / (R)nmax : (R)0.0;
        mu = FN(policy_eval_sa)(ag->apolicy, nqs, A, (R)ag->aepsilon, na);
        residual = r + gamma * (sigma * nqs[na] + ((R)1.0 - sigma) * exp_nqs) - qa;
    }
    e = &bk->e[(bk->head + bk->len) % n];
    for (d = 0; d < D; d++) e->s[d] = s[d];
    e->a = a; e->q = qa; e->residual = residual; e->pi = pi; e->mu = mu;
    bk->len += 1;
    if (bk->len >= n) {
        R g = bk->e[bk->head].q, z = (R)1.0, isr = (R)1.0, qsa, err;
        FN(qs_entry)* anchor;
        for (k = 0; k < n; k++) {
            const FN(qs_entry)* b1 = &bk->e[(bk->head + k) % n];
            g += z * b1->residual;
            if (k + 1 < n) {
                const FN(qs_entry)* b2 = &bk->e[(bk->head + k + 1) % n];
                z *= gamma * (((R)1.0 - sigma) * b2->pi + sigma);
            }
            isr *= (R)1.0 - sigma + sigma * b1->pi / b1->mu;
        }
        anchor = &bk->e[bk->head];
        bk->head = (bk->head + 1) % n; bk->len -= 1;
        qsa = FN(orc_q_evaluate_index)(b, W, A, anchor->s, anchor->a);
        err = (R)ag->alpha * isr * (g - qsa);
        FN(orc_q_update_index)(b, W, A, anchor->s, anchor->a, (R)ag->lr, err);
    }
    if (term) bk->len = 0;
    return residual;
}

/* ------------------------------------------------------------------ */
/* Vectorised driver loop (examples/q_learning.rs:34-55 x N envs)      */
/* ------------------------------------------------------------------ */

typedef struct {
    orc_agent ag;
    int64_t n_envs;
    R* state;        /* [N][D] */
    int32_t* action; /* [N]    */
    uint32_t* ep_step;
    R* W;            /* per-env: [N][F][A]; shared: [F][A] */
    R* Z;            /* per-env [N][F][A]: eligibility traces (lambda agents) or fa_td weights (GreedyGQ) */
    uint64_t t;      /* global batch-step counter */
    FN(qs_backup)* qs; /* [N] QSigma backups */
    R* eps;          /* [N] EpsilonGreedy.epsilon of every learner (the pub field, epsilon_greedy.rs:19): one value each, decayed per
                      * episode when the agent carries a schedule (orc_agent.eps_decay) */
    R* qc;           /* [N][A] Q(s,.) of the current state carried between orc_run_train_dev calls (the device's qcache) */
    int q_valid;     /* 0: qc is stale -> recompute from W at the next orc_run_train_dev call */
    /* teacher forcing (orc_run_teacher): when round32 is set every successor state is rounded to fp32 before anything uses it, so the
     * transitions this run handles are fp32-representable and can be replayed, value for value, through the device's
     * Handler::handle; tape_* (may be NULL) receive the batch-step's transitions as the agent saw them */
    /* SARSALambda / QLambda over ONE shared tile table: every learner's sparse trace (orc_run_train_sparse_lambda) */
    uint32_t* sp_keys; R* sp_vals; int* sp_len;
    int round32;
    R* tape_from; int32_t* tape_act; R* tape_rew; R* tape_to; uint8_t* tape_term; R* tape_td;
} FN(orc_run);

static R* FN(run_W)(FN(orc_run)* run, int64_t i) {
    size_t FA = (size_t)orc_basis_nfeat(&run->ag.basis) * (size_t)ORC_N_OUT(&run->ag);
    return run->ag.shared_w ? run->W : run->W + (size_t)i * FA;
}

/* Q(s,.) of learner i for the behaviour policy; prediction agents have no Q (their policy is Random, which ignores it) */
static void FN(run_q)(FN(orc_run)* run, int64_t i, const R* s, R* q) {
    int A = run->ag.n_actions, k;
    if (ORC_IS_PRED(run->ag.algo)) { for (k = 0; k < A; k++) q[k] = (R)0.0; return; }
    FN(orc_q_evaluate)(&run->ag.basis, FN(run_W)(run, i), A, s, q);
}

/* ---- the per-learner epsilon schedule ------------------------------------------------------------------------------------------
 * sched: the agent carries one.  The value lives in R (f64: the reference's; float instantiations: the device's fp32 field), and
 * gen_bool's threshold is taken from it as orc_eps_threshold takes it from the configured value (R * 2^24 is exact). */
static int FN(eps_sched)(const orc_agent* ag) { return ag->eps_decay > 0.0 && ag->eps_decay != 1.0; }
/* the agent as learner i sees it: its own epsilon in the behaviour policy -- and in the agent's policy when that is the same object */
static orc_agent FN(agent_of)(const FN(orc_run)* run, int64_t i) {
    orc_agent a = run->ag;
    if (FN(eps_sched)(&run->ag)) {
        a.epsilon = (double)run->eps[i]; a.eps_thr = orc_eps_threshold((double)run->eps[i]);
        if (a.apol_same) { a.aepsilon = a.epsilon; a.aeps_thr = a.eps_thr; }
    }
    return a;
}
/* an episode of learner i has ended: agent.policy.epsilon *= decay  (examples/sarsa_lambda.rs:68), floored */
static void FN(eps_episode_end)(FN(orc_run)* run, int64_t i) {
    if (FN(eps_sched)(&run->ag)) {
        R e = run->eps[i] * (R)run->ag.eps_decay;
        run->eps[i] = (e > (R)run->ag.eps_min) ? e : (R)run->ag.eps_min;
    }
}
R* FN(orc_run_eps)(void* h) { return ((FN(orc_run)*)h)->eps; }

void* FN(orc_run_create)(const orc_agent* ag, int64_t n_envs) {
    FN(orc_run)* run = (FN(orc_run)*)calloc(1, sizeof(*run));
    size_t FA = (size_t)orc_basis_nfeat(&ag->basis) * (size_t)ORC_N_OUT(ag);
    run->ag = *ag; run->n_envs = n_envs;
    run->state = (R*)calloc((size_t)n_envs * (size_t)ag->basis.dim, sizeof(R));
    run->action = (int32_t*)calloc((size_t)n_envs, sizeof(int32_t));
    run->ep_step = (uint32_t*)calloc((size_t)n_envs, sizeof(uint32_t));
    run->W = (R*)calloc(ag->shared_w ? FA : FA * (size_t)n_envs, sizeof(R));   /* LFA::vector zero-inits */
    run->Z = ORC_HAS_AUX(ag->algo) ? (R*)calloc(FA * (size_t)n_envs, sizeof(R)) : NULL;
    run->t = 0;
    run->qc = (R*)calloc((size_t)n_envs * ORC_MAX_ACTIONS, sizeof(R)); run->q_valid = 0;
    run->eps = (R*)calloc((size_t)n_envs, sizeof(R));
    { int64_t i; for (i = 0; i < n_envs; i++) run->eps[i] = (R)ag->epsilon; }
    run->qs = NULL;
    if (ag->algo == ORC_Q_SIGMA) {
        int64_t i; run->qs = (FN(qs_backup)*)calloc((size_t)n_envs, sizeof(FN(qs_backup)));
        for (i = 0; i < n_envs; i++) run->qs[i].n_steps = ag->n_steps < 1 ? 1 : (ag->n_steps > ORC_MAX_NSTEPS ? ORC_MAX_NSTEPS : ag->n_steps);
    }
    return run;
}
void FN(orc_run_destroy)(void* h) {
    FN(orc_run)* run = (FN(orc_run)*)h;
    free(run->state); free(run->action); free(run->ep_step); free(run->W); free(run->Z); free(run->qc); free(run->eps); free(run->qs);
    free(run->sp_keys); free(run->sp_vals); free(run->sp_len); free(run);
}
R* FN(orc_run_state)(void* h) { return ((FN(orc_run)*)h)->state; }
int32_t* FN(orc_run_action)(void* h) { return ((FN(orc_run)*)h)->action; }
uint32_t* FN(orc_run_ep_step)(void* h) { return ((FN(orc_run)*)h)->ep_step; }
R* FN(orc_run_weights)(void* h) { return ((FN(orc_run)*)h)->W; }
R* FN(orc_run_traces)(void* h) { return ((FN(orc_run)*)h)->Z; }
uint64_t FN(orc_run_t)(void* h) { return ((FN(orc_run)*)h)->t; }
void FN(orc_run_set_epsilon)(void* h, double eps) {
    FN(orc_run)* run = (FN(orc_run)*)h; int64_t i;
    run->ag.epsilon = eps; run->ag.eps_thr = orc_eps_threshold(eps);
    if (run->ag.apol_same) { run->ag.aepsilon = eps; run->ag.aeps_thr = run->ag.eps_thr; }     /* one policy object: the agent's moves too */
    for (i = 0; i < run->n_envs; i++) run->eps[i] = (R)eps;
}

/* per-episode `Domain::default()` + initial `policy.sample`   examples/q_learning.rs:37-38 */
void FN(orc_run_reset)(void* h) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag;
    int D = ag->basis.dim, A = ag->n_actions; int64_t i;
    run->q_valid = 0;
    /* QSigma: fresh episodes start from an EMPTY n-step backup, as after a terminal transition (q_sigma.rs:154) -- what rsrl_hip_reset does: entries of
     * the abandoned trajectories must not enter the first anchor updates of the new ones (found unrestated by tests/fuzz_parity.py, round 5) */
    if (run->qs) for (i = 0; i < run->n_envs; i++) run->qs[i].len = 0;
    for (i = 0; i < run->n_envs; i++) {
        R q[ORC_MAX_ACTIONS]; uint32_t x[4];
        R* s = run->state + (size_t)i * D;
        FN(orc_domain_reset)(ag->domain, s);
        FN(run_q)(run, i, s, q);
        orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t, ORC_BLK_INIT, x);
        run->action[i] = FN(orc_policy_sample)(ag->policy, q, A, FN(agent_of)(run, i).eps_thr, (R)ag->tau, x);
        run->ep_step[i] = 0;
    }
}

/* n_steps batch-steps of the driver loop; order of operations: SURVEY.md Appendix A.7.
 * per-env W : transition -> handle (pre-update W) -> sample (post-update W) -> maybe reset+sample
 * shared  W : all envs compute e_i, phi(s_i) on W_t; W_{t+1} = W_t + lr*sum_i e_i phi(s_i) x onehot(a_i)
 *             (accumulated in env order); then all envs sample with W_{t+1}. N=1 == reference rule. */
/* dw_hook (may be NULL): called once per batch-step in shared-W mode on the local delta (F*A values) before it is
 * applied -- the place where a multi-rank run all-reduces the delta (tests/test_distributed_cpu.py). */
void FN(orc_run_train_hook)(void* h, int64_t n_steps, orc_stats* st, void (*dw_hook)(R* dW, int n, void* user), void* user) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag;
    int D = ag->basis.dim, A = ag->n_actions, F = orc_basis_nfeat(&ag->basis);
    int64_t N = run->n_envs, i, k;
    R* ns_all = (R*)malloc(sizeof(R) * (size_t)N * D);
    uint8_t* term_all = (uint8_t*)malloc((size_t)N);
    R* dW = ag->shared_w ? (R*)malloc(sizeof(R) * (size_t)F * A) : NULL;
    /* Shared tile coding in the device's arithmetic type (float instantiations): the mini-batch delta is accumulated in 64-bit
     * FIXED POINT, as the device does (rsrl_amd/csrc/models.hpp TileModel::block_accumulate, k_apply_rep): a term lr*e is scaled
     * by the exact power of two 1/lsb, lsb = 2^(floor(log2 lr) - 28), clamped to +-2^42 and rounded to an integer; the sum over
     * the batch is exact and order-independent and converts back with one rounding -- so this loop and the HIP path agree bit
     * for bit whatever order the device's atomics retire in.  (f64 keeps the plain sum: the reference rule.) */
    const int fixed = ag->shared_w && ag->basis.kind == ORC_TILE && sizeof(R) == 4;
    int64_t* qacc = fixed ? (int64_t*)malloc(sizeof(int64_t) * (size_t)F * A) : NULL;
    float lsb_f = 1.0f, inv_lsb_f = 1.0f;
    orc_stats acc; memset(&acc, 0, sizeof(acc));
    if (fixed) {
        const float lrf = (float)ag->lr; uint32_t u, eb, ex, v;
        memcpy(&u, &lrf, 4); eb = (u >> 23) & 0xffu; ex = (eb < 30u ? 30u : eb) - 28u;
        v = ex << 23; memcpy(&lsb_f, &v, 4); v = (254u - ex) << 23; memcpy(&inv_lsb_f, &v, 4);
    }
    for (k = 0; k < n_steps; k++, run->t++) {
        if (dW) memset(dW, 0, sizeof(R) * (size_t)F * A);
        if (qacc) memset(qacc, 0, sizeof(int64_t) * (size_t)F * A);
        for (i = 0; i < N; i++) {
            R* s = run->state + (size_t)i * D; R* ns = ns_all + (size_t)i * D;
            R r, delta; int a = run->action[i], term; uint32_t xi[4];
            const orc_agent agl = FN(agent_of)(run, i); const orc_agent* ag = &agl;      /* (shadows: learner i's own epsilon, when scheduled) */
            memcpy(ns, s, sizeof(R) * D);
            term = FN(orc_domain_step)(ag->domain, ns, a, &r);              /* Domain::transition lib.rs:436-446 */
            term_all[i] = (uint8_t)term;
            if (run->round32) { int d; for (d = 0; d < D; d++) ns[d] = (R)(float)ns[d]; }
            if (run->tape_from) {
                memcpy(run->tape_from + (size_t)i * D, s, sizeof(R) * D); memcpy(run->tape_to + (size_t)i * D, ns, sizeof(R) * D);
                run->tape_act[i] = a; run->tape_rew[i] = r; run->tape_term[i] = (uint8_t)term;
            }
            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t, ORC_BLK_INNER, xi);
            if (ORC_IS_PRED(ag->algo)) {
                delta = FN(orc_handle_td)(ag, FN(run_W)(run, i), run->Z ? run->Z + (size_t)i * F : NULL, s, r, ns, term);
            } else if (ORC_IS_LAMBDA(ag->algo)) {
                delta = FN(orc_handle_lambda)(ag, FN(run_W)(run, i), run->Z + (size_t)i * F * A, s, a, r, ns, term, xi);
            } else if (ag->algo == ORC_Q_SIGMA) {
                delta = FN(orc_handle_qsigma)(ag, FN(run_W)(run, i), &run->qs[i], s, a, r, ns, term, xi);
            } else if (ag->algo == ORC_GREEDY_GQ) {
                delta = FN(orc_handle_gq)(ag, FN(run_W)(run, i), run->Z + (size_t)i * F * A, s, a, r, ns, term);
            } else if (!ag->shared_w) {
                delta = FN(orc_handle)(ag, FN(run_W)(run, i), s, a, r, ns, term, xi);
            } else {
                R e = FN(orc_td_error)(ag, run->W, s, a, r, ns, term, xi, &delta);
                if (fixed) {
                    int idx[ORC_MAX_TILINGS], tt; float sf[8], sc = (float)((R)ag->lr * e) * inv_lsb_f;
                    for (tt = 0; tt < D; tt++) sf[tt] = (float)s[tt];
                    orc_tile_indices(&ag->basis, sf, idx);
                    sc = sc < -4.398046511104e12f ? -4.398046511104e12f : (sc > 4.398046511104e12f ? 4.398046511104e12f : sc);
                    for (tt = 0; tt < ag->basis.n_tilings; tt++) qacc[(size_t)idx[tt] * A + a] += (int64_t)rintf(sc);
                } else
                FN(orc_q_update_index)(&ag->basis, dW, A, s, a, (R)ag->lr, e);   /* dW += lr*e*phi(s) on column a */
            }
            acc.sum_abs_td_error += fabs((double)delta);
            acc.sum_reward += (double)r;
            if (run->tape_td) run->tape_td[i] = delta;
        }
        if (fixed) { int j; for (j = 0; j < F * A; j++) dW[j] = (R)((float)qacc[j] * lsb_f); }
        if (dW && dw_hook) dw_hook(dW, F * A, user);
        if (dW) { int j; for (j = 0; j < F * A; j++) run->W[j] += dW[j]; }
        for (i = 0; i < N; i++) {
            R* s = run->state + (size_t)i * D; R* ns = ns_all + (size_t)i * D;
            R q[ORC_MAX_ACTIONS]; uint32_t x[4]; int na;
            FN(run_q)(run, i, ns, q);                                       /* projection #4, UPDATED W */
            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t, ORC_BLK_STEP, x);
            na = FN(orc_policy_sample)(ag->policy, q, A, FN(agent_of)(run, i).eps_thr, (R)ag->tau, x);
            run->ep_step[i] += 1;
            acc.env_steps += 1;
            if (term_all[i] || (ag->max_episode_steps > 0 && run->ep_step[i] >= ag->max_episode_steps)) {
                acc.episodes += 1;
                if (!term_all[i]) acc.episodes_truncated += 1;
                acc.sum_episode_steps += run->ep_step[i];
                FN(eps_episode_end)(run, i);                                /* agent.policy.epsilon *= 0.995  examples/sarsa_lambda.rs:68 */
                FN(orc_domain_reset)(ag->domain, ns);
                FN(run_q)(run, i, ns, q);
                orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t, ORC_BLK_RESET, x);
                na = FN(orc_policy_sample)(ag->policy, q, A, FN(agent_of)(run, i).eps_thr, (R)ag->tau, x);
                run->ep_step[i] = 0;
            }
            memcpy(s, ns, sizeof(R) * D);
            run->action[i] = na;
        }
    }
    free(ns_all); free(term_all); free(dW); free(qacc);
    if (st) *st = acc;
}

void FN(orc_run_train)(void* h, int64_t n_steps, orc_stats* st) { FN(orc_run_train_hook)(h, n_steps, st, NULL, NULL); }

/* ONE batch-step of orc_run_train as a TEACHER: successor states are rounded to fp32 (the reset states are fp32-exact), and the
 * transitions the agents handled -- from [N][D], action, reward, to [N][D], terminal flag -- and their TD errors are written out.
 * A test replays them through the device's rsrl_hip_handle (whose k-th call uses the same agent-side draws as this run's k-th
 * batch-step) and compares weights: both sides have then learned from identical fp32-representable inputs, which is the
 * "teacher-forced" comparison of SURVEY 8(d).  The run's own arithmetic stays R throughout. */
void FN(orc_run_teacher)(void* h, orc_stats* st, R* from, int32_t* act, R* rew, R* to, uint8_t* term, R* td) {
    FN(orc_run)* run = (FN(orc_run)*)h;
    run->round32 = 1;
    run->tape_from = from; run->tape_act = act; run->tape_rew = rew; run->tape_to = to; run->tape_term = term; run->tape_td = td;
    FN(orc_run_train_hook)(h, 1, st, NULL, NULL);
    run->tape_from = NULL; run->tape_act = NULL; run->tape_rew = NULL; run->tape_to = NULL; run->tape_term = NULL; run->tape_td = NULL;
    run->round32 = 0;
}

/* The same driver loop with the work the reference repeats removed -- phi(s) and Q(s,.) carried over from the previous
 * step, phi(s') projected once, no heap traffic, learners walked one after the other (they are independent): the
 * "optimised CPU" figure of SURVEY 8(d), so that the GPU/CPU ratio is not inflated by the reference's call pattern.
 * Same feature values and dot-product order as orc_run_train => identical states, actions and weights (tested).
 * QLearning on a Fourier basis with per-env weights only; returns -1 otherwise. */
int FN(orc_run_train_fast)(void* h, int64_t n_steps, orc_stats* st) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag; const orc_basis* b = &ag->basis;
    int D = b->dim, A = ag->n_actions, F = orc_basis_nfeat(b), f, d;
    int64_t N = run->n_envs, i, k;
    R *phi_s, *phi_n, *tmp;
    orc_stats acc; memset(&acc, 0, sizeof(acc));
    if (ag->algo != ORC_QLEARNING || b->kind != ORC_FOURIER || ag->shared_w || FN(eps_sched)(ag)) return -1;
    phi_s = (R*)malloc(sizeof(R) * (size_t)F); phi_n = (R*)malloc(sizeof(R) * (size_t)F);
    for (i = 0; i < N; i++) {
        R* s = run->state + (size_t)i * D; R* W = FN(run_W)(run, i);
        R q_s[ORC_MAX_ACTIONS], q_n[ORC_MAX_ACTIONS], ns[8];
        int a = run->action[i]; uint32_t ep = run->ep_step[i];
        FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), s, phi_s);
        FN(dot_columns)(phi_s, W, A, F, q_s);
        for (k = 0; k < n_steps; k++) {
            R r, m, delta, scale; int term, na, done; uint32_t x[4];
            for (d = 0; d < D; d++) ns[d] = s[d];
            term = FN(orc_domain_step)(ag->domain, ns, a, &r);
            ep += 1;
            done = term || (ag->max_episode_steps > 0 && ep >= ag->max_episode_steps);
            if (!term) {
                FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), ns, phi_n);
                FN(dot_columns)(phi_n, W, A, F, q_n);
                FN(orc_find_max)(q_n, A, &m);
                delta = r + (R)ag->gamma * m - q_s[a];
            } else {
                delta = r - q_s[a];
            }
            scale = (R)ag->lr * delta;
            for (f = 0; f < F; f++) W[(size_t)f * A + a] = FN(fma_)(scale, phi_s[f], W[(size_t)f * A + a]);
            acc.sum_abs_td_error += fabs((double)delta); acc.sum_reward += (double)r; acc.env_steps += 1;
            if (done) {
                acc.episodes += 1; if (!term) acc.episodes_truncated += 1; acc.sum_episode_steps += ep;
                FN(orc_domain_reset)(ag->domain, ns);
                FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), ns, phi_n);
                ep = 0;
            }
            FN(dot_columns)(phi_n, W, A, F, q_n);                          /* policy.sample sees the UPDATED weights */
            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t + (uint64_t)k, done ? ORC_BLK_RESET : ORC_BLK_STEP, x);
            na = FN(orc_policy_sample)(ag->policy, q_n, A, ag->eps_thr, (R)ag->tau, x);
            for (d = 0; d < D; d++) s[d] = ns[d];
            for (d = 0; d < A; d++) q_s[d] = q_n[d];
            tmp = phi_s; phi_s = phi_n; phi_n = tmp;
            a = na;
        }
        run->action[i] = a; run->ep_step[i] = ep;
    }
    run->t += (uint64_t)n_steps;
    free(phi_s); free(phi_n);
    if (st) *st = acc;
    return 0;
}

/* TD error of the one-step control agents from Q(s,.) and Q(s',.) (both with the PRE-update weights), as the device's
 * td_dispatch evaluates it; *e_out = the error sent on to the approximator (alpha*delta for ExpectedSARSA and PAL).
 *   q_learning.rs:51-71, sarsa.rs:53-75, expected_sarsa.rs:45-66, pal.rs:34-60 */
static R FN(td_from_q)(const orc_agent* ag, const R* q_s, int a, const R* q_n, R r, int term, const uint32_t xin[4], R* e_out) {
    int A = ag->n_actions, j; R delta;
    if (term) {
        delta = r - q_s[a];
    } else if (ag->algo == ORC_QLEARNING) {
        R m; FN(orc_find_max)(q_n, A, &m);
        delta = r + (R)ag->gamma * m - q_s[a];
    } else if (ag->algo == ORC_SARSA) {
        int ia = FN(orc_policy_sample)(ag->apolicy, q_n, A, ag->aeps_thr, (R)ag->atau, xin);
        delta = r + (R)ag->gamma * q_n[ia] - q_s[a];
    } else if (ag->algo == ORC_PAL) {
        int as = FN(orc_argmax_first)(q_s, A), nas = FN(orc_argmax_first)(q_n, A);
        R td = r + (R)ag->gamma * q_n[as] - q_s[a];
        R al = td - (R)ag->alpha * (q_s[as] - q_s[a]);
        R alt = td - (R)ag->alpha * (q_n[nas] - q_n[a]);
        delta = (al > alt) ? al : alt;
    } else {
        R p[ORC_MAX_ACTIONS], ev = 0;
        FN(orc_policy_probs)(ag->apolicy, q_n, A, (R)ag->aepsilon, (R)ag->atau, p);
        for (j = 0; j < A; j++) ev = ev + q_n[j] * p[j];
        delta = r + (R)ag->gamma * ev - q_s[a];
    }
    *e_out = (ag->algo == ORC_EXPECTED_SARSA || ag->algo == ORC_PAL) ? (R)ag->alpha * delta : delta;
    return delta;
}

/* The driver loop in the DEVICE's evaluation order (rsrl_amd/csrc/kernels_reg.hpp: k_train_reg / k_step_reg_lm), for bitwise
 * comparison with the HIP path (instantiation _f32d) and as a CPU cross-check of that order against orc_run_train
 * (_f32 / _f64: same trajectories up to rounding).  Differences from orc_run_train, all value-preserving up to rounding:
 *   - phi(s) and Q(s,.) are carried from the previous step instead of being projected / evaluated again;
 *   - a TERMINAL transition needs no Q(s') (delta = r - Q(s,a)): the projection slot takes s0 = Domain::default();
 *   - Q(s',.) with the UPDATED weights = Q(s',.) with the old ones + (lr*e) * <phi(s), phi(s')> in column a (the update
 *     is rank-1 in that column); only a step-cap truncation re-evaluates Q(s0,.) from W;
 *   - Q(s,.) is carried across calls (qc / q_valid), as the device carries it across launches.
 * QLearning / SARSA / ExpectedSARSA / PAL on a Fourier basis with per-env weights; returns -1 otherwise. */
void FN(orc_run_invalidate_q)(void* h) { ((FN(orc_run)*)h)->q_valid = 0; }
int FN(orc_run_train_dev)(void* h, int64_t n_steps, orc_stats* st) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag; const orc_basis* b = &ag->basis;
    int D = b->dim, A = ag->n_actions, F = orc_basis_nfeat(b), f, d, j;
    int64_t N = run->n_envs, i, k;
    R *phi_s, *phi_n, *tmp;
    orc_stats acc; memset(&acc, 0, sizeof(acc));
    if (b->kind != ORC_FOURIER || ag->shared_w ||
        !(ag->algo == ORC_QLEARNING || ag->algo == ORC_SARSA || ag->algo == ORC_EXPECTED_SARSA || ag->algo == ORC_PAL)) return -1;
    phi_s = (R*)malloc(sizeof(R) * (size_t)F); phi_n = (R*)malloc(sizeof(R) * (size_t)F);
    for (i = 0; i < N; i++) {
        R* s = run->state + (size_t)i * D; R* W = FN(run_W)(run, i); R* qc = run->qc + (size_t)i * ORC_MAX_ACTIONS;
        R q_s[ORC_MAX_ACTIONS], q_n[ORC_MAX_ACTIONS], ns[8];
        int a = run->action[i]; uint32_t ep = run->ep_step[i];
        orc_agent agl = FN(agent_of)(run, i); const orc_agent* ag = &agl;          /* (shadows: learner i's own epsilon, when scheduled) */
        FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), s, phi_s);
        if (run->q_valid) { for (j = 0; j < A; j++) q_s[j] = qc[j]; }
        else FN(dot_columns)(phi_s, W, A, F, q_s);
        for (k = 0; k < n_steps; k++) {
            const uint64_t t = run->t + (uint64_t)k;
            R r, delta, e, scale, dot; int term, trunc, na; uint32_t x[4], xin[4] = { 0, 0, 0, 0 };
            for (d = 0; d < D; d++) ns[d] = s[d];
            term = FN(orc_domain_step)(ag->domain, ns, a, &r);
            ep += 1;
            trunc = !term && ag->max_episode_steps > 0 && ep >= ag->max_episode_steps;
            if (term) FN(orc_domain_reset)(ag->domain, ns);
            FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), ns, phi_n);
            FN(dot_columns)(phi_n, W, A, F, q_n);                          /* PRE-update weights */
            if (ag->algo == ORC_SARSA) orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, ORC_BLK_INNER, xin);
            delta = FN(td_from_q)(ag, q_s, a, q_n, r, term, xin, &e);
            scale = (R)ag->lr * e;
            for (f = 0; f < F; f++) W[(size_t)f * A + a] = FN(fma_)(scale, phi_s[f], W[(size_t)f * A + a]);
            {   /* <phi(s), phi(s')> in the 4-way interleaved order of every dot product on this path */
                R pa[4] = { 0, 0, 0, 0 };
                for (f = 0; f < F; f++) pa[f & 3] = FN(fma_)(phi_s[f], phi_n[f], pa[f & 3]);
                dot = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            }
            q_n[a] = FN(fma_)(scale, dot, q_n[a]);
            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, term ? ORC_BLK_RESET : ORC_BLK_STEP, x);
            if (term || trunc) { FN(eps_episode_end)(run, i); agl = FN(agent_of)(run, i); }      /* examples/sarsa_lambda.rs:68 */
            na = FN(orc_policy_sample)(ag->policy, q_n, A, ag->eps_thr, (R)ag->tau, x);
            acc.sum_abs_td_error += fabs((double)delta); acc.sum_reward += (double)r; acc.env_steps += 1;
            if (term) { acc.episodes += 1; acc.sum_episode_steps += ep; ep = 0; }
            if (trunc) {
                acc.episodes += 1; acc.episodes_truncated += 1; acc.sum_episode_steps += ep; ep = 0;
                FN(orc_domain_reset)(ag->domain, ns);
                FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), ns, phi_n);
                FN(dot_columns)(phi_n, W, A, F, q_n);                      /* UPDATED weights */
                orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, ORC_BLK_RESET, x);
                na = FN(orc_policy_sample)(ag->policy, q_n, A, ag->eps_thr, (R)ag->tau, x);
            }
            for (d = 0; d < D; d++) s[d] = ns[d];
            for (j = 0; j < A; j++) q_s[j] = q_n[j];
            tmp = phi_s; phi_s = phi_n; phi_n = tmp;
            a = na;
        }
        run->action[i] = a; run->ep_step[i] = ep;
        for (j = 0; j < A; j++) qc[j] = q_s[j];
    }
    run->t += (uint64_t)n_steps;
    run->q_valid = 1;
    free(phi_s); free(phi_n);
    if (st) *st = acc;
    return 0;
}

/* The driver loop in the evaluation order of the device's WAVE family (rsrl_amd/csrc/kernels_wave.hpp: one wavefront per
 * learner, Fourier order 7 on a 4-D state space, F = 8^4 = 4096), for bitwise comparison with the HIP path (_f32d):
 *   - internal feature index k = c0*512 + c1*64 + c2*8 + c3 over ALL coefficient vectors; k = 0 (cos 0 = 1, computed as the
 *     product of the tables' ones) is the constant that with_bias() stacks last: reference feature f <-> k = (f + 1) mod F;
 *   - lane l = c1*8 + c2 owns the 64 features (j = c0, v = c3); a dot product is the lane's 4 interleaved chains over
 *     (j, v) in order, combined as (a0 + a1) + (a2 + a3), then the wave total by the DPP ladder row_shr 1, 2, 4, 8,
 *     row_bcast 15 (rows 1, 3), row_bcast 31 (rows 2, 3), read from lane 63;
 *   - Q(s',a) with the updated column is re-evaluated (no rank-1 shortcut in this family); Q(s,.) is carried;
 *   - bf16 weight storage (w_bf16 != 0): every UPDATED weight is rounded to bf16 by stochastic rounding with the 16-bit
 *     window (e >> 2) of word (e & 3) of the lane's Philox block (block id 16 + lane), e = j*8 + v.
 *     Round 6: also for SARSALambda / QLambda, GreedyGQ, TD, TDLambda and QSigma (W bf16, the trace / fa_td's weights / the backups f32): every entry of W that is
 *     stored in a step is rounded ONCE, after all of the step's updates of it, with block id 16 + 64 * column + lane.
 *   - the eligibility-trace agents (round 3; rsrl_amd/csrc/kernels_wave_lambda.hpp): orc_handle_lambda's
 *     operations with the wave-order dot products -- Q(s,.) carried, Q(s',.) with the pre-update weights, z = rule(rate*z + g)
 *     and w += alpha*residual*z on every entry, then Q(s',.) of ALL columns with the updated weights.
 * One-step control agents and SARSALambda / QLambda, per-env weights; returns -1 otherwise. */
static R FN(wave_total)(const R* lane_part) {
    R v[64], n[64]; int l, sh, row;
    for (l = 0; l < 64; l++) v[l] = lane_part[l];
    for (sh = 1; sh <= 8; sh <<= 1) {
        for (l = 0; l < 64; l++) n[l] = v[l] + (((l & 15) >= sh) ? v[l - sh] : (R)0.0);
        for (l = 0; l < 64; l++) v[l] = n[l];
    }
    for (l = 0; l < 64; l++) { row = l >> 4; n[l] = v[l] + ((row == 1 || row == 3) ? v[(row - 1) * 16 + 15] : (R)0.0); }
    for (l = 0; l < 64; l++) v[l] = n[l];
    for (l = 0; l < 64; l++) { row = l >> 4; n[l] = v[l] + ((row >= 2) ? v[31] : (R)0.0); }
    return n[63];
}
static void FN(wave_project)(const orc_basis* b, const R* s, R* phi /* [64 lanes][8 j][8 v] */) {
    R ct[4][8], st[4][8]; int d, n, l, j, v;
    for (d = 0; d < 4; d++) {
        const R lo = FN(basis_lo)(b)[d], hi = FN(basis_hi)(b)[d];
        const R sc = (s[d] - lo) * ((R)1.0 / (hi - lo));
        ct[d][0] = (R)1.0; st[d][0] = (R)0.0;
#ifdef ORC_DEVTRIG
        FN(sincospi01_dev)(sc, &st[d][1], &ct[d][1]);
#else
        ct[d][1] = (R)cos(M_PI * (double)sc); st[d][1] = (R)sin(M_PI * (double)sc);
#endif
        for (n = 2; n < 8; n++) {
            ct[d][n] = FN(fma_)(-st[d][n - 1], st[d][1], ct[d][n - 1] * ct[d][1]);
            st[d][n] = FN(fma_)(ct[d][n - 1], st[d][1], st[d][n - 1] * ct[d][1]);
        }
    }
    for (l = 0; l < 64; l++) {
        const int c1 = l >> 3, c2 = l & 7;
        for (j = 0; j < 8; j++) {
            R re = ct[0][j], im = st[0][j], nre, nim;
            nre = FN(fma_)(-im, st[1][c1], re * ct[1][c1]); nim = FN(fma_)(re, st[1][c1], im * ct[1][c1]); re = nre; im = nim;
            nre = FN(fma_)(-im, st[2][c2], re * ct[2][c2]); nim = FN(fma_)(re, st[2][c2], im * ct[2][c2]); re = nre; im = nim;
            for (v = 0; v < 8; v++) phi[(l * 8 + j) * 8 + v] = FN(fma_)(-im, st[3][v], re * ct[3][v]);
        }
    }
}
/* reference row of the internal index k = j*512 + l*8 + v */
static inline size_t FN(wave_row)(int l, int j, int v) { const int k = j * 512 + l * 8 + v; return (size_t)((k + 4095) & 4095); }
static R FN(wave_dot)(const R* phi, const R* W, int A, int a) {
    R part[64]; int l, j, v;
    for (l = 0; l < 64; l++) {
        R acc[4] = { 0, 0, 0, 0 };
        for (j = 0; j < 8; j++)
            for (v = 0; v < 8; v++) acc[v & 3] = FN(fma_)(phi[(l * 8 + j) * 8 + v], W[FN(wave_row)(l, j, v) * A + a], acc[v & 3]);
        part[l] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    return FN(wave_total)(part);
}
/* stochastic rounding to bf16 of element el = j*8 + v of a lane (kernels_wave.hpp sr_bits / round_bf16_sr): the 16-bit window (el >> 2) of word
 * (el & 3) of the lane's Philox block is added below the kept mantissa, then truncation */
static inline R FN(sr_bf16)(R x, const uint32_t* rnd, int el) {
    float xw = (float)x; uint32_t bits;
    memcpy(&bits, &xw, 4);
    bits += (rnd[el & 3] >> (el >> 2)) & 0xffffu;
    bits &= 0xffff0000u;
    memcpy(&xw, &bits, 4);
    return (R)xw;
}
int FN(orc_run_train_wave)(void* h, int64_t n_steps, orc_stats* st, int w_bf16) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag; const orc_basis* b = &ag->basis;
    int D = b->dim, A = ag->n_actions, F = orc_basis_nfeat(b), d, j, l, v, AW = ag->n_actions;     /* AW: columns of the weight matrix */
    int64_t N = run->n_envs, i, k;
    R *phi_s, *phi_n, *tmp, *w_before = NULL;
    orc_stats acc; memset(&acc, 0, sizeof(acc));
    if (FN(eps_sched)(ag) && !(ag->algo == ORC_QLEARNING || ag->algo == ORC_SARSA || ag->algo == ORC_EXPECTED_SARSA || ag->algo == ORC_PAL || ORC_IS_LAMBDA(ag->algo)))
        return -1;                   /* the schedule on the wave family: one-step agents and the lambda agents (k_train_wave / _pk <ESCHED>, k_wave_lambda) */
    if (b->kind != ORC_FOURIER || b->order != 7 || D != 4 || F != 4096 || ag->shared_w || sizeof(R) != 4 ||
        !(ag->algo == ORC_QLEARNING || ag->algo == ORC_SARSA || ag->algo == ORC_EXPECTED_SARSA || ag->algo == ORC_PAL ||
          (ORC_IS_LAMBDA(ag->algo) && run->Z) || (ag->algo == ORC_GREEDY_GQ && run->Z) ||
          ag->algo == ORC_TD || (ag->algo == ORC_TD_LAMBDA && run->Z) || (ag->algo == ORC_Q_SIGMA && run->qs))) return -1;
    if (ORC_IS_PRED(ag->algo)) AW = 1;
    phi_s = (R*)malloc(sizeof(R) * 4096); phi_n = (R*)malloc(sizeof(R) * 4096);
    for (i = 0; i < N; i++) {
        R* s = run->state + (size_t)i * D; R* W = FN(run_W)(run, i);
        R q_s[ORC_MAX_ACTIONS], q_n[ORC_MAX_ACTIONS], ns[8], ns_pre[8];
        int a = run->action[i]; uint32_t ep = run->ep_step[i];
        orc_agent agl = FN(agent_of)(run, i); const orc_agent* ag = &agl;          /* (shadows: learner i's own epsilon, when scheduled) */
        FN(wave_project)(b, s, phi_s);
        for (j = 0; j < A; j++) q_s[j] = q_n[j] = (R)0.0;
        for (j = 0; j < AW; j++) q_s[j] = FN(wave_dot)(phi_s, W, AW, j);
        for (k = 0; k < n_steps; k++) {
            const uint64_t t = run->t + (uint64_t)k;
            R r, delta, e, scale; int term, trunc, na; uint32_t x[4], xin[4] = { 0, 0, 0, 0 };
            for (d = 0; d < D; d++) ns[d] = s[d];
            term = FN(orc_domain_step)(ag->domain, ns, a, &r);
            for (d = 0; d < D; d++) ns_pre[d] = ns[d];
            ep += 1;
            trunc = !term && ag->max_episode_steps > 0 && ep >= ag->max_episode_steps;
            if (term) FN(orc_domain_reset)(ag->domain, ns);
            FN(wave_project)(b, ns, phi_n);
            for (j = 0; j < AW; j++) q_n[j] = FN(wave_dot)(phi_n, W, AW, j);
            if (ag->algo == ORC_Q_SIGMA) {
                /* QSigma on the wave family (kernels_wave_aux.hpp k_wave_qsigma): q_sigma.rs:138-201 through the wave-order switch; the n-step
                 * backup, the anchor's update and the residual are orc_handle_qsigma's */
                orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, ORC_BLK_INNER, xin);
                if (w_bf16) { if (!w_before) w_before = (R*)malloc(sizeof(R) * (size_t)F * A); memcpy(w_before, W, sizeof(R) * (size_t)F * A); }
                FN(g_wave_order) = 1;
                delta = FN(orc_handle_qsigma)(ag, W, &run->qs[i], s, a, r, ns_pre, term, xin);
                FN(g_wave_order) = 0;
                if (w_bf16) {
                    /* bf16 storage (round 6): the anchor's column -- the only one a step moves -- is rounded stochastically entry by entry before it is stored
                     * (block id 16 + 64 * column + lane).  An entry the update left where it was is bf16-representable already and the rounding keeps it: rounding
                     * the entries that moved is rounding the column. */
                    int col = -1; size_t at;
                    for (at = 0; at < (size_t)F * A && col < 0; at++) if (W[at] != w_before[at]) col = (int)(at % (size_t)A);
                    if (col >= 0)
                        for (l = 0; l < 64; l++) {
                            uint32_t rnd[4];
                            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, 16u + 64u * (uint32_t)col + (uint32_t)l, rnd);
                            for (j = 0; j < 8; j++)
                                for (v = 0; v < 8; v++) { R* wp = &W[FN(wave_row)(l, j, v) * A + col]; *wp = FN(sr_bf16)(*wp, rnd, j * 8 + v); }
                        }
                }
                for (j = 0; j < A; j++) q_n[j] = FN(wave_dot)(phi_n, W, A, j);          /* (phi_n is phi of the restart state after a terminal step) */
                goto sampled_target;
            }
            if (ag->algo == ORC_GREEDY_GQ) {
                /* GreedyGQ on the wave family (rsrl_amd/csrc/kernels_wave_aux.hpp): greedy_gq.rs:73-141 with every dot product in the wave order */
                R* V = run->Z + (size_t)i * F * A; R m, td_est, sc1, sc2, sc3; int na_star;
                const R qsa = q_s[a];
                td_est = FN(wave_dot)(phi_s, V, A, a);
                na_star = FN(orc_find_max)(q_n, A, &m);
                delta = term ? (r - qsa) : (r + (R)ag->gamma * m - qsa);
                sc1 = (R)ag->lr * delta; sc2 = (R)ag->lr * (-(R)ag->gamma * td_est); sc3 = (R)ag->lr_td * (delta - td_est);
                for (l = 0; l < 64; l++) {
                    uint32_t rnd1[4] = { 0, 0, 0, 0 }, rnd2[4] = { 0, 0, 0, 0 };
                    if (w_bf16) {
                        orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, 16u + 64u * (uint32_t)a + (uint32_t)l, rnd1);
                        orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, 16u + 64u * (uint32_t)na_star + (uint32_t)l, rnd2);
                    }
                    for (j = 0; j < 8; j++)
                        for (v = 0; v < 8; v++) {
                            const size_t row = FN(wave_row)(l, j, v); const int e_ = (l * 8 + j) * 8 + v;
                            W[row * A + a] = FN(fma_)(sc1, phi_s[e_], W[row * A + a]);
                            if (!term) W[row * A + na_star] = FN(fma_)(sc2, phi_n[e_], W[row * A + na_star]);
                            V[row * A + a] = FN(fma_)(sc3, phi_s[e_], V[row * A + a]);
                            if (w_bf16) {                                       /* one rounding per stored entry, after both updates when the columns coincide */
                                W[row * A + a] = FN(sr_bf16)(W[row * A + a], rnd1, j * 8 + v);
                                if (!term && na_star != a) W[row * A + na_star] = FN(sr_bf16)(W[row * A + na_star], rnd2, j * 8 + v);
                            }
                        }
                }
                for (j = 0; j < A; j++) q_n[j] = FN(wave_dot)(phi_n, W, A, j);
                goto sampled_target;
            }
            if (ORC_IS_PRED(ag->algo)) {
                /* TD / TDLambda on the wave family: td.rs:31-59, td_lambda.rs:41-78; one weight column, V(s) in the wave order */
                R* Z = run->Z ? run->Z + (size_t)i * F : NULL; R rate = (R)orc_trace_rate(ag);
                delta = term ? (r - q_s[0]) : (r + (R)ag->gamma * q_n[0] - q_s[0]);
                for (l = 0; l < 64; l++) {
                    uint32_t rnd[4] = { 0, 0, 0, 0 };
                    if (w_bf16) orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, 16u + (uint32_t)l, rnd);
                    for (j = 0; j < 8; j++)
                        for (v = 0; v < 8; v++) {
                            const size_t row = FN(wave_row)(l, j, v); const int e_ = (l * 8 + j) * 8 + v;
                            if (ag->algo == ORC_TD_LAMBDA) {
                                R z = FN(fma_)(rate, Z[row], phi_s[e_]);
                                if (ag->trace == ORC_TRACE_SATURATE) { z = (z < (R)1.0) ? z : (R)1.0; z = (z > (R)-1.0) ? z : (R)-1.0; }
                                W[row] = FN(fma_)(delta, z, W[row]);
                                Z[row] = term ? (R)0.0 : z;
                            } else {
                                W[row] = FN(fma_)((R)ag->lr * delta, phi_s[e_], W[row]);
                            }
                            if (w_bf16) W[row] = FN(sr_bf16)(W[row], rnd, j * 8 + v);
                        }
                }
                q_n[0] = FN(wave_dot)(phi_n, W, 1, 0);
                goto sampled_target;
            }
            if (ORC_IS_LAMBDA(ag->algo)) {
                R* Z = run->Z + (size_t)i * F * A; R rate, m; int c, na_in;
                const R qsa = q_s[a];
                const int cut = ag->algo == ORC_Q_LAMBDA && a != FN(orc_argmax_first)(q_s, A);
                rate = (R)orc_trace_rate(ag);
                if (cut) memset(Z, 0, sizeof(R) * (size_t)F * A);
                if (term) delta = r - qsa;
                else if (ag->algo == ORC_SARSA_LAMBDA) {
                    orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, ORC_BLK_INNER, xin);
                    na_in = FN(orc_policy_sample)(ag->apolicy, q_n, A, ag->aeps_thr, (R)ag->atau, xin);
                    delta = r + (R)ag->gamma * q_n[na_in] - qsa;
                } else { FN(orc_find_max)(q_n, A, &m); delta = r + (R)ag->gamma * m - qsa; }
                scale = (R)ag->alpha * delta;
                for (l = 0; l < 64; l++) {
                  uint32_t rnd[ORC_MAX_ACTIONS][4];
                  for (c = 0; c < A; c++) { rnd[c][0] = rnd[c][1] = rnd[c][2] = rnd[c][3] = 0; if (w_bf16) orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, 16u + 64u * (uint32_t)c + (uint32_t)l, rnd[c]); }
                  for (j = 0; j < 8; j++)
                        for (v = 0; v < 8; v++)
                            for (c = 0; c < A; c++) {
                                const size_t at = FN(wave_row)(l, j, v) * A + c;
                                const R g = (c == a) ? phi_s[(l * 8 + j) * 8 + v] : (R)0.0;
                                R z = FN(fma_)(rate, Z[at], g);
                                if (ag->trace == ORC_TRACE_SATURATE) { z = (z < (R)1.0) ? z : (R)1.0; z = (z > (R)-1.0) ? z : (R)-1.0; }
                                W[at] = FN(fma_)(scale, z, W[at]);
                                if (w_bf16) W[at] = FN(sr_bf16)(W[at], rnd[c], j * 8 + v);
                                Z[at] = term ? (R)0.0 : z;
                            }
                }
                for (j = 0; j < A; j++) q_n[j] = FN(wave_dot)(phi_n, W, A, j);          /* every column moved */
                goto sampled_target;
            }
            if (ag->algo == ORC_SARSA) orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, ORC_BLK_INNER, xin);
            delta = FN(td_from_q)(ag, q_s, a, q_n, r, term, xin, &e);
            scale = (R)ag->lr * e;
            for (l = 0; l < 64; l++) {
                uint32_t rnd[4] = { 0, 0, 0, 0 };
                if (w_bf16) orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, 16u + (uint32_t)l, rnd);
                for (j = 0; j < 8; j++)
                    for (v = 0; v < 8; v++) {
                        R* wp = &W[FN(wave_row)(l, j, v) * A + a];
                        float xw = (float)FN(fma_)(scale, phi_s[(l * 8 + j) * 8 + v], *wp);
                        if (w_bf16) {
                            const int el = j * 8 + v; uint32_t bits;
                            memcpy(&bits, &xw, 4);
                            bits += (rnd[el & 3] >> (el >> 2)) & 0xffffu;
                            bits &= 0xffff0000u;
                            memcpy(&xw, &bits, 4);
                        }
                        *wp = (R)xw;
                    }
            }
            q_n[a] = FN(wave_dot)(phi_n, W, A, a);                          /* Q(s',a) with the UPDATED column */
sampled_target:
            if (term || trunc) { FN(eps_episode_end)(run, i); agl = FN(agent_of)(run, i); }      /* examples/sarsa_lambda.rs:68: the episode's last handle is done */
            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, term ? ORC_BLK_RESET : ORC_BLK_STEP, x);
            na = FN(orc_policy_sample)(ag->policy, q_n, A, ag->eps_thr, (R)ag->tau, x);
            acc.sum_abs_td_error += fabs((double)delta); acc.sum_reward += (double)r; acc.env_steps += 1;
            if (term) { acc.episodes += 1; acc.sum_episode_steps += ep; ep = 0; }
            if (trunc) {
                acc.episodes += 1; acc.episodes_truncated += 1; acc.sum_episode_steps += ep; ep = 0;
                FN(orc_domain_reset)(ag->domain, ns);
                FN(wave_project)(b, ns, phi_n);
                for (j = 0; j < AW; j++) q_n[j] = FN(wave_dot)(phi_n, W, AW, j);
                orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, ORC_BLK_RESET, x);
                na = FN(orc_policy_sample)(ag->policy, q_n, A, ag->eps_thr, (R)ag->tau, x);
            }
            for (d = 0; d < D; d++) s[d] = ns[d];
            for (j = 0; j < A; j++) q_s[j] = q_n[j];
            tmp = phi_s; phi_s = phi_n; phi_n = tmp;
            a = na;
        }
        run->action[i] = a; run->ep_step[i] = ep;
    }
    run->t += (uint64_t)n_steps;
    free(phi_s); free(phi_n); free(w_before);
    if (st) *st = acc;
    return 0;
}
/* SARSALambda / QLambda over ONE SHARED tile-coded table with a SPARSE trace per learner -- the reference's Trace<B, R> is generic over its
 * buffer (rsrl/src/traces.rs:5-12) and ships a sparse one (rsrl/src/params/sparse.rs:13-97) -- restating rsrl_amd/csrc/kernels_sparse_lambda.hpp
 * operation for operation (float instantiations: bit-identical to the HIP path; R = double: the same rule in the reference's precision with a
 * plain sum over the learners):
 *   batch-step = the synchronous mini-batch rule of the shared-weight modes: every learner's residual (sarsa_lambda.rs:53-98,
 *   q_lambda.rs:56-99) and trace update against W_t; W_{t+1} = W_t + sum_i (alpha * residual_i) * z_i; a terminal transition empties z_i; then
 *   every learner samples from W_{t+1} and finished episodes restart.
 *   the list: ORC_SPARSE_CAP = 512 (key = tile index * A + action, value) entries per learner, held as T sub-lists of 512 / T entries, one per
 *   tiling (a key belongs to one tiling; round 6: the device updates a learner's trace tiling by tiling inside its scatter kernel).  Per step and
 *   tiling: (1) every entry v <- rule(fma(rate, v, hit ? 1 : 0)) with hit = its key is the tiling's new key; (2) the new key, if not in the
 *   sub-list: appended, or -- sub-list full -- written over its entry with the smallest |v| (ties: lowest slot), value rule(fma(rate, 0, 1));
 *   (3) terms (alpha*residual) * v into the sum (float: 64-bit fixed point, lsb = 2^(floor(log2 alpha) - 28), clamped to +-2^42 -- exact and
 *   order-independent).
 * Returns -1 for any other configuration. */
#define ORC_SPARSE_CAP 512
int FN(orc_run_train_sparse_lambda)(void* h, int64_t n_steps, orc_stats* st) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag; const orc_basis* b = &ag->basis;
    int D = b->dim, A = ag->n_actions, F = orc_basis_nfeat(b), T = b->n_tilings, d, tt, c, e;
    int64_t N = run->n_envs, i, k;
    const int fixed = sizeof(R) == 4, sarsa = ag->algo == ORC_SARSA_LAMBDA;
    const R alpha = (R)ag->alpha;
    double rate_d = ag->gamma * ag->lambda; R rate, fresh;
    R* ns_all; uint8_t* flag_all; R* dW; int64_t* qacc; float lsb_f = 1.0f, inv_lsb_f = 1.0f;
    orc_agent tgt = *ag;
    orc_stats acc; memset(&acc, 0, sizeof(acc));
    if (!ORC_IS_LAMBDA(ag->algo) || b->kind != ORC_TILE || !ag->shared_w || FN(eps_sched)(ag)) return -1;
    if (ag->trace == ORC_TRACE_DUTCH) rate_d *= (1.0 - ag->alpha);
    rate = (R)rate_d;                                                         /* (the device's make_lambda: the product in double, rounded once) */
    tgt.algo = sarsa ? ORC_SARSA : ORC_QLEARNING;                             /* the TD target formula */
    if (!run->sp_keys) {
        run->sp_keys = (uint32_t*)calloc((size_t)N * ORC_SPARSE_CAP, sizeof(uint32_t));
        run->sp_vals = (R*)calloc((size_t)N * ORC_SPARSE_CAP, sizeof(R));
        run->sp_len = (int*)calloc((size_t)N * (size_t)T, sizeof(int));
    }
    fresh = FN(fma_)(rate, (R)0.0, (R)1.0);
    if (ag->trace == ORC_TRACE_SATURATE) { fresh = (fresh < (R)1.0) ? fresh : (R)1.0; fresh = (fresh > (R)-1.0) ? fresh : (R)-1.0; }
    if (fixed) {
        const float af = (float)ag->alpha; uint32_t u, eb, ex, v;
        memcpy(&u, &af, 4); eb = (u >> 23) & 0xffu; ex = (eb < 30u ? 30u : eb) - 28u;
        v = ex << 23; memcpy(&lsb_f, &v, 4); v = (254u - ex) << 23; memcpy(&inv_lsb_f, &v, 4);
    }
    ns_all = (R*)malloc(sizeof(R) * (size_t)N * D); flag_all = (uint8_t*)malloc((size_t)N);
    dW = (R*)malloc(sizeof(R) * (size_t)F * A); qacc = (int64_t*)malloc(sizeof(int64_t) * (size_t)F * A);
    for (k = 0; k < n_steps; k++, run->t++) {
        memset(dW, 0, sizeof(R) * (size_t)F * A); memset(qacc, 0, sizeof(int64_t) * (size_t)F * A);
        for (i = 0; i < N; i++) {
            R* s = run->state + (size_t)i * D; R* ns = ns_all + (size_t)i * D;
            uint32_t* Kl = run->sp_keys + (size_t)i * ORC_SPARSE_CAP; R* Vl = run->sp_vals + (size_t)i * ORC_SPARSE_CAP; int* lens = run->sp_len + (size_t)i * T;
            const int cap = ORC_SPARSE_CAP / T;
            R r, delta, e_, scale, q_s[ORC_MAX_ACTIONS], q_n[ORC_MAX_ACTIONS]; int a = run->action[i], term, trunc, is[ORC_MAX_TILINGS], in[ORC_MAX_TILINGS];
            uint32_t xin[4] = { 0, 0, 0, 0 }; float sf[8];
            memcpy(ns, s, sizeof(R) * D);
            term = FN(orc_domain_step)(ag->domain, ns, a, &r);
            run->ep_step[i] += 1;
            trunc = !term && ag->max_episode_steps > 0 && run->ep_step[i] >= ag->max_episode_steps;
            flag_all[i] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0));
            if (run->round32) for (d = 0; d < D; d++) ns[d] = (R)(float)ns[d];          /* teacher forcing: the transition as a fp32 caller receives it */
            if (run->tape_from) {
                memcpy(run->tape_from + (size_t)i * D, s, sizeof(R) * D); memcpy(run->tape_to + (size_t)i * D, ns, sizeof(R) * D);
                run->tape_act[i] = a; run->tape_rew[i] = r; run->tape_term[i] = (uint8_t)term;
            }
            for (d = 0; d < D; d++) sf[d] = (float)s[d];
            orc_tile_indices(b, sf, is);
            for (d = 0; d < D; d++) sf[d] = (float)ns[d];
            orc_tile_indices(b, sf, in);
            for (c = 0; c < A; c++) { q_s[c] = (R)0.0; q_n[c] = (R)0.0; }
            for (tt = 0; tt < T; tt++) for (c = 0; c < A; c++) { q_s[c] = q_s[c] + run->W[(size_t)is[tt] * A + c]; q_n[c] = q_n[c] + run->W[(size_t)in[tt] * A + c]; }
            { const int cut = !sarsa && a != FN(orc_argmax_first)(q_s, A);            /* Watkins's cut (q_lambda.rs:62-66) */
              if (sarsa) orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t, ORC_BLK_INNER, xin);
              delta = FN(td_from_q)(&tgt, q_s, a, q_n, r, term, xin, &e_);
              scale = alpha * delta;
              for (tt = 0; tt < T; tt++) {                                              /* the trace, tiling by tiling */
                uint32_t* K = Kl + (size_t)tt * cap; R* V = Vl + (size_t)tt * cap; int len = cut ? 0 : lens[tt], found_ = 0;
                const uint32_t nk_ = (uint32_t)is[tt] * (uint32_t)A + (uint32_t)a;
                for (e = 0; e < len; e++) {
                    const int hit = K[e] == nk_; R v;
                    if (hit) found_ = 1;
                    v = FN(fma_)(rate, V[e], hit ? (R)1.0 : (R)0.0);
                    if (ag->trace == ORC_TRACE_SATURATE) { v = (v < (R)1.0) ? v : (R)1.0; v = (v > (R)-1.0) ? v : (R)-1.0; }
                    V[e] = v;
                }
                if (!found_) {
                    int slot;
                    if (len < cap) slot = len++;
                    else {
                        R best = V[0] < 0 ? -V[0] : V[0]; slot = 0;
                        for (e = 1; e < cap; e++) { const R m = V[e] < 0 ? -V[e] : V[e]; if (m < best) { best = m; slot = e; } }
                    }
                    K[slot] = nk_; V[slot] = fresh;
                }
                for (e = 0; e < len; e++) {
                    const R term_ = scale * V[e];
                    if (fixed) {
                        float sc = (float)term_ * inv_lsb_f;
                        sc = sc < -4.398046511104e12f ? -4.398046511104e12f : (sc > 4.398046511104e12f ? 4.398046511104e12f : sc);
                        qacc[K[e]] += (int64_t)rintf(sc);
                    } else dW[K[e]] += term_;
                }
                lens[tt] = term ? 0 : len;                                              /* trace.reset() */
              }
            }
            if (run->tape_td) run->tape_td[i] = delta;
            acc.sum_abs_td_error += fabs((double)delta); acc.sum_reward += (double)r; acc.env_steps += 1;
        }
        { int j; for (j = 0; j < F * A; j++) run->W[j] += fixed ? (R)((float)qacc[j] * lsb_f) : dW[j]; }
        for (i = 0; i < N; i++) {
            R* s = run->state + (size_t)i * D; R* ns = ns_all + (size_t)i * D;
            R q[ORC_MAX_ACTIONS]; uint32_t x[4]; int idx[ORC_MAX_TILINGS]; float sf[8];
            if (flag_all[i]) {
                acc.episodes += 1; if (flag_all[i] & 2) acc.episodes_truncated += 1;
                acc.sum_episode_steps += run->ep_step[i]; run->ep_step[i] = 0;
                FN(orc_domain_reset)(ag->domain, ns);
            }
            for (d = 0; d < D; d++) sf[d] = (float)ns[d];
            orc_tile_indices(b, sf, idx);
            for (c = 0; c < A; c++) q[c] = (R)0.0;
            for (tt = 0; tt < T; tt++) for (c = 0; c < A; c++) q[c] = q[c] + run->W[(size_t)idx[tt] * A + c];
            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t, ORC_BLK_STEP, x);
            run->action[i] = FN(orc_policy_sample)(ag->policy, q, A, ag->eps_thr, (R)ag->tau, x);
            memcpy(s, ns, sizeof(R) * D);
        }
    }
    free(ns_all); free(flag_all); free(dW); free(qacc);
    if (st) *st = acc;
    return 0;
}
/* one batch-step of the sparse-trace loop as a TEACHER (orc_run_teacher's contract: successor states rounded to fp32, the transitions the agents saw on the tape):
 * replayed through rsrl_hip_handle -- transition i is learner i's -- device and oracle learn from identical inputs.  Returns orc_run_train_sparse_lambda's status. */
int FN(orc_run_teacher_sparse_lambda)(void* h, orc_stats* st, R* from, int32_t* act, R* rew, R* to, uint8_t* term, R* td) {
    FN(orc_run)* run = (FN(orc_run)*)h; int rc;
    run->round32 = 1;
    run->tape_from = from; run->tape_act = act; run->tape_rew = rew; run->tape_to = to; run->tape_term = term; run->tape_td = td;
    rc = FN(orc_run_train_sparse_lambda)(h, 1, st);
    run->tape_from = NULL; run->tape_act = NULL; run->tape_rew = NULL; run->tape_to = NULL; run->tape_term = NULL; run->tape_td = NULL;
    run->round32 = 0;
    return rc;
}
/* learner i's sparse trace as the dense (F, A) matrix it stands for; `out` holds F*A zeros on entry */
void FN(orc_run_sparse_trace)(void* h, int64_t i, R* out) {
    FN(orc_run)* run = (FN(orc_run)*)h; int e;
    if (!run->sp_keys) return;
    { const int T = run->ag.basis.n_tilings, cap = ORC_SPARSE_CAP / T; int tt;
      for (tt = 0; tt < T; tt++)
        for (e = 0; e < run->sp_len[(size_t)i * T + tt]; e++)
            out[run->sp_keys[(size_t)i * ORC_SPARSE_CAP + (size_t)tt * cap + e]] = run->sp_vals[(size_t)i * ORC_SPARSE_CAP + (size_t)tt * cap + e]; }
}

/* the wave family's initial policy.sample (k_wave_reset): Q(s0,.) in the wave order */
void FN(orc_run_reset_wave)(void* h) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag;
    int D = ag->basis.dim, A = ag->n_actions, j; int64_t i;
    R* phi = (R*)malloc(sizeof(R) * 4096);
    run->q_valid = 0;
    if (run->qs) for (i = 0; i < run->n_envs; i++) run->qs[i].len = 0;         /* (as orc_run_reset) */
    for (i = 0; i < run->n_envs; i++) {
        R q[ORC_MAX_ACTIONS]; uint32_t x[4];
        R* s = run->state + (size_t)i * D;
        FN(orc_domain_reset)(ag->domain, s);
        FN(wave_project)(&ag->basis, s, phi);
        for (j = 0; j < A; j++) q[j] = FN(wave_dot)(phi, FN(run_W)(run, i), A, j);
        orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), run->t, ORC_BLK_INIT, x);
        run->action[i] = FN(orc_policy_sample)(ag->policy, q, A, ag->eps_thr, (R)ag->tau, x);
        run->ep_step[i] = 0;
    }
    free(phi);
}

/* Shared weights, dense basis, in the DEVICE's evaluation order (rsrl_amd/csrc/models.hpp: k_shared_step with 512-learner
 * blocks), for bitwise comparison with the HIP path (_f32d).  The rule is SURVEY A.7's (every learner's error against the
 * same W_t, the summed delta applied once, every learner then samples with W_{t+1}); what is mirrored is the ORDER of the sum:
 *   block (512 learners): for each (action b, feature f), per wave of 64 consecutive learners FOUR chains
 *       acc_c = fma([a_i == b], lr*e_i*phi_i[f], acc_c) over the learners i = c mod 4 in ascending order (the device runs them
 *       as rank-1 MFMA updates on four accumulators: same products, same roundings, same order), the wave's part =
 *       (acc_0 + acc_1) + (acc_2 + acc_3), the eight parts added in wave order                    -> one row per block
 *   blocks: every block's sums are rounded to 64-bit fixed point (lsb = 2^(floor(log2 lr) - 28), clamped to +-2^42), the
 *       integers are added over the blocks (exact, any order) and converted back with one rounding -> W_t = W_{t-1} + total
 * and the launch structure: phase C of batch-step t-1 (sample with W_t; finished episodes restart) and phase A of batch-step
 * t run together, a train call opens with phase A alone and closes with the last fold + phase C alone.
 * One-step control agents on a Fourier basis, shared W; returns -1 otherwise. */
int FN(orc_run_train_shared_dev)(void* h, int64_t n_steps, orc_stats* st) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag; const orc_basis* b = &ag->basis;
    enum { BLOCK = 512, PER = 64, H = BLOCK / PER };
    int D = b->dim, A = ag->n_actions, F = orc_basis_nfeat(b), AF, d, f, j, a, hh, n_rows;
    int64_t N = run->n_envs, i, k, blk;
    R *phi, *W, *terms, *phis; uint8_t* flags; int* acts; int64_t* qsum;
    float lsb_f, inv_lsb_f;
    orc_stats acc; memset(&acc, 0, sizeof(acc));
    if (FN(eps_sched)(ag)) return -1;
    if (b->kind != ORC_FOURIER || !ag->shared_w || n_steps < 1 ||
        !(ag->algo == ORC_QLEARNING || ag->algo == ORC_SARSA || ag->algo == ORC_EXPECTED_SARSA || ag->algo == ORC_PAL)) return -1;
    AF = A * F; n_rows = (int)((N + BLOCK - 1) / BLOCK); W = run->W;
    phi = (R*)malloc(sizeof(R) * (size_t)F); qsum = (int64_t*)calloc((size_t)AF, sizeof(int64_t));
    { const float lrf = (float)ag->lr; uint32_t u, eb, ex, v;
      memcpy(&u, &lrf, 4); eb = (u >> 23) & 0xffu; ex = (eb < 30u ? 30u : eb) - 28u;
      v = ex << 23; memcpy(&lsb_f, &v, 4); v = (254u - ex) << 23; memcpy(&inv_lsb_f, &v, 4); }
    terms = (R*)malloc(sizeof(R) * BLOCK); phis = (R*)malloc(sizeof(R) * (size_t)BLOCK * F); acts = (int*)malloc(sizeof(int) * BLOCK);
    flags = (uint8_t*)calloc((size_t)N, 1);
    for (k = 0; k <= n_steps; k++) {
        const uint64_t t = run->t + (uint64_t)k;
        const int do_c = k > 0, do_a = k < n_steps;
        if (do_c)                                                     /* fold the previous batch-step's delta: W_t = W_{t-1} + total */
            for (a = 0; a < A; a++) for (f = 0; f < F; f++) { j = a * F + f; W[(size_t)f * A + a] = W[(size_t)f * A + a] + (R)((float)qsum[j] * lsb_f); }
        memset(qsum, 0, sizeof(int64_t) * (size_t)AF);
        for (blk = 0; blk < n_rows; blk++) {
            for (i = blk * BLOCK; i < (blk + 1) * (int64_t)BLOCK; i++) {
                const int li = (int)(i - blk * BLOCK);
                R s[8], ns[8], q_s[ORC_MAX_ACTIONS], q_n[ORC_MAX_ACTIONS], r, e, delta; uint32_t x[4], xin[4] = { 0, 0, 0, 0 }; uint32_t ep; int done, term, trunc;
                if (i >= N) { terms[li] = (R)0.0; acts[li] = 0; for (f = 0; f < F; f++) phis[(size_t)li * F + f] = (R)0.0; continue; }
                ep = run->ep_step[i];
                done = do_c && flags[i] != 0;
                if (done) { FN(orc_domain_reset)(ag->domain, s); ep = 0; }
                else for (d = 0; d < D; d++) s[d] = run->state[(size_t)i * D + d];
                FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), s, phi);
                FN(dot_columns)(phi, W, A, F, q_s);
                if (do_c) {
                    orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t - 1, done ? ORC_BLK_RESET : ORC_BLK_STEP, x);
                    a = FN(orc_policy_sample)(ag->policy, q_s, A, ag->eps_thr, (R)ag->tau, x);
                } else a = run->action[i];
                if (do_a) {
                    for (d = 0; d < D; d++) ns[d] = s[d];
                    term = FN(orc_domain_step)(ag->domain, ns, a, &r);
                    ep += 1;
                    trunc = !term && ag->max_episode_steps > 0 && ep >= ag->max_episode_steps;
                    for (f = 0; f < F; f++) phis[(size_t)li * F + f] = phi[f];            /* phi(s): the gradient direction */
                    FN(orc_fourier_project)(b->order, D, FN(basis_lo)(b), FN(basis_hi)(b), ns, phi);
                    FN(dot_columns)(phi, W, A, F, q_n);
                    if (ag->algo == ORC_SARSA) orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), t, ORC_BLK_INNER, xin);
                    delta = FN(td_from_q)(ag, q_s, a, q_n, r, term, xin, &e);
                    terms[li] = (R)ag->lr * e; acts[li] = a;
                    for (d = 0; d < D; d++) run->state[(size_t)i * D + d] = ns[d];
                    run->action[i] = a; run->ep_step[i] = ep;
                    flags[i] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0));
                    acc.sum_abs_td_error += fabs((double)delta); acc.sum_reward += (double)r; acc.env_steps += 1;
                    if (term || trunc) { acc.episodes += 1; acc.episodes_truncated += trunc ? 1 : 0; acc.sum_episode_steps += ep; }
                } else {                                              /* closing launch: phase C only */
                    if (done) { for (d = 0; d < D; d++) run->state[(size_t)i * D + d] = s[d]; run->ep_step[i] = 0; }
                    run->action[i] = a;
                }
            }
            if (do_a)
                for (a = 0; a < A; a++) for (f = 0; f < F; f++) {
                    R tot = (R)0.0;
                    for (hh = 0; hh < H; hh++) {                    /* per wave: four chains (learner k -> chain k mod 4), (c0 + c1) + (c2 + c3) */
                        R ch[4] = { (R)0.0, (R)0.0, (R)0.0, (R)0.0 }, part; int li;
                        for (li = hh * PER; li < (hh + 1) * PER; li++) {
                            const R v = terms[li] * phis[(size_t)li * F + f];
                            ch[li & 3] = FN(fma_)((acts[li] == a) ? (R)1.0 : (R)0.0, v, ch[li & 3]);
                        }
                        part = (ch[0] + ch[1]) + (ch[2] + ch[3]);
                        tot = (hh == 0) ? part : tot + part;
                    }
                    { float sc = (float)tot * inv_lsb_f;
                      sc = sc < -4.398046511104e12f ? -4.398046511104e12f : (sc > 4.398046511104e12f ? 4.398046511104e12f : sc);
                      qsum[a * F + f] += (int64_t)rintf(sc); }
                }
        }
    }
    run->t += (uint64_t)n_steps;
    free(phi); free(qsum); free(terms); free(phis); free(acts); free(flags);
    if (st) *st = acc;
    return 0;
}

/* Domain::rollout with pi = policy.mode, Some(limit)   rsrl_domains/src/lib.rs:448-479; n_states lib.rs:340
 * One fresh default env per learner i, evaluated with learner i's weights. */
/* the gap between the largest and the second largest action value: how far this action selection is from flipping under a rounding */
static R FN(argmax_margin)(const R* q, int A) {
    R best = q[0], second = (R)0.0; int a, have = 0;
    for (a = 1; a < A; a++) {
        if (q[a] > best) { second = best; best = q[a]; have = 1; }
        else if (!have || q[a] > second) { second = q[a]; have = 1; }
    }
    return have ? best - second : (R)0.0;
}
/* min_margin (optional, [N]): the smallest argmax margin over the learner's action selections (the greedy-rollout comparison
 * is meaningful where it is above fp32 resolution) */
int FN(orc_run_rollout_greedy_margin)(void* h, int64_t step_limit, uint32_t* n_states, R* total_reward, R* min_margin) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag;
    int A = ag->n_actions; int64_t i;
    if (step_limit < 1 || ag->policy == ORC_RANDOM || ORC_IS_PRED(ag->algo)) return -1;
    for (i = 0; i < run->n_envs; i++) {
        R s[8], q[ORC_MAX_ACTIONS], r, tot = 0, mm, m; int a, term; int64_t steps = 0;
        const R* W = FN(run_W)(run, i);
        FN(orc_domain_reset)(ag->domain, s);
        /* first step happens eagerly, before take(limit-1)  (lib.rs:457-459) */
        FN(orc_q_evaluate)(&ag->basis, W, A, s, q);
        mm = FN(argmax_margin)(q, A);
        a = FN(orc_policy_mode)(ag->policy, q, A, (R)ag->tau);
        term = FN(orc_domain_step)(ag->domain, s, a, &r);
        while (steps < step_limit - 1) {
            steps++; tot += r;
            if (term) break;                                            /* successors stops after Terminal */
            if (steps >= step_limit - 1) break;
            FN(orc_q_evaluate)(&ag->basis, W, A, s, q);
            m = FN(argmax_margin)(q, A); if (m < mm) mm = m;
            a = FN(orc_policy_mode)(ag->policy, q, A, (R)ag->tau);
            term = FN(orc_domain_step)(ag->domain, s, a, &r);
        }
        n_states[i] = (uint32_t)(steps + 1);
        if (total_reward) total_reward[i] = tot;
        if (min_margin) min_margin[i] = mm;
    }
    return 0;
}
int FN(orc_run_rollout_greedy)(void* h, int64_t step_limit, uint32_t* n_states, R* total_reward) {
    return FN(orc_run_rollout_greedy_margin)(h, step_limit, n_states, total_reward, (R*)0);
}

/* Domain::rollout with ANY policy as the closure: pi = |s| policy.sample(rng, s)     rsrl_domains/src/lib.rs:448-479 takes any FnMut(&S) -> A
 * (policies/mod.rs:65-78).  policy / eps / tau: the sampling policy and its parameters; the k-th action selection of learner i in
 * rollout call `call` draws the whole Philox block (counter (call << 32) | k, block ORC_BLK_ROLLOUT): word 0 = explore?, word 1 =
 * the random action, word 2 = tie-break / softmax u -- what orc_policy_sample expects. */
int FN(orc_run_rollout_policy)(void* h, int policy, double eps, double tau, uint64_t call, int64_t step_limit, uint32_t* n_states, R* total_reward,
                               int32_t* actions /* [step_limit-1][N] or NULL */) {
    FN(orc_run)* run = (FN(orc_run)*)h; const orc_agent* ag = &run->ag;
    int A = ag->n_actions; int64_t i;
    const uint32_t thr = orc_eps_threshold(eps);
    if (step_limit < 1 || ORC_IS_PRED(ag->algo) || policy < 0 || policy > ORC_RANDOM) return -1;
    for (i = 0; i < run->n_envs; i++) {
        R s[8], q[ORC_MAX_ACTIONS], r, tot = 0; int a, term; int64_t steps = 0; uint32_t x[4]; uint64_t k = 0;
        const R* W = FN(run_W)(run, i);
        FN(orc_domain_reset)(ag->domain, s);
        FN(orc_q_evaluate)(&ag->basis, W, A, s, q);
        orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), (call << 32) | (k++ & 0xffffffffu), ORC_BLK_ROLLOUT, x);
        a = FN(orc_policy_sample)(policy, q, A, thr, (R)tau, x);
        term = FN(orc_domain_step)(ag->domain, s, a, &r);
        while (steps < step_limit - 1) {
            if (actions) actions[steps * run->n_envs + i] = a;
            steps++; tot += r;
            if (term) break;
            if (steps >= step_limit - 1) break;
            FN(orc_q_evaluate)(&ag->basis, W, A, s, q);
            orc_draw(ag->seed, (uint64_t)(ag->env_offset + i), (call << 32) | (k++ & 0xffffffffu), ORC_BLK_ROLLOUT, x);
            a = FN(orc_policy_sample)(policy, q, A, thr, (R)tau, x);
            term = FN(orc_domain_step)(ag->domain, s, a, &r);
        }
        n_states[i] = (uint32_t)(steps + 1);
        if (total_reward) total_reward[i] = tot;
    }
    return 0;
}
