/* check_constdiv.c -- TEST INFRASTRUCTURE.  Exhaustive check of the sequence the HIP path uses in place of an IEEE fp32 division by a
 * compile-time constant d (rsrl_amd/csrc/device_core.hpp div_const):
 *
 *     x / d   ->   (float)((double)x * RN64(1.0 / (double)d))          v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64
 *
 * against the reference's operation, the correctly rounded fp32 quotient (rsrl_domains/src/ode.rs:36 `/ 6.0`, cart_pole.rs:60
 * `/ TOTAL_MASS`, and the tile coder's (s - lo) / (hi - lo)), for EVERY float x -- zeros of both signs, denormals, infinities and NaNs
 * included.  Why it can hold: the double product is within 2^-52 (relative) of x / d, and x / d is never closer than ~2^-49 to a
 * rounding boundary of fp32 unless it IS representable (d x midpoint has more than 24 significant bits) -- so the second rounding
 * rounds the way the first one would have.  This program is the proof for the divisors actually used: it tries every x.
 * (mode 1 checks Markstein's fp32 sequence q = x*r; e = fma(-q, d, x); q' = fma(e, r, q); copysign -- 4 fp32 instructions, but wrong
 * below |x| ~ 1e-37 where the residual underflows: not used.)
 *
 *     check_constdiv <mode: 0 = via f64, 1 = Markstein> <stride> <n_threads> <lo_abs> <hi_abs> d1 [d2 ...]
 *
 * compares all x with lo_abs <= |x| <= hi_abs, plus zeros, infinities and NaNs, and prints per divisor the number of mismatching bit
 * patterns (NaNs compare equal to NaNs).  Exit status 0 iff every divisor has none. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static int g_mode = 0;
static inline float div_const(float x, float d, float r) {
    if (g_mode == 0) return (float)((double)x * (1.0 / (double)d));
    const float q = x * r;
    const float e = fmaf(-q, d, x);
    const float q2 = fmaf(e, r, q);
    return u2f((f2u(q2) & 0x7fffffffu) | (f2u(x) & 0x80000000u));
}

typedef struct { float d; uint32_t stride; float lo, hi; int tid, nth; uint64_t bad, seen; uint32_t first_bad; } job_t;

static void* work(void* p) {
    job_t* j = (job_t*)p;
    const float d = j->d, r = 1.0f / d;
    uint64_t bad = 0, seen = 0; uint32_t first = 0;
    const uint64_t chunk = ((uint64_t)1 << 32) / (uint64_t)j->nth;
    const uint64_t b0 = chunk * (uint64_t)j->tid, b1 = j->tid == j->nth - 1 ? ((uint64_t)1 << 32) : b0 + chunk;
    for (uint64_t b = b0; b < b1; b += j->stride) {
        const float x = u2f((uint32_t)b), ax = fabsf(x);
        if (!(ax == 0.0f || (ax >= j->lo && ax <= j->hi) || isnan(x) || isinf(x))) continue;
        const float want = x / d, got = div_const(x, d, r);
        seen++;
        if (f2u(want) != f2u(got) && !(isnan(want) && isnan(got))) { if (!bad) first = (uint32_t)b; bad++; }
    }
    j->bad = bad; j->seen = seen; j->first_bad = first;
    return NULL;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s mode stride n_threads lo_abs hi_abs d1 [d2 ...]\n", argv[0]); return 2; }
    g_mode = atoi(argv[1]);
    const uint32_t stride = (uint32_t)strtoul(argv[2], NULL, 10);
    int nth = atoi(argv[3]); if (nth < 1) nth = 1; if (nth > 64) nth = 64;
    const float lo = strtof(argv[4], NULL), hi = strtof(argv[5], NULL);
    int rc = 0;
    for (int a = 6; a < argc; a++) {
        const float d = strtof(argv[a], NULL);
        pthread_t th[64]; job_t jobs[64];
        for (int t = 0; t < nth; t++) { jobs[t] = (job_t){d, stride ? stride : 1, lo, hi, t, nth, 0, 0, 0}; pthread_create(&th[t], NULL, work, &jobs[t]); }
        uint64_t bad = 0, seen = 0; uint32_t first = 0;
        for (int t = 0; t < nth; t++) { pthread_join(th[t], NULL); if (jobs[t].bad && !bad) first = jobs[t].first_bad; bad += jobs[t].bad; seen += jobs[t].seen; }
        printf("d=%.9g (bits %08x) r=%.9g: %llu inputs, %llu mismatches", (double)d, f2u(d), (double)(1.0f / d), (unsigned long long)seen, (unsigned long long)bad);
        if (bad) printf(" (first at x bits %08x = %.9g: want %.9g got %.9g)", first, (double)u2f(first), (double)(u2f(first) / d), (double)div_const(u2f(first), d, 1.0f / d));
        printf("\n");
        if (bad) rc = 1;
    }
    return rc;
}
