// pk_forward: does gfx950 interlock a packed-fp32 producer and the VALU instruction that consumes its result in the NEXT issue slot?
//
// rsrl_amd/_asmfilter.py removes the `s_nop 0` ROCm 7.2's hazard recogniser places between v_pk_{fma,mul,add}_f32 and a consumer of the result
// (its DstSelForwarding rule reads bit 3 of src0_modifiers, which is DST_OP_SEL on VOP3 but op_sel_hi[0] on VOP3P).  This program isolates the
// pair: producer and consumer sit back to back inside ONE inline-asm statement (the recogniser does not look inside), with no wait state (A), with
// `s_nop 0` (B) and with `s_nop 4` (C) between them, for three kinds of consumer -- a VOP2 read of the low result, a VOP3 read of the high result,
// a VOP3P read of the pair -- and each of the three packed producers.  The destination pair holds POISON before the producer, so a consumer that read the
// register file before the producer's write-back would see the poison.  Every variant is compared bit for bit with the same arithmetic done by the
// compiler (fmaf / plain multiplies and adds, -ffp-contract=off).  One line of JSON: pairs executed and mismatches per variant.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_forward pk_forward.hip        run: ./pk_forward [pairs, default 1048576]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd(uint32_t& s) { s = mix(s + 0x9e3779b9u); return (float)(int32_t)(s >> 8) * (1.0f / 4194304.0f) - 2.0f; }   // [-2, 2)

// One asm statement per (producer, gap): the destination pair v[20:21] is poisoned, a few unrelated instructions later the producer writes it, and
// the consumer follows in the very next slot (or after the gap).  Three statements per variant, one per kind of consumer:
//   c0: VOP2  v_add_f32  r, v20, k            (low half)
//   c1: VOP3P v_pk_mul_f32 r2, v[20:21], kk    (the pair)
//   c2: VOP3  v_fma_f32  r, v21, k, v20        (high half first, both halves)
#define POISON "v_mov_b32 v20, %[px]\n\tv_mov_b32 v21, %[py]\n\ts_nop 4\n\t"
#define P_FMA "v_pk_fma_f32 v[20:21], %[a], %[b], %[c]\n\t"
#define P_MUL "v_pk_mul_f32 v[20:21], %[a], %[b]\n\t"
#define P_ADD "v_pk_add_f32 v[20:21], %[a], %[b]\n\t"
#define G0 ""
#define G1 "s_nop 0\n\t"
#define G2 "s_nop 4\n\t"
#define IN : [a] "v"(a), [b] "v"(b), [c] "v"(c), [k] "v"(k), [kk] "v"(kk), [px] "v"(poison.x), [py] "v"(poison.y) : "v20", "v21", "v22", "v23"
// (the packed consumer writes the fixed pair v[22:23], copied out as two scalars after a long wait: element extraction from a 64-bit vector asm
// output was seen to read the low half twice)
#define VARIANT(P, G)                                                                             \
    asm volatile(POISON P G "v_add_f32 %[r], v20, %[k]" : [r] "=&v"(r0) IN);                       \
    asm volatile(POISON P G "v_pk_mul_f32 v[22:23], v[20:21], %[kk]\n\ts_nop 4\n\tv_mov_b32 %[rx], v22\n\tv_mov_b32 %[ry], v23" : [rx] "=&v"(r2x), [ry] "=&v"(r2y) IN); \
    asm volatile(POISON P G "v_fma_f32 %[r], v21, %[k], v20" : [r] "=&v"(r1) IN);

template <int PROD, int GAP>
__device__ __forceinline__ void one(f2 a, f2 b, f2 c, float k, f2 poison, uint32_t (&out)[4]) {
    float r0, r1, r2x, r2y; const f2 kk{k, k};
    if constexpr (PROD == 0 && GAP == 0) { VARIANT(P_FMA, G0) }
    if constexpr (PROD == 0 && GAP == 1) { VARIANT(P_FMA, G1) }
    if constexpr (PROD == 0 && GAP == 2) { VARIANT(P_FMA, G2) }
    if constexpr (PROD == 1 && GAP == 0) { VARIANT(P_MUL, G0) }
    if constexpr (PROD == 1 && GAP == 1) { VARIANT(P_MUL, G1) }
    if constexpr (PROD == 1 && GAP == 2) { VARIANT(P_MUL, G2) }
    if constexpr (PROD == 2 && GAP == 0) { VARIANT(P_ADD, G0) }
    if constexpr (PROD == 2 && GAP == 1) { VARIANT(P_ADD, G1) }
    if constexpr (PROD == 2 && GAP == 2) { VARIANT(P_ADD, G2) }
    out[0] = __builtin_bit_cast(uint32_t, r0);
    out[1] = __builtin_bit_cast(uint32_t, r2x);
    out[2] = __builtin_bit_cast(uint32_t, r2y);
    out[3] = __builtin_bit_cast(uint32_t, r1);
}

template <int PROD>
__device__ __forceinline__ void expect(f2 a, f2 b, f2 c, float k, uint32_t (&out)[4]) {
    f2 d;
    if constexpr (PROD == 0) d = f2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
    else if constexpr (PROD == 1) d = f2{a.x * b.x, a.y * b.y};
    else d = f2{a.x + b.x, a.y + b.y};
    out[0] = __builtin_bit_cast(uint32_t, d.x + k);
    out[1] = __builtin_bit_cast(uint32_t, d.x * k);
    out[2] = __builtin_bit_cast(uint32_t, d.y * k);
    out[3] = __builtin_bit_cast(uint32_t, __builtin_fmaf(d.y, k, d.x));
}

// mism[PROD * 3 + GAP] += results that differ from the compiler's arithmetic
template <int PROD>
__device__ __forceinline__ void run(uint32_t& s, unsigned long long* mism, unsigned long long (&local)[9]) {
    const f2 a{rnd(s), rnd(s)}, b{rnd(s), rnd(s)}, c{rnd(s), rnd(s)};
    const float k = rnd(s);
    const f2 poison{__builtin_bit_cast(float, 0x7fc0dead), __builtin_bit_cast(float, 0x7fc0beef)};     // NaNs: a stale read cannot pass for a result
    uint32_t want[4], got[4];
    expect<PROD>(a, b, c, k, want);
    one<PROD, 0>(a, b, c, k, poison, got);
    local[PROD * 3 + 0] += (got[0] != want[0]) + (got[1] != want[1]) + (got[2] != want[2]) + (got[3] != want[3]);
    one<PROD, 1>(a, b, c, k, poison, got);
    local[PROD * 3 + 1] += (got[0] != want[0]) + (got[1] != want[1]) + (got[2] != want[2]) + (got[3] != want[3]);
    one<PROD, 2>(a, b, c, k, poison, got);
    local[PROD * 3 + 2] += (got[0] != want[0]) + (got[1] != want[1]) + (got[2] != want[2]) + (got[3] != want[3]);
}

__global__ void k_pk_forward(int iters, unsigned long long* mism) {
    uint32_t s = mix(blockIdx.x * blockDim.x + threadIdx.x + 1u);
    unsigned long long local[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        run<0>(s, mism, local);
        run<1>(s, mism, local);
        run<2>(s, mism, local);
    }
    for (int j = 0; j < 9; ++j)
        if (local[j]) atomicAdd(&mism[j], local[j]);
}

int main(int argc, char** argv) {
    const long long want_pairs = argc > 1 ? atoll(argv[1]) : 1048576;
    const int threads = 256, blocks = 64;                                   // one wave per SIMD on 64 CUs' worth: lone waves, like the fused loops
    const int iters = (int)((want_pairs + (long long)threads * blocks - 1) / ((long long)threads * blocks));
    unsigned long long* d; unsigned long long h[9];
    CHECK(hipMalloc(&d, sizeof(h)));
    CHECK(hipMemset(d, 0, sizeof(h)));
    hipLaunchKernelGGL(k_pk_forward, dim3(blocks), dim3(threads), 0, 0, iters, d);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const long long lanes = (long long)threads * blocks * iters;
    printf("{\"lane_pairs_per_variant\": %lld, \"consumers_per_pair\": 4, \"mismatches\": {", lanes);
    const char* prod[3] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"};
    const char* gap[3] = {"no_wait_state", "s_nop_0", "s_nop_4"};
    for (int p = 0; p < 3; ++p)
        for (int g = 0; g < 3; ++g) printf("%s\"%s/%s\": %llu", (p + g) ? ", " : "", prod[p], gap[g], h[p * 3 + g]);
    printf("}}\n");
    return 0;
}
