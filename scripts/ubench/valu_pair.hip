// Microbenchmark: which ADJACENT instructions of one wave does a gfx950 SIMD issue as a pair?
// profiles/r01_ubench_valu_issue.txt: 16 independent v_fma_f32 cost 2.46 cycles each, a dependent chain 4.3-4.9 -- at ANY number of waves per SIMD, so the
// cheap rate is a property of neighbouring instructions of the same wave, not of latency hiding.  This file prices fixed instruction sequences written in
// assembly (64 instructions per loop iteration, 16 destination registers, sources from 8 other registers) so that the compiler cannot reorder them:
//   build: hipcc --offload-arch=gfx950 -O3 -o valu_pair valu_pair.hip ;  run: ./valu_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;      // (the 1024-instruction bodies run ITER / 4 iterations)

// one iteration = BODY (64 instructions); registers: destinations v[16..31], sources v[8..15], s[20..23]
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)

#define KERNEL(NAME, BODY)                                                                                             \
    __global__ void NAME(float* out, int iters, unsigned long long* clk) {                                              \
        float acc = 0.0f;                                                                                               \
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();                                \
        asm volatile(                                                                                                   \
            "v_cvt_f32_u32 v8, %2\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 0.5\n v_mov_b32 v11, 2.0\n"                      \
            "v_mov_b32 v12, 0x3f800100\n v_mov_b32 v13, 3\n v_mov_b32 v14, 0x7060302\n v_mov_b32 v15, 0xffff0000\n"    \
            "s_mov_b32 s20, 0x3f800001\n s_mov_b32 s21, 16\n"                                                           \
            "v_mov_b32 v16, v8\n v_mov_b32 v17, v8\n v_mov_b32 v18, v8\n v_mov_b32 v19, v8\n v_mov_b32 v20, v8\n v_mov_b32 v21, v8\n"  \
            "v_mov_b32 v22, v8\n v_mov_b32 v23, v8\n v_mov_b32 v24, v8\n v_mov_b32 v25, v8\n v_mov_b32 v26, v8\n v_mov_b32 v27, v8\n"  \
            "v_mov_b32 v28, v8\n v_mov_b32 v29, v8\n v_mov_b32 v30, v8\n v_mov_b32 v31, v8\n"                           \
            "s_mov_b32 s22, %1\n"                                                                                      \
            "1:\n" BODY                                                                                                 \
            "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n"                                         \
            "v_add_f32 %0, v16, v17\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v31\n"          \
            : "=v"(acc) : "s"(iters), "v"(threadIdx.x)                                                                                   \
            : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", \
              "v27", "v28", "v29", "v30", "v31", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "s20", "s21", "s22", "s23", "vcc", "scc", "memory");                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; } \
    }

// 16 independent instructions with destinations v16..v31; OP(d) expands to one instruction writing v<d>
#define SEQ16(OP) OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23) OP(24) OP(25) OP(26) OP(27) OP(28) OP(29) OP(30) OP(31)
// alternate two kinds
#define ALT16(A, B) A(16) B(17) A(18) B(19) A(20) B(21) A(22) B(23) A(24) B(25) A(26) B(27) A(28) B(29) A(30) B(31)

#define FMA(d) "v_fma_f32 v" #d ", v" #d ", v9, v10\n"
#define FMAS(d) "v_fma_f32 v" #d ", v" #d ", s20, v10\n"
#define FMAC(d) "v_fmac_f32 v" #d ", v9, v10\n"
#define ADDF(d) "v_add_f32 v" #d ", v" #d ", v10\n"
#define MULF(d) "v_mul_f32 v" #d ", v" #d ", v12\n"
#define MAXF(d) "v_max_f32 v" #d ", v" #d ", v10\n"
#define AND(d) "v_and_b32 v" #d ", v" #d ", v15\n"
#define ANDL(d) "v_and_b32 v" #d ", 0xffff0000, v" #d "\n"
#define SHL(d) "v_lshlrev_b32 v" #d ", 16, v" #d "\n"
#define ADDU(d) "v_add_u32 v" #d ", v" #d ", v13\n"
#define XOR(d) "v_xor_b32 v" #d ", v" #d ", v13\n"
#define BFE(d) "v_bfe_u32 v" #d ", v" #d ", 3, 16\n"
#define PERM(d) "v_perm_b32 v" #d ", v" #d ", v9, v14\n"
#define MOV(d) "v_mov_b32 v" #d ", v9\n"
#define CND(d) "v_cndmask_b32 v" #d ", v" #d ", v9, vcc\n"
#define MULLO(d) "v_mul_lo_u32 v" #d ", v" #d ", v13\n"
#define MULHI(d) "v_mul_hi_u32 v" #d ", v" #d ", v13\n"
#define EXP(d) "v_exp_f32 v" #d ", v" #d "\n"
#define RCP(d) "v_rcp_f32 v" #d ", v" #d "\n"
#define CVT(d) "v_cvt_f32_u32 v" #d ", v" #d "\n"
#define DPP(d) "v_mov_b32_dpp v" #d ", v" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define ADDDPP(d) "v_add_f32_dpp v" #d ", v" #d ", v" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define MED3(d) "v_med3_f32 v" #d ", v" #d ", v9, v11\n"
#define CMP(d) "v_cmp_lt_f32 vcc, v" #d ", v9\n"
#define RDL(d) "v_readlane_b32 s23, v" #d ", 3\n"
#define MULS(d) "v_mul_f32 v" #d ", s20, v" #d "\n"
#define MUL24(d) "v_mul_u32_u24 v" #d ", 0x10000, v" #d "\n"
#define MAD24(d) "v_mad_u32_u24 v" #d ", v" #d ", v13, v13\n"
#define SHR(d) "v_lshrrev_b32 v" #d ", 16, v" #d "\n"
#define OR(d) "v_or_b32 v" #d ", v" #d ", v13\n"
#define SUBF(d) "v_sub_f32 v" #d ", v" #d ", v10\n"
#define MINF(d) "v_min_f32 v" #d ", v" #d ", v10\n"
#define LSHLOR(d) "v_lshl_or_b32 v" #d ", v" #d ", 16, v13\n"
#define ANDOR(d) "v_and_or_b32 v" #d ", v" #d ", v15, v13\n"
#define BFI(d) "v_bfi_b32 v" #d ", v15, v" #d ", v13\n"
#define ALIGN(d) "v_alignbit_b32 v" #d ", v" #d ", v13, 16\n"
#define CVTBF(d) "v_cvt_f32_bf16 v" #d ", v" #d "\n"
#define BITOP3(d) "v_bitop3_b32 v" #d ", v" #d ", v13, v15 bitop3:0x6c\n"
#define FMAAK(d) "v_fmaak_f32 v" #d ", v" #d ", v9, 0x3f800100\n"
#define FMAMK(d) "v_fmamk_f32 v" #d ", v" #d ", 0x3f800100, v10\n"
#define RNDNE(d) "v_rndne_f32 v" #d ", v" #d "\n"
#define LDEXP(d) "v_ldexp_f32 v" #d ", v" #d ", v13\n"
#define DIVSC(d) "v_div_scale_f32 v" #d ", vcc, v" #d ", v9, v" #d "\n"
#define DIVFX(d) "v_div_fixup_f32 v" #d ", v" #d ", v9, v10\n"
#define WRL(d) "v_writelane_b32 v" #d ", s20, 3\n"
#define ACCW(d) "v_accvgpr_write_b32 a" #d ", v" #d "\n"
#define ACCR(d) "v_accvgpr_read_b32 v" #d ", a" #d "\n"
#define CMPCND(d) "v_cmp_lt_f32 vcc, v" #d ", v9\n v_cndmask_b32 v" #d ", v" #d ", v10, vcc\n"
#define SEQ8C(OP) OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23)
#define PKADD(d, e) "v_pk_add_f32 v[" #d ":" #e "], v[" #d ":" #e "], v[8:9]\n"
#define MAD64(d, e) "v_mad_u64_u32 v[" #d ":" #e "], vcc, v" #d ", v13, v[10:11]\n"
#define SDWA(d) "v_add_u32_sdwa v" #d ", v" #d ", v13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"

// packed: destinations are pairs v[16:17] ... v[30:31]; 8 per SEQ
#define PK(d, e) "v_pk_fma_f32 v[" #d ":" #e "], v[" #d ":" #e "], v[8:9], v[10:11]\n"
#define PKMUL(d, e) "v_pk_mul_f32 v[" #d ":" #e "], v[" #d ":" #e "], v[8:9]\n"
#define SEQ8PK(P) P(16, 17) P(18, 19) P(20, 21) P(22, 23) P(24, 25) P(26, 27) P(28, 29) P(30, 31)
// packed fma on pair (d,e) followed by a plain op on another register
#define PK_THEN(P, O) P(16, 17) O(24) P(18, 19) O(25) P(20, 21) O(26) P(22, 23) O(27) P(16, 17) O(28) P(18, 19) O(29) P(20, 21) O(30) P(22, 23) O(31)

// dependent pairs: B reads what A just wrote.  PAIRS_ADJ = A1 B1 A2 B2 ..., PAIRS_SPLIT = A1 A2 B1 B2 ...
#define DEP2(d) "v_fma_f32 v" #d ", v" #d ", v9, v10\n v_fma_f32 v" #d ", v" #d ", v9, v10\n"
#define SEQ8DEP(D) D(16) D(17) D(18) D(19) D(20) D(21) D(22) D(23)
#define SPLIT2(d, e) "v_fma_f32 v" #d ", v" #d ", v9, v10\n v_fma_f32 v" #e ", v" #e ", v9, v10\n v_fma_f32 v" #d ", v" #d ", v9, v10\n v_fma_f32 v" #e ", v" #e ", v9, v10\n"
#define SEQ4SPLIT SPLIT2(16, 17) SPLIT2(18, 19) SPLIT2(20, 21) SPLIT2(22, 23)
// one dependent chain everywhere
#define CHAIN(d) "v_fma_f32 v16, v16, v9, v10\n"
// distance-2 dependence: A B A' B' where A' depends on A (two interleaved chains)
#define TWO(d) "v_fma_f32 v16, v16, v9, v10\n v_fma_f32 v17, v17, v9, v10\n"
#define SEQ8(X) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0)
#define THREE(d) "v_fma_f32 v16, v16, v9, v10\n v_fma_f32 v17, v17, v9, v10\n v_fma_f32 v18, v18, v9, v10\n"
#define FOUR(d) "v_fma_f32 v16, v16, v9, v10\n v_fma_f32 v17, v17, v9, v10\n v_fma_f32 v18, v18, v9, v10\n v_fma_f32 v19, v19, v9, v10\n"

KERNEL(k_fma, R4(SEQ16(FMA)))
KERNEL(k_fma_256, R16(SEQ16(FMA)))
KERNEL(k_fma_1024, R16(R4(SEQ16(FMA))))
KERNEL(k_pk_1024, R16(R8(SEQ8PK(PK))))
KERNEL(k_bfe_1024, R16(R4(SEQ16(BFE))))
KERNEL(k_mix_1024, R16(R4(ALT16(FMA, BFE))) )
KERNEL(k_chain_1024, R16(R4(SEQ16(CHAIN))))
KERNEL(k_fma_sgpr, R4(SEQ16(FMAS)))
KERNEL(k_fmac, R4(SEQ16(FMAC)))
KERNEL(k_add_mul, R4(ALT16(ADDF, MULF)))
KERNEL(k_max, R4(SEQ16(MAXF)))
KERNEL(k_and, R4(SEQ16(AND)))
KERNEL(k_and_lit, R4(SEQ16(ANDL)))
KERNEL(k_shl_and, R4(ALT16(SHL, AND)))
KERNEL(k_addu, R4(SEQ16(ADDU)))
KERNEL(k_xor, R4(SEQ16(XOR)))
KERNEL(k_bfe, R4(SEQ16(BFE)))
KERNEL(k_perm, R4(SEQ16(PERM)))
KERNEL(k_mov, R4(SEQ16(MOV)))
KERNEL(k_cnd, R4(SEQ16(CND)))
KERNEL(k_mullo, R4(SEQ16(MULLO)))
KERNEL(k_mulhi, R4(SEQ16(MULHI)))
KERNEL(k_exp, R4(SEQ16(EXP)))
KERNEL(k_rcp, R4(SEQ16(RCP)))
KERNEL(k_cvt, R4(SEQ16(CVT)))
KERNEL(k_dpp, R4(SEQ16(DPP)))
KERNEL(k_add_dpp, R4(SEQ16(ADDDPP)))
KERNEL(k_med3, R4(SEQ16(MED3)))
KERNEL(k_sdwa, R4(SEQ16(SDWA)))
KERNEL(k_mul_sgpr, R4(SEQ16(MULS)))
KERNEL(k_mul24, R4(SEQ16(MUL24)))
KERNEL(k_mad24, R4(SEQ16(MAD24)))
KERNEL(k_shl, R4(SEQ16(SHL)))
KERNEL(k_shr, R4(SEQ16(SHR)))
KERNEL(k_or, R4(SEQ16(OR)))
KERNEL(k_sub, R4(SEQ16(SUBF)))
KERNEL(k_min, R4(SEQ16(MINF)))
KERNEL(k_lshlor, R4(SEQ16(LSHLOR)))
KERNEL(k_andor, R4(SEQ16(ANDOR)))
KERNEL(k_bfi, R4(SEQ16(BFI)))
KERNEL(k_align, R4(SEQ16(ALIGN)))
KERNEL(k_cvtbf, R4(SEQ16(CVTBF)))
KERNEL(k_bitop3, R4(SEQ16(BITOP3)))
KERNEL(k_fmaak, R4(SEQ16(FMAAK)))
KERNEL(k_fmamk, R4(SEQ16(FMAMK)))
KERNEL(k_rndne, R4(SEQ16(RNDNE)))
KERNEL(k_ldexp, R4(SEQ16(LDEXP)))
KERNEL(k_divsc, R4(SEQ16(DIVSC)))
KERNEL(k_divfx, R4(SEQ16(DIVFX)))
KERNEL(k_wrl, R4(SEQ16(WRL)))
KERNEL(k_accw, R4(SEQ16(ACCW)))
KERNEL(k_accr, R4(SEQ16(ACCR)))
KERNEL(k_cmpcnd, R8(SEQ8C(CMPCND)))
KERNEL(k_pkadd, R8(SEQ8PK(PKADD)))
KERNEL(k_mad64, R8(SEQ8PK(MAD64)))
KERNEL(k_mul24_and, R4(ALT16(MUL24, AND)))
KERNEL(k_fma_and, R4(ALT16(FMA, AND)))
KERNEL(k_fma_bfe, R4(ALT16(FMA, BFE)))
KERNEL(k_fma_exp, R4(ALT16(FMA, EXP)))
KERNEL(k_fma_cnd, R4(ALT16(FMA, CND)))
KERNEL(k_fma_mullo, R4(ALT16(FMA, MULLO)))
KERNEL(k_fma_dpp, R4(ALT16(FMA, DPP)))
KERNEL(k_fma_cmp, R4(ALT16(FMA, CMP)))
KERNEL(k_fma_rdl, R4(ALT16(FMA, RDL)))
KERNEL(k_pk, R8(SEQ8PK(PK)))
KERNEL(k_pkmul, R8(SEQ8PK(PKMUL)))
KERNEL(k_pk_fma, R4(PK_THEN(PK, FMA)))
KERNEL(k_pk_and, R4(PK_THEN(PK, AND)))
KERNEL(k_dep_adjacent, R4(SEQ8DEP(DEP2)))
KERNEL(k_dep_split, R4(SEQ4SPLIT))
KERNEL(k_chain1, R4(SEQ16(CHAIN)))
KERNEL(k_chain2, R4(SEQ8(TWO)))
KERNEL(k_chain3, R4(SEQ8(THREE)) R4(SEQ8(THREE)) R4(SEQ8(THREE)))
KERNEL(k_chain4, R4(SEQ8(FOUR)) R4(SEQ8(FOUR)))

typedef void (*kern_t)(float*, int, unsigned long long*);
struct Case { const char* name; kern_t fn; int per_iter; };

int main() {
    float* d_out; unsigned long long* d_clk;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8)); CHECK(hipMalloc(&d_clk, 16));
    const Case cases[] = {
        {"v_fma_f32 x16 indep", k_fma, 64}, {"v_fma_f32 x16, 256 per iteration", k_fma_256, 256}, {"v_fma_f32 x16, 1024 per iteration", k_fma_1024, 1024},
        {"v_pk_fma_f32, 1024 per iteration", k_pk_1024, 1024}, {"v_bfe_u32, 1024 per iteration", k_bfe_1024, 1024}, {"fma / bfe, 1024 per iteration", k_mix_1024, 1024},
        {"1 chain, 1024 per iteration", k_chain_1024, 1024},
        {"v_fma_f32 sgpr operand", k_fma_sgpr, 64}, {"v_fmac_f32", k_fmac, 64},
        {"v_add_f32 / v_mul_f32", k_add_mul, 64}, {"v_max_f32", k_max, 64}, {"v_and_b32", k_and, 64}, {"v_and_b32 literal", k_and_lit, 64},
        {"v_lshlrev / v_and (open a pair)", k_shl_and, 64}, {"v_add_u32", k_addu, 64}, {"v_xor_b32", k_xor, 64}, {"v_bfe_u32", k_bfe, 64},
        {"v_perm_b32", k_perm, 64}, {"v_mov_b32", k_mov, 64}, {"v_cndmask_b32", k_cnd, 64}, {"v_mul_lo_u32", k_mullo, 64}, {"v_mul_hi_u32", k_mulhi, 64},
        {"v_exp_f32", k_exp, 64}, {"v_rcp_f32", k_rcp, 64}, {"v_cvt_f32_u32", k_cvt, 64}, {"v_mov_b32_dpp row_shr", k_dpp, 64}, {"v_add_f32_dpp row_shr", k_add_dpp, 64},
        {"v_med3_f32", k_med3, 64}, {"v_add_u32_sdwa", k_sdwa, 64},
        {"v_mul_f32 sgpr operand (VOP2)", k_mul_sgpr, 64}, {"v_mul_u32_u24 literal", k_mul24, 64}, {"v_mad_u32_u24", k_mad24, 64}, {"v_lshlrev_b32", k_shl, 64},
        {"v_lshrrev_b32", k_shr, 64}, {"v_or_b32", k_or, 64}, {"v_sub_f32", k_sub, 64}, {"v_min_f32", k_min, 64}, {"v_lshl_or_b32", k_lshlor, 64},
        {"v_and_or_b32", k_andor, 64}, {"v_bfi_b32", k_bfi, 64}, {"v_alignbit_b32", k_align, 64}, {"v_cvt_f32_bf16", k_cvtbf, 64}, {"v_bitop3_b32", k_bitop3, 64},
        {"v_fmaak_f32", k_fmaak, 64}, {"v_fmamk_f32", k_fmamk, 64}, {"v_rndne_f32", k_rndne, 64}, {"v_ldexp_f32", k_ldexp, 64}, {"v_div_scale_f32", k_divsc, 64},
        {"v_div_fixup_f32", k_divfx, 64}, {"v_writelane_b32", k_wrl, 64}, {"v_accvgpr_write_b32", k_accw, 64}, {"v_accvgpr_read_b32", k_accr, 64},
        {"v_cmp + v_cndmask pairs", k_cmpcnd, 128}, {"v_pk_add_f32", k_pkadd, 64}, {"v_mad_u64_u32", k_mad64, 64}, {"mul_u32_u24 / and (open a pair)", k_mul24_and, 64},
        {"fma / and alternating", k_fma_and, 64}, {"fma / bfe alternating", k_fma_bfe, 64}, {"fma / exp alternating", k_fma_exp, 64}, {"fma / cndmask alternating", k_fma_cnd, 64},
        {"fma / mul_lo alternating", k_fma_mullo, 64}, {"fma / mov_dpp alternating", k_fma_dpp, 64}, {"fma / v_cmp alternating", k_fma_cmp, 64},
        {"fma / v_readlane alternating", k_fma_rdl, 64},
        {"v_pk_fma_f32 x8 indep", k_pk, 64}, {"v_pk_mul_f32 x8 indep", k_pkmul, 64}, {"pk_fma / fma alternating", k_pk_fma, 64}, {"pk_fma / and alternating", k_pk_and, 64},
        {"dependent pairs A1 B1 A2 B2", k_dep_adjacent, 64}, {"dependent pairs A1 A2 B1 B2", k_dep_split, 64},
        {"1 chain", k_chain1, 64}, {"2 interleaved chains", k_chain2, 64}, {"3 interleaved chains", k_chain3, 288}, {"4 interleaved chains", k_chain4, 256},
    };
    printf("%-36s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD at 2.4 GHz; 1 / 2 / 4 / 8 waves per SIMD)\n", "sequence", "1", "2", "4", "8");
    for (const Case& c : cases) {
        printf("%-36s", c.name);
        double mhz[4]; int n_mhz = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, d_out, 10, d_clk);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            const int iters = c.per_iter >= 1024 ? ITER / 4 : ITER;
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h_clk[2]; CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
            // wave 0's own clocks: s_memtime ticks / (s_memrealtime ticks at 100 MHz) = the rate s_memtime counts at during the run (MHz)
            mhz[n_mhz++] = h_clk[1] ? 100.0 * (double)h_clk[0] / (double)h_clk[1] : 0.0;
            printf(" %8.2f", ms * 1e6 / ((double)iters * c.per_iter * wps) * 2.4);
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        }
        printf("   s_memtime MHz %.0f %.0f %.0f %.0f\n", mhz[0], mhz[1], mhz[2], mhz[3]);
    }
    return 0;
}
