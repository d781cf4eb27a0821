// How many wait states does gfx950 need between a VALU write of a VGPR and its read through DPP?  The ISA manual (and hipcc's hazard recogniser) say two.
// The split workgroups cannot have two from an `s_nop 1` anyway -- a partner's s_wakeup ends an s_nop after one state (snop_wakeup.hip) -- so the question decides
// between `s_nop 0` (one state, which nothing can shorten) and two `v_nop` in the hand-written DPP statements of dl_group.hpp (DL_DPP_WAIT).
// Every producer x consumer pair the kernels contain one state apart (tools/check_dpp_hazards.py on a -DDL_DPP_WAIT=1 listing: v_fma_f32 / v_mov_b32 ->
// v_max_f32_dpp / v_fmac_f32_dpp row_newbcast) and the neighbouring forms, with 0 (control), 1 and 2 states, alone and beside a wave that loops over s_wakeup,
// VALU + DPP work or MFMAs.  The register is overwritten with a marker first, so a stale read differs from a fresh one.
// usage: dpp_wait [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
enum { P_ADD, P_FMA, P_MOV, P_MUL, P_FMAC_DPP, P_RCP, P_CNDMASK, P_COUNT };
static const char* p_name[] = {"v_add_f32", "v_fma_f32", "v_mov_b32", "v_mul_f32", "v_fmac_f32_dpp (a link of a chain)", "v_rcp_f32", "v_cndmask_b32"};
enum { C_MOV_SHR1, C_MAX_NB2, C_FMAC_NB5, C_MOV_NB15, C_ADD_SHL4, C_MOV_QUAD, C_COUNT };
static const char* c_name[] = {"v_mov_b32_dpp row_shr:1", "v_max_f32_dpp row_newbcast:2", "v_fmac_f32_dpp row_newbcast:5", "v_mov_b32_dpp row_newbcast:15", "v_add_f32_dpp row_shl:4", "v_mov_b32_dpp quad_perm:[1,0,3,2]"};
#define PRE "v_cmp_gt_f32 vcc, %1, %2\n\tv_mov_b32 v20, 0x7fc01234\n\tv_mov_b32 v21, %3\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t"
#define P_STR_ADD "v_add_f32 v20, %1, %2\n\t"
#define P_STR_FMA "v_fma_f32 v20, -%1, %1, %2\n\t"
#define P_STR_MOV "v_mov_b32 v20, %1\n\t"
#define P_STR_MUL "v_mul_f32 v20, %1, %2\n\t"
#define P_STR_FMACD "v_mov_b32 v20, %1\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_fmac_f32_dpp v20, v20, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define P_STR_RCP "v_rcp_f32 v20, %2\n\t"
#define P_STR_CND "v_cndmask_b32 v20, %1, %2, vcc\n\t"
#define C_STR_MOV_SHR1 "v_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define C_STR_MAX_NB2 "v_max_f32_dpp %0, v20, v21 row_newbcast:2 row_mask:0xf bank_mask:0xf"
#define C_STR_MOV_NB15 "v_mov_b32_dpp %0, v20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
#define C_STR_ADD_SHL4 "v_add_f32_dpp %0, v20, v21 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define C_STR_MOV_QUAD "v_mov_b32_dpp %0, v20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "v_nop\n\tv_nop\n\t"
#define W6 "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t"
#define OPS : "=&v"(r) : "v"(a), "v"(b), "v"(c) : "v20", "v21", "vcc"
#define RUN(PS, CS, WS) asm volatile(PRE PS WS CS OPS)
#define RUN_FMAC(PS, WS) asm volatile(PRE "v_mov_b32 %0, v21\n\t" PS WS "v_fmac_f32_dpp %0, v20, v21 row_newbcast:5 row_mask:0xf bank_mask:0xf" OPS)
#define BY_WAIT(PS, CS) do { if constexpr (W == 0) RUN(PS, CS, W0); else if constexpr (W == 1) RUN(PS, CS, W1); else if constexpr (W == 2) RUN(PS, CS, W2); else RUN(PS, CS, W6); } while (0)
#define BY_WAIT_FMAC(PS) do { if constexpr (W == 0) RUN_FMAC(PS, W0); else if constexpr (W == 1) RUN_FMAC(PS, W1); else if constexpr (W == 2) RUN_FMAC(PS, W2); else RUN_FMAC(PS, W6); } while (0)
#define BY_CONS(PS) do { if constexpr (C == C_MOV_SHR1) BY_WAIT(PS, C_STR_MOV_SHR1); else if constexpr (C == C_MAX_NB2) BY_WAIT(PS, C_STR_MAX_NB2); else if constexpr (C == C_FMAC_NB5) BY_WAIT_FMAC(PS); \
    else if constexpr (C == C_MOV_NB15) BY_WAIT(PS, C_STR_MOV_NB15); else if constexpr (C == C_ADD_SHL4) BY_WAIT(PS, C_STR_ADD_SHL4); else BY_WAIT(PS, C_STR_MOV_QUAD); } while (0)
template <int P, int C, int W> __device__ __forceinline__ float pc(float a, float b, float c) {
    float r;
    if constexpr (P == P_ADD) BY_CONS(P_STR_ADD); else if constexpr (P == P_FMA) BY_CONS(P_STR_FMA); else if constexpr (P == P_MOV) BY_CONS(P_STR_MOV); else if constexpr (P == P_MUL) BY_CONS(P_STR_MUL);
    else if constexpr (P == P_FMAC_DPP) BY_CONS(P_STR_FMACD); else if constexpr (P == P_RCP) BY_CONS(P_STR_RCP); else BY_CONS(P_STR_CND);
    return r;
}
template <int P, int C>
__global__ __launch_bounds__(512) void k(const float* A, const float* B, unsigned long long* bad, int iters, int neighbour) {
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x, wave = threadIdx.x >> 6;
    float a = A[t], b = B[t];
    if (wave < 4) {
        unsigned long long cnt[3] = {0, 0, 0};
        for (int it = 0; it < iters; it++) {
            const float c = a * 0.37f - b;
            const unsigned ref = __float_as_uint(pc<P, C, 6>(a, b, c));
            cnt[0] += __float_as_uint(pc<P, C, 0>(a, b, c)) != ref;
            cnt[1] += __float_as_uint(pc<P, C, 1>(a, b, c)) != ref;
            cnt[2] += __float_as_uint(pc<P, C, 2>(a, b, c)) != ref;
            a = a * 1.0001f + 0.001f; b = b * 0.9999f - 0.002f;
            for (int d = 0; d < ((wave * 7 + it) & 7); d++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(0.0f));
        }
        for (int i = 0; i < 3; i++) if (cnt[i]) atomicAdd(&bad[i], cnt[i]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&done, 1);
    } else {
        float x = a + 1.5f, y = b + 1.5f;
        while (*(volatile int*)&done < 4) {
            for (int i = 0; i < 8; i++) {
                if (neighbour == 1) asm volatile("s_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3");
                else if (neighbour == 3) { typedef float f4 __attribute__((ext_vector_type(4))); f4 a0 = {x, y, x, y}, a1 = a0;
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %3, %2, %1\n\tv_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %3, %2, %1\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
                    x = a0[0] * 1e-30f + 0.5f; y = a1[1] * 1e-30f + 0.25f; }
                else asm volatile("v_add_f32 %0, %0, %1\n\tv_fma_f32 %1, %0, %1, %0\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32_dpp %1, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_rcp_f32 %0, %0\n\tv_fma_f32 %1, %0, %1, %0" : "+v"(x), "+v"(y));
            }
        }
        if (x + y == 123.456f) bad[7] = 1;
    }
}
template <int P, int C>
static void run(const float* A, const float* B, unsigned long long* bad, int iters, int neighbour, unsigned long long* tot) {
    hipMemset(bad, 0, 64);
    hipLaunchKernelGGL((k<P, C>), dim3(256), dim3(neighbour == 0 ? 256 : 512), 0, 0, A, B, bad, iters, neighbour);
    unsigned long long hb[3]; hipMemcpy(hb, bad, 24, hipMemcpyDeviceToHost);
    printf("  %10llu %8llu %8llu", hb[0], hb[1], hb[2]);
    tot[0] += hb[0]; tot[1] += hb[1]; tot[2] += hb[2];
}
template <int P, int C>
static void row(const float* A, const float* B, unsigned long long* bad, int iters, unsigned long long* tot) {
    printf("%-36s -> %-36s", p_name[P], c_name[C]);
    for (int nb = 0; nb < 4; nb++) run<P, C>(A, B, bad, iters, nb, tot);
    printf("\n");
}
template <int P>
static void rows(const float* A, const float* B, unsigned long long* bad, int iters, unsigned long long* tot) {
    row<P, C_MOV_SHR1>(A, B, bad, iters, tot); row<P, C_MAX_NB2>(A, B, bad, iters, tot); row<P, C_FMAC_NB5>(A, B, bad, iters, tot);
    row<P, C_MOV_NB15>(A, B, bad, iters, tot); row<P, C_ADD_SHL4>(A, B, bad, iters, tot); row<P, C_MOV_QUAD>(A, B, bad, iters, tot);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    const int n = 256 * 512;
    float *A, *B; unsigned long long* bad;
    hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&bad, 64);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 977) / 977.f - 0.5f;
    hipMemcpy(A, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 613) / 613.f + 0.1f;
    hipMemcpy(B, h, n * 4, hipMemcpyHostToDevice);
    printf("stale DPP reads of %lld lane-reads per cell; columns: [alone | beside s_wakeup | beside VALU + DPP work | beside MFMA 16x16x4 + 4x4x1] x [0 | 1 (s_nop 0) | 2 (2 x v_nop)] wait states, against 6 x v_nop\n", 256LL * 256 * iters);
    unsigned long long tot[3] = {0, 0, 0};
    rows<P_ADD>(A, B, bad, iters, tot); rows<P_FMA>(A, B, bad, iters, tot); rows<P_MOV>(A, B, bad, iters, tot); rows<P_MUL>(A, B, bad, iters, tot);
    rows<P_FMAC_DPP>(A, B, bad, iters, tot); rows<P_RCP>(A, B, bad, iters, tot); rows<P_CNDMASK>(A, B, bad, iters, tot);
    printf("totals: no wait %llu, one state %llu, two states %llu\n", tot[0], tot[1], tot[2]);
    return 0;
}
