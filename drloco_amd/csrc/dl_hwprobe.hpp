// dl_hwprobe.hpp -- the two hardware facts the hand-written wait states of this library rest on, as kernels that can FAIL (no reference counterpart: the
// reference has no device code).  Shared by the library (dl_hw_probe: the short form -- the guard of the one-wait-state build in dl_create and a -m gpu test) and by the
// long-form micro benchmarks tools/ubench/dpp_wait.hip / snop_wakeup.hip (profiles/r05_dpp_wait.txt, r05_snop_wakeup.txt).
//
//  (1) VALU write -> DPP read of the same VGPR: the gfx9 / CDNA ISA manual asks for TWO wait states; measured on gfx950: stale reads with none, never with ONE.
//      k_dpp<P, C>: producer P writes v20 (which held a marker), W wait states, consumer C reads v20 through DPP; compared with the same pair six states apart.
//      Producers x consumers: every pair the product listing contains one state apart (tools/check_dpp_hazards.py --pairs) and the neighbouring forms.
//  (2) an s_wakeup executed by ANOTHER wave of the workgroup ends the s_nop this wave is in after one wait state, whatever its count: a reader of an MFMA result
//      behind a single `s_nop 7` sees the stale accumulator beside a wave that loops over s_wakeup; never behind `v_nop`s or behind two s_nop instructions.
//      k_snop<WK, DK>: a chain of v_mfma_f32_4x4x1, the wait WK, an LDS store of the result; and v_add_f32 -> DK -> v_mov_b32_dpp.
#pragma once
#include <hip/hip_runtime.h>

namespace dl {
namespace hwprobe {

enum { P_ADD, P_FMA, P_MOV, P_MUL, P_FMAC_DPP, P_RCP, P_CNDMASK, P_COUNT };
enum { C_MOV_SHR1, C_MAX_NB2, C_FMAC_NB5, C_MOV_NB15, C_ADD_SHL4, C_MOV_QUAD, C_COUNT };
enum { NB_NONE, NB_WAKE, NB_VALU, NB_MFMA, NB_COUNT };          // what waves 4 .. 7 of the workgroup do beside the test waves 0 .. 3

#define HWP_PRE "v_cmp_gt_f32 vcc, %1, %2\n\tv_mov_b32 v20, 0x7fc01234\n\tv_mov_b32 v21, %3\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t"
#define HWP_P_ADD "v_add_f32 v20, %1, %2\n\t"
#define HWP_P_FMA "v_fma_f32 v20, -%1, %1, %2\n\t"
#define HWP_P_MOV "v_mov_b32 v20, %1\n\t"
#define HWP_P_MUL "v_mul_f32 v20, %1, %2\n\t"
#define HWP_P_FMACD "v_mov_b32 v20, %1\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_fmac_f32_dpp v20, v20, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define HWP_P_RCP "v_rcp_f32 v20, %2\n\t"
#define HWP_P_CND "v_cndmask_b32 v20, %1, %2, vcc\n\t"
#define HWP_C_MOV_SHR1 "v_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define HWP_C_MAX_NB2 "v_max_f32_dpp %0, v20, v21 row_newbcast:2 row_mask:0xf bank_mask:0xf"
#define HWP_C_MOV_NB15 "v_mov_b32_dpp %0, v20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
#define HWP_C_ADD_SHL4 "v_add_f32_dpp %0, v20, v21 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define HWP_C_MOV_QUAD "v_mov_b32_dpp %0, v20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define HWP_W0 ""
#define HWP_W1 "s_nop 0\n\t"
#define HWP_W2 "v_nop\n\tv_nop\n\t"
#define HWP_W6 "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t"
#define HWP_OPS : "=&v"(r) : "v"(a), "v"(b), "v"(c) : "v20", "v21", "vcc"
#define HWP_RUN(PS, CS, WS) asm volatile(HWP_PRE PS WS CS HWP_OPS)
#define HWP_RUN_FMAC(PS, WS) asm volatile(HWP_PRE "v_mov_b32 %0, v21\n\t" PS WS "v_fmac_f32_dpp %0, v20, v21 row_newbcast:5 row_mask:0xf bank_mask:0xf" HWP_OPS)
#define HWP_BY_WAIT(PS, CS) do { if constexpr (W == 0) HWP_RUN(PS, CS, HWP_W0); else if constexpr (W == 1) HWP_RUN(PS, CS, HWP_W1); else if constexpr (W == 2) HWP_RUN(PS, CS, HWP_W2); else HWP_RUN(PS, CS, HWP_W6); } while (0)
#define HWP_BY_WAIT_FMAC(PS) do { if constexpr (W == 0) HWP_RUN_FMAC(PS, HWP_W0); else if constexpr (W == 1) HWP_RUN_FMAC(PS, HWP_W1); else if constexpr (W == 2) HWP_RUN_FMAC(PS, HWP_W2); else HWP_RUN_FMAC(PS, HWP_W6); } while (0)
#define HWP_BY_CONS(PS) do { if constexpr (C == C_MOV_SHR1) HWP_BY_WAIT(PS, HWP_C_MOV_SHR1); else if constexpr (C == C_MAX_NB2) HWP_BY_WAIT(PS, HWP_C_MAX_NB2); else if constexpr (C == C_FMAC_NB5) HWP_BY_WAIT_FMAC(PS); \
    else if constexpr (C == C_MOV_NB15) HWP_BY_WAIT(PS, HWP_C_MOV_NB15); else if constexpr (C == C_ADD_SHL4) HWP_BY_WAIT(PS, HWP_C_ADD_SHL4); else HWP_BY_WAIT(PS, HWP_C_MOV_QUAD); } while (0)

// producer P, W wait states (0, 1 = `s_nop 0`, 2 = two v_nop, otherwise six v_nop: the reference), consumer C
template <int P, int C, int W> __device__ __forceinline__ float pc(float a, float b, float c) {
    float r;
    if constexpr (P == P_ADD) HWP_BY_CONS(HWP_P_ADD); else if constexpr (P == P_FMA) HWP_BY_CONS(HWP_P_FMA); else if constexpr (P == P_MOV) HWP_BY_CONS(HWP_P_MOV); else if constexpr (P == P_MUL) HWP_BY_CONS(HWP_P_MUL);
    else if constexpr (P == P_FMAC_DPP) HWP_BY_CONS(HWP_P_FMACD); else if constexpr (P == P_RCP) HWP_BY_CONS(HWP_P_RCP); else HWP_BY_CONS(HWP_P_CND);
    return r;
}

// the neighbour waves' loop (waves 4 .. 7 of the workgroup, one per SIMD beside the test waves) until the test waves are done
__device__ __forceinline__ void neighbour_loop(volatile int* done, int neighbour, float x, float y, unsigned long long* bad) {
    while (*done < 4) {
        for (int i = 0; i < 8; i++) {
            if (neighbour == NB_WAKE) asm volatile("s_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3");
            else if (neighbour == NB_MFMA) { typedef float f4 __attribute__((ext_vector_type(4))); f4 a0 = {x, y, x, y}, a1 = a0;
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %3, %2, %1\n\tv_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %3, %2, %1\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
                x = a0[0] * 1e-30f + 0.5f; y = a1[1] * 1e-30f + 0.25f; }
            else asm volatile("v_add_f32 %0, %0, %1\n\tv_fma_f32 %1, %0, %1, %0\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32_dpp %1, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_rcp_f32 %0, %0\n\tv_fma_f32 %1, %0, %1, %0" : "+v"(x), "+v"(y));
        }
    }
    if (x + y == 123.456f) bad[7] = 1;
}

// bad[0 .. 2] += stale DPP reads with 0 / 1 / 2 wait states (256 test lanes per workgroup x iters each)
template <int P, int C>
__global__ __launch_bounds__(512) void k_dpp(const float* A, const float* B, unsigned long long* bad, int iters, int neighbour) {
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
    } else neighbour_loop((volatile int*)&done, neighbour, a + 1.5f, b + 1.5f, bad);
}

// ---- (2) s_nop beside s_wakeup
enum { W_NOP7, W_NOP3x2, W_VNOP8, W_NOP2, W_NOP0x3, W_COUNT };
enum { D_NOP1, D_VNOP2, D_NOP0x2, D_NONE, D_NOP0, D_COUNT };
enum { N_NONE, N_SLEEP, N_WAKE, N_SLOAD, N_SLEEPWAKE, N_COUNT };
#define HWP_M4 "v_mfma_f32_4x4x1_16b_f32 v[20:23], %5, %6, v[20:23]\n\t"
#define HWP_CHAIN4 HWP_M4 "s_nop 1\n\t" HWP_M4 "s_nop 1\n\t" HWP_M4 "s_nop 1\n\t" HWP_M4 "s_nop 1\n\t" HWP_M4 "s_nop 1\n\t" HWP_M4 "s_nop 1\n\t" HWP_M4 "s_nop 1\n\t" HWP_M4
#define HWP_INIT "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %4\n\ts_nop 4\n\t"
#define HWP_READ_DS "ds_write2_b32 %7, v22, v23 offset0:2 offset1:3\n\tds_write2_b32 %7, v20, v21 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b128 v[24:27], %7\n\ts_waitcnt lgkmcnt(0)\n\t"
#define HWP_OUT "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27"
#define HWP_MOPS : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]) : "v"(c0), "v"(a), "v"(b), "v"(lds_addr) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory"
template <int WK> __device__ __forceinline__ void mfma_then_store(unsigned lds_addr, float c0, float a, float b, float* o) {
    if constexpr (WK == W_NOP7) asm volatile(HWP_INIT HWP_CHAIN4 "s_nop 7\n\t" HWP_READ_DS HWP_OUT HWP_MOPS);
    else if constexpr (WK == W_NOP3x2) asm volatile(HWP_INIT HWP_CHAIN4 "s_nop 3\n\ts_nop 3\n\t" HWP_READ_DS HWP_OUT HWP_MOPS);
    else if constexpr (WK == W_VNOP8) asm volatile(HWP_INIT HWP_CHAIN4 "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t" HWP_READ_DS HWP_OUT HWP_MOPS);
    else if constexpr (WK == W_NOP2) asm volatile(HWP_INIT HWP_CHAIN4 "s_nop 2\n\t" HWP_READ_DS HWP_OUT HWP_MOPS);
    else if constexpr (WK == W_NOP0x3) asm volatile(HWP_INIT HWP_CHAIN4 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\t" HWP_READ_DS HWP_OUT HWP_MOPS);
    else asm volatile(HWP_INIT HWP_CHAIN4 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" HWP_READ_DS HWP_OUT HWP_MOPS);          // the reference: 24 states in three instructions
}
#define HWP_ADD "v_mov_b32 v20, 0\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_add_f32 v20, %1, %2\n\t"
#define HWP_DPP "v_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define HWP_DOPS : "=v"(r) : "v"(a), "v"(b) : "v20"
template <int DK> __device__ __forceinline__ float add_then_dpp(float a, float b) {
    float r;
    if constexpr (DK == D_NOP1) asm volatile(HWP_ADD "s_nop 1\n\t" HWP_DPP HWP_DOPS);
    else if constexpr (DK == D_VNOP2) asm volatile(HWP_ADD "v_nop\n\tv_nop\n\t" HWP_DPP HWP_DOPS);
    else if constexpr (DK == D_NONE) asm volatile(HWP_ADD HWP_DPP HWP_DOPS);
    else if constexpr (DK == D_NOP0) asm volatile(HWP_ADD "s_nop 0\n\t" HWP_DPP HWP_DOPS);
    else if constexpr (DK == D_NOP0x2) asm volatile(HWP_ADD "s_nop 0\n\ts_nop 0\n\t" HWP_DPP HWP_DOPS);
    else asm volatile(HWP_ADD "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t" HWP_DPP HWP_DOPS);
    return r;
}
// bad[0 .. 3] += stale rows 0 .. 3 of the MFMA result behind the wait WK, bad[4] += stale DPP reads behind the wait DK; dynamic LDS: 512 * 64 + 64 bytes
template <int WK, int DK>
__global__ __launch_bounds__(512) void k_snop(const float* A, const float* B, unsigned long long* bad, int iters, int neighbour, const float* gmem) {
    extern __shared__ float lds[];
    volatile int* done = (volatile int*)(lds + 512 * 16);
    if (threadIdx.x == 0) *done = 0;
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x, wave = threadIdx.x >> 6;
    const unsigned addr = (unsigned)(threadIdx.x * 64);
    float a = A[t], b = B[t];
    if (wave < 4) {
        unsigned long long cnt[5] = {0, 0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
            float ref[4], got[4];
            const float c0 = a - b;
            mfma_then_store<99>(addr, c0, a, b, ref);
            mfma_then_store<WK>(addr, c0, a, b, got);
            for (int i = 0; i < 4; i++) cnt[i] += __float_as_uint(ref[i]) != __float_as_uint(got[i]);
            const float dr = add_then_dpp<99>(a, b), dg = add_then_dpp<DK>(a, b);
            cnt[4] += __float_as_uint(dr) != __float_as_uint(dg);
            a = a * 1.0001f + 0.001f; b = b * 0.9999f - 0.002f;
            for (int d = 0; d < ((wave * 7 + it) & 15); d++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(0.0f));
        }
        for (int i = 0; i < 5; i++) if (cnt[i]) atomicAdd(&bad[i], cnt[i]);
        if ((threadIdx.x & 63) == 0) atomicAdd((int*)done, 1);
    } else {
        int sv = 0;
        while (*done < 4) {
            for (int i = 0; i < 8; i++) {
                if (neighbour == N_SLEEP) asm volatile("s_sleep 1\n\ts_sleep 1\n\ts_sleep 1\n\ts_sleep 1");
                else if (neighbour == N_WAKE) asm volatile("s_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3\n\ts_wakeup\n\ts_nop 3");
                else if (neighbour == N_SLOAD) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_load_dword %0, %1, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(sv) : "s"(gmem) : "memory");
                else if (neighbour == N_SLEEPWAKE) asm volatile("s_sleep 1\n\ts_wakeup\n\ts_sleep 1\n\ts_wakeup");
            }
        }
        if (sv == 0x7fffffff) bad[7] = 1;
    }
}

// ---- (3) VALU writes an SGPR (v_cmp, v_readlane, v_readfirstlane) -> VALU reads it (v_cndmask's mask, a scalar operand): gfx940+ asks for TWO wait states, hipcc pads them with
// ONE `s_nop 1` -- which an s_wakeup cuts to one state in the split / rollout kernels (some 500 such sites per step kernel).  k_sgpr<K>: the pair with no wait, `s_nop 0`,
// `s_nop 1` and two v_nop against the same pair six states apart; the SGPR holds a different value (the complementary mask, a marker) before the write.
enum { S_RFL_MOV, S_RL_WRITELANE, S_CMP_CND64, S_CMP_CNDVCC, S_RL_CMP, S_ADD_ADDC, S_CONTROL, S_COUNT };          // S_CONTROL: the reader IN FRONT of the writer (every lane must differ: the comparison can fail)
#define HWP_S_W0 ""
#define HWP_S_W1 "s_nop 0\n\t"
#define HWP_S_WN1 "s_nop 1\n\t"
#define HWP_S_W2 "v_nop\n\tv_nop\n\t"
#define HWP_S_W6 "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t"
#define HWP_V4 "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t"
#define HWP_SOPS : "=&v"(r) : "v"(a), "v"(b) : "s20", "s21", "vcc", "v20"
#define HWP_S_RFL(WS) asm volatile("s_mov_b32 s20, 0x55aa1234\n\t" HWP_V4 "v_readfirstlane_b32 s20, %1\n\t" WS "v_mov_b32 %0, s20" HWP_SOPS)
#define HWP_S_RLW(WS) asm volatile("v_mov_b32 %0, %2\n\ts_mov_b32 s20, 0x55aa1234\n\t" HWP_V4 "v_readlane_b32 s20, %1, 3\n\t" WS "v_writelane_b32 %0, s20, 5" HWP_SOPS)
#define HWP_S_C64(WS) asm volatile("v_cmp_le_f32 s[20:21], %1, %2\n\t" HWP_V4 "v_cmp_gt_f32 s[20:21], %1, %2\n\t" WS "v_cndmask_b32_e64 %0, %1, %2, s[20:21]" HWP_SOPS)
#define HWP_S_CVC(WS) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\t" HWP_V4 "v_cmp_gt_f32 vcc, %1, %2\n\t" WS "v_cndmask_b32_e32 %0, %1, %2, vcc" HWP_SOPS)
#define HWP_S_RLC(WS) asm volatile("s_mov_b32 s20, 0x7fffffff\n\t" HWP_V4 "v_readlane_b32 s20, %1, 3\n\t" WS "v_cmp_le_i32 vcc, s20, %2\n\t" HWP_V4 "v_cndmask_b32_e32 %0, %1, %2, vcc" HWP_SOPS)
#define HWP_S_ADC(WS) asm volatile("v_cmp_le_i32 vcc, 0, %1\n\t" HWP_V4 "v_add_co_u32_e32 v20, vcc, %1, %1\n\t" WS "v_addc_co_u32_e32 %0, vcc, %2, %2, vcc" HWP_SOPS)
#define HWP_S_CTL(WS) asm volatile("s_mov_b32 s20, 0x55aa1234\n\t" HWP_V4 WS "v_mov_b32 %0, s20\n\tv_readfirstlane_b32 s20, %1" HWP_SOPS)
#define HWP_S_BY_WAIT(M) do { if constexpr (W == 0) M(HWP_S_W0); else if constexpr (W == 1) M(HWP_S_W1); else if constexpr (W == 2) M(HWP_S_WN1); else if constexpr (W == 3) M(HWP_S_W2); else M(HWP_S_W6); } while (0)
template <int K, int W> __device__ __forceinline__ float sgpr_pair(float a, float b) {
    float r;
    if constexpr (K == S_RFL_MOV) HWP_S_BY_WAIT(HWP_S_RFL); else if constexpr (K == S_RL_WRITELANE) HWP_S_BY_WAIT(HWP_S_RLW); else if constexpr (K == S_CMP_CND64) HWP_S_BY_WAIT(HWP_S_C64);
    else if constexpr (K == S_CMP_CNDVCC) HWP_S_BY_WAIT(HWP_S_CVC); else if constexpr (K == S_RL_CMP) HWP_S_BY_WAIT(HWP_S_RLC); else if constexpr (K == S_ADD_ADDC) HWP_S_BY_WAIT(HWP_S_ADC);
    else { if constexpr (W == 6) HWP_S_RFL(HWP_S_W6); else HWP_S_BY_WAIT(HWP_S_CTL); }
    return r;
}
// bad[0 .. 3] += lanes that differ from the six-state pair with no wait / `s_nop 0` / `s_nop 1` / two v_nop (256 test lanes per workgroup x iters each)
template <int K>
__global__ __launch_bounds__(512) void k_sgpr(const float* A, const float* B, unsigned long long* bad, int iters, int neighbour) {
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x, wave = threadIdx.x >> 6;
    float a = A[t], b = B[t];
    if (wave < 4) {
        unsigned long long cnt[4] = {0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
            const unsigned ref = __float_as_uint(sgpr_pair<K, 6>(a, b));
            cnt[0] += __float_as_uint(sgpr_pair<K, 0>(a, b)) != ref;
            cnt[1] += __float_as_uint(sgpr_pair<K, 1>(a, b)) != ref;
            cnt[2] += __float_as_uint(sgpr_pair<K, 2>(a, b)) != ref;
            cnt[3] += __float_as_uint(sgpr_pair<K, 3>(a, b)) != ref;
            a = a * 1.0001f + 0.001f; b = b * 0.9999f - 0.002f;
            if ((it & 3) == 3) { const float t2 = a; a = b; b = t2; }          // (both outcomes of the compares)
            for (int d = 0; d < ((wave * 7 + it) & 7); d++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(0.0f));
        }
        for (int i = 0; i < 4; i++) if (cnt[i]) atomicAdd(&bad[i], cnt[i]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&done, 1);
    } else neighbour_loop((volatile int*)&done, neighbour, a + 1.5f, b + 1.5f, bad);
}

// ---- the short form (host side): out[0 .. 2] = stale DPP reads with 0 / 1 / 2 wait states summed over all producer x consumer pairs, alone and beside s_wakeup
// (out[1], out[2] also take the six SGPR write -> read pairs of (3) beside s_wakeup: hipcc's `s_nop 1` in front of them is worth one state there);
// out[3] = stale MFMA-result rows behind a single `s_nop 7` beside s_wakeup, out[4] = behind 8 x v_nop, out[5] = behind `s_nop 3 ; s_nop 3`; out[6] = lane-reads per
// DPP cell, out[7] = DPP cells.  Returns a hipError_t (0 = ok).  ~0.1 s at iters = 64.
struct ProbeBufs { float *A = nullptr, *B = nullptr, *g = nullptr; unsigned long long* bad = nullptr; };
template <int P, int C> inline void probe_dpp_cell(const ProbeBufs& pb, int iters, unsigned long long* out) {
    for (int nb : {(int)NB_NONE, (int)NB_WAKE}) {
        (void)hipMemset(pb.bad, 0, 64);
        hipLaunchKernelGGL((k_dpp<P, C>), dim3(64), dim3(nb == NB_NONE ? 256 : 512), 0, 0, pb.A, pb.B, pb.bad, iters, nb);
        unsigned long long hb[3] = {0, 0, 0};
        (void)hipMemcpy(hb, pb.bad, 24, hipMemcpyDeviceToHost);
        for (int i = 0; i < 3; i++) out[i] += hb[i];
        out[7] += 1;
    }
}
template <int P> inline void probe_dpp_row(const ProbeBufs& pb, int iters, unsigned long long* out) {
    probe_dpp_cell<P, C_MOV_SHR1>(pb, iters, out); probe_dpp_cell<P, C_MAX_NB2>(pb, iters, out); probe_dpp_cell<P, C_FMAC_NB5>(pb, iters, out);
    probe_dpp_cell<P, C_MOV_NB15>(pb, iters, out); probe_dpp_cell<P, C_ADD_SHL4>(pb, iters, out); probe_dpp_cell<P, C_MOV_QUAD>(pb, iters, out);
}
template <int K> inline void probe_sgpr_cell(const ProbeBufs& pb, int iters, unsigned long long* out) {          // beside s_wakeup: `s_nop 0` and `s_nop 1` are both ONE state there
    (void)hipMemset(pb.bad, 0, 64);
    hipLaunchKernelGGL((k_sgpr<K>), dim3(64), dim3(512), 0, 0, pb.A, pb.B, pb.bad, iters, (int)NB_WAKE);
    unsigned long long hb[4] = {0, 0, 0, 0};
    (void)hipMemcpy(hb, pb.bad, 32, hipMemcpyDeviceToHost);
    out[1] += hb[1] + hb[2]; out[2] += hb[3];
    out[7] += 1;
}
template <int WK> inline unsigned long long probe_snop_cell(const ProbeBufs& pb, int iters) {
    (void)hipMemset(pb.bad, 0, 64);
    hipLaunchKernelGGL((k_snop<WK, D_NOP1>), dim3(64), dim3(512), 512 * 64 + 64, 0, pb.A, pb.B, pb.bad, iters, (int)N_WAKE, pb.g);
    unsigned long long hb[4] = {0, 0, 0, 0};
    (void)hipMemcpy(hb, pb.bad, 32, hipMemcpyDeviceToHost);
    return hb[0] + hb[1] + hb[2] + hb[3];
}
inline int probe(int iters, unsigned long long* out) {
    for (int i = 0; i < 8; i++) out[i] = 0;
    const int n = 64 * 512;
    ProbeBufs pb;
    hipError_t e;
    if ((e = hipMalloc((void**)&pb.A, n * 4)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void**)&pb.B, n * 4)) != hipSuccess) { (void)hipFree(pb.A); return (int)e; }
    if ((e = hipMalloc((void**)&pb.g, 4096)) != hipSuccess) { (void)hipFree(pb.A); (void)hipFree(pb.B); return (int)e; }
    if ((e = hipMalloc((void**)&pb.bad, 64)) != hipSuccess) { (void)hipFree(pb.A); (void)hipFree(pb.B); (void)hipFree(pb.g); return (int)e; }
    {
        float* h = new float[n];
        for (int i = 0; i < n; i++) h[i] = (float)(i % 977) / 977.f - 0.5f;
        (void)hipMemcpy(pb.A, h, n * 4, hipMemcpyHostToDevice);
        for (int i = 0; i < n; i++) h[i] = (float)(i % 613) / 613.f + 0.1f;
        (void)hipMemcpy(pb.B, h, n * 4, hipMemcpyHostToDevice);
        delete[] h;
        (void)hipMemset(pb.g, 0, 4096);
    }
    probe_dpp_row<P_ADD>(pb, iters, out); probe_dpp_row<P_FMA>(pb, iters, out); probe_dpp_row<P_MOV>(pb, iters, out); probe_dpp_row<P_MUL>(pb, iters, out);
    probe_dpp_row<P_FMAC_DPP>(pb, iters, out); probe_dpp_row<P_RCP>(pb, iters, out); probe_dpp_row<P_CNDMASK>(pb, iters, out);
    probe_sgpr_cell<S_RFL_MOV>(pb, iters, out); probe_sgpr_cell<S_RL_WRITELANE>(pb, iters, out); probe_sgpr_cell<S_CMP_CND64>(pb, iters, out);
    probe_sgpr_cell<S_CMP_CNDVCC>(pb, iters, out); probe_sgpr_cell<S_RL_CMP>(pb, iters, out); probe_sgpr_cell<S_ADD_ADDC>(pb, iters, out);
    out[6] = 64ull * 256 * (unsigned long long)iters;
    out[3] = probe_snop_cell<W_NOP7>(pb, 4 * iters); out[4] = probe_snop_cell<W_VNOP8>(pb, 4 * iters); out[5] = probe_snop_cell<W_NOP3x2>(pb, 4 * iters);
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipFree(pb.A); (void)hipFree(pb.B); (void)hipFree(pb.g); (void)hipFree(pb.bad);
    return (int)e;
}

}  // namespace hwprobe
}  // namespace dl
