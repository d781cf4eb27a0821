// Does something another wave of the workgroup executes cut an s_nop of THIS wave short?  (tools/ubench/mfma_ds_store.hip: beside waves that loop over
// s_sleep / s_wakeup / s_load_dword a reader 8 wait states behind a v_mfma_f32_4x4x1 -- it needs 3 -- sees the stale accumulator in ~12 % of the cases; with the
// wait split over two s_nop instructions never.)  Here the neighbour's instruction and the form of the wait are varied one at a time:
//   neighbours (waves 4 .. 7 of the workgroup, one per SIMD beside test waves 0 .. 3): nothing | s_sleep 1 | s_wakeup | s_load_dword + s_waitcnt | s_sleep 1 + s_wakeup
//   wait between the last MFMA and the ds_write2_b32 of its rows: s_nop 7 | s_nop 3 ; s_nop 3 | 8 x v_nop | s_nop 2 (the bare need, 3 states) | s_nop 0 ; s_nop 0 ; s_nop 0
//   and the same for a VALU -> DPP hazard (v_add into a register that held 0, wait, v_mov_dpp row_shr:1 of the sum; the ISA asks for 2 wait states): s_nop 1 | 2 x v_nop |
//   s_nop 0 ; s_nop 0 | no wait at all (the control: is the hazard real?) | s_nop 0 (one state: what an s_nop 1 is worth once an s_wakeup ended it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
enum { N_NONE, N_SLEEP, N_WAKE, N_SLOAD, N_SLEEPWAKE, N_COUNT };
static const char* n_name[] = {"nothing", "s_sleep 1", "s_wakeup", "s_load_dword", "s_sleep 1 + s_wakeup"};
enum { W_NOP7, W_NOP3x2, W_VNOP8, W_NOP2, W_NOP0x3, W_COUNT };
static const char* w_name[] = {"s_nop 7", "s_nop 3 ; s_nop 3", "8 x v_nop", "s_nop 2", "s_nop 0 x 3"};
enum { D_NOP1, D_VNOP2, D_NOP0x2, D_NONE, D_NOP0, D_COUNT };
static const char* d_name[] = {"s_nop 1", "2 x v_nop", "s_nop 0 x 2", "no wait (control)", "s_nop 0 (1 state)"};
#define M4 "v_mfma_f32_4x4x1_16b_f32 v[20:23], %5, %6, v[20:23]\n\t"
#define CHAIN4 M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4
#define INIT "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %4\n\ts_nop 4\n\t"
#define READ_DS "ds_write2_b32 %7, v22, v23 offset0:2 offset1:3\n\tds_write2_b32 %7, v20, v21 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b128 v[24:27], %7\n\ts_waitcnt lgkmcnt(0)\n\t"
#define OUT "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27"
#define OPS : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]) : "v"(c0), "v"(a), "v"(b), "v"(lds_addr) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory"
template <int WK> __device__ __forceinline__ void mfma_then_store(unsigned lds_addr, float c0, float a, float b, float* o) {
    if constexpr (WK == W_NOP7) asm volatile(INIT CHAIN4 "s_nop 7\n\t" READ_DS OUT OPS);
    else if constexpr (WK == W_NOP3x2) asm volatile(INIT CHAIN4 "s_nop 3\n\ts_nop 3\n\t" READ_DS OUT OPS);
    else if constexpr (WK == W_VNOP8) asm volatile(INIT CHAIN4 "v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\t" READ_DS OUT OPS);
    else if constexpr (WK == W_NOP2) asm volatile(INIT CHAIN4 "s_nop 2\n\t" READ_DS OUT OPS);
    else if constexpr (WK == W_NOP0x3) asm volatile(INIT CHAIN4 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\t" READ_DS OUT OPS);
    else asm volatile(INIT CHAIN4 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" READ_DS OUT OPS);          // the reference: 24 states in three instructions
}
template <int DK> __device__ __forceinline__ float add_then_dpp(float a, float b) {
    float r;
    if constexpr (DK == D_NOP1) asm volatile("v_mov_b32 v20, 0\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_add_f32 v20, %1, %2\n\ts_nop 1\n\tv_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b) : "v20");
    else if constexpr (DK == D_VNOP2) asm volatile("v_mov_b32 v20, 0\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_add_f32 v20, %1, %2\n\tv_nop\n\tv_nop\n\tv_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b) : "v20");
    else if constexpr (DK == D_NONE) asm volatile("v_mov_b32 v20, 0\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_add_f32 v20, %1, %2\n\tv_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b) : "v20");
    else if constexpr (DK == D_NOP0) asm volatile("v_mov_b32 v20, 0\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_add_f32 v20, %1, %2\n\ts_nop 0\n\tv_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b) : "v20");
    else if constexpr (DK == D_NOP0x2) asm volatile("v_mov_b32 v20, 0\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_add_f32 v20, %1, %2\n\ts_nop 0\n\ts_nop 0\n\tv_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b) : "v20");
    else asm volatile("v_mov_b32 v20, 0\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_add_f32 v20, %1, %2\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_mov_b32_dpp %0, v20 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(a), "v"(b) : "v20");
    return r;
}
template <int WK, int DK>
__global__ __launch_bounds__(512) void k(const float* A, const float* B, unsigned long long* bad, int iters, int neighbour, const float* gmem) {
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
template <int WK, int DK>
static void run(const float* A, const float* B, unsigned long long* bad, const float* gmem, int iters, int neighbour, bool show_dpp) {
    hipMemset(bad, 0, 64);
    hipLaunchKernelGGL((k<WK, DK>), dim3(256), dim3(neighbour == N_NONE ? 256 : 512), 512 * 64 + 64, 0, A, B, bad, iters, neighbour, gmem);
    unsigned long long hb[5]; hipMemcpy(hb, bad, 40, hipMemcpyDeviceToHost);
    if (show_dpp) printf("   %-20s %12llu", d_name[DK], hb[4]); else printf("   %-20s %llu/%llu/%llu/%llu", w_name[WK], hb[0], hb[1], hb[2], hb[3]);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int n = 256 * 512;
    float *A, *B, *gmem; unsigned long long* bad;
    hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&bad, 64); hipMalloc(&gmem, 4096); hipMemset(gmem, 0, 4096);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 977) / 977.f - 0.5f;
    hipMemcpy(A, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 613) / 613.f + 0.1f;
    hipMemcpy(B, h, n * 4, hipMemcpyHostToDevice);
    printf("stale results of %lld lane-reads; test waves 0..3 of a workgroup, the neighbour's loop on waves 4..7\n", 256LL * 256 * iters);
    for (int nb = 0; nb < N_COUNT; nb++) {
        printf("beside %s\n  v_mfma_f32_4x4x1 chain -> wait -> ds_write2_b32 (rows 0/1/2/3 stale):\n", n_name[nb]);
        run<W_NOP7, D_NOP1>(A, B, bad, gmem, iters, nb, false); run<W_NOP3x2, D_NOP1>(A, B, bad, gmem, iters, nb, false); run<W_VNOP8, D_NOP1>(A, B, bad, gmem, iters, nb, false); printf("\n");
        run<W_NOP2, D_NOP1>(A, B, bad, gmem, iters, nb, false); run<W_NOP0x3, D_NOP1>(A, B, bad, gmem, iters, nb, false); printf("\n");
        printf("  v_add_f32 -> wait -> v_mov_b32_dpp row_shr:1 (stale lanes):\n");
        run<W_NOP7, D_NOP1>(A, B, bad, gmem, iters, nb, true); run<W_NOP7, D_VNOP2>(A, B, bad, gmem, iters, nb, true); run<W_NOP7, D_NOP0x2>(A, B, bad, gmem, iters, nb, true); run<W_NOP7, D_NONE>(A, B, bad, gmem, iters, nb, true); run<W_NOP7, D_NOP0>(A, B, bad, gmem, iters, nb, true); printf("\n");
    }
    return 0;
}
