// Does something another wave of the workgroup executes cut an s_nop of THIS wave short?  (tools/ubench/mfma_ds_store.hip: beside waves that loop over
// s_sleep / s_wakeup / s_load_dword a reader 8 wait states behind a v_mfma_f32_4x4x1 -- it needs 3 -- sees the stale accumulator in ~12 % of the cases; with the
// wait split over two s_nop instructions never.)  Here the neighbour's instruction and the form of the wait are varied one at a time:
//   neighbours (waves 4 .. 7 of the workgroup, one per SIMD beside test waves 0 .. 3): nothing | s_sleep 1 | s_wakeup | s_load_dword + s_waitcnt | s_sleep 1 + s_wakeup
//   wait between the last MFMA and the ds_write2_b32 of its rows: s_nop 7 | s_nop 3 ; s_nop 3 | 8 x v_nop | s_nop 2 (the bare need, 3 states) | s_nop 0 ; s_nop 0 ; s_nop 0
//   and the same for a VALU -> DPP hazard (v_add into a register that held 0, wait, v_mov_dpp row_shr:1 of the sum; the ISA asks for 2 wait states): s_nop 1 | 2 x v_nop |
//   s_nop 0 ; s_nop 0 | no wait at all (the control: is the hazard real?) | s_nop 0 (one state: what an s_nop 1 is worth once an s_wakeup ended it)
// The kernels live in drloco_amd/csrc/dl_hwprobe.hpp (the library runs their short form: dl_hw_probe); this file is the long form behind profiles/r05_snop_wakeup.txt.
// build: hipcc --offload-arch=gfx950 -O2 -I drloco_amd/csrc tools/ubench/snop_wakeup.hip -o build_variants/snop_wakeup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "dl_hwprobe.hpp"
using namespace dl::hwprobe;
static const char* n_name[] = {"nothing", "s_sleep 1", "s_wakeup", "s_load_dword", "s_sleep 1 + s_wakeup"};
static const char* w_name[] = {"s_nop 7", "s_nop 3 ; s_nop 3", "8 x v_nop", "s_nop 2", "s_nop 0 x 3"};
static const char* d_name[] = {"s_nop 1", "2 x v_nop", "s_nop 0 x 2", "no wait (control)", "s_nop 0 (1 state)"};
template <int WK, int DK>
static void run(const float* A, const float* B, unsigned long long* bad, const float* gmem, int iters, int neighbour, bool show_dpp) {
    hipMemset(bad, 0, 64);
    hipLaunchKernelGGL((k_snop<WK, DK>), dim3(256), dim3(neighbour == N_NONE ? 256 : 512), 512 * 64 + 64, 0, A, B, bad, iters, neighbour, gmem);
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
