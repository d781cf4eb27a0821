// How many wait states does gfx950 need between a VALU write of a VGPR and its read through DPP?  The ISA manual (and hipcc's hazard recogniser) say two.
// The split workgroups cannot have two from an `s_nop 1` anyway -- a partner's s_wakeup ends an s_nop after one state (snop_wakeup.hip) -- so the question decides
// between `s_nop 0` (one state, which nothing can shorten) and two `v_nop` in the hand-written DPP statements of dl_group.hpp (DL_DPP_WAIT).
// Every producer x consumer pair the kernels contain one state apart (tools/check_dpp_hazards.py on a -DDL_DPP_WAIT=1 listing: v_fma_f32 / v_mov_b32 ->
// v_max_f32_dpp / v_fmac_f32_dpp row_newbcast) and the neighbouring forms, with 0 (control), 1 and 2 states, alone and beside a wave that loops over s_wakeup,
// VALU + DPP work or MFMAs.  The register is overwritten with a marker first, so a stale read differs from a fresh one.
// usage: dpp_wait [iters]
// The kernels live in drloco_amd/csrc/dl_hwprobe.hpp (the library runs their short form: dl_hw_probe -- the guard of the one-wait-state build and a -m gpu test);
// this file is the long form behind profiles/r05_dpp_wait.txt.   build: hipcc --offload-arch=gfx950 -O2 -I drloco_amd/csrc tools/ubench/dpp_wait.hip -o build_variants/dpp_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "dl_hwprobe.hpp"
using namespace dl::hwprobe;
static const char* p_name[] = {"v_add_f32", "v_fma_f32", "v_mov_b32", "v_mul_f32", "v_fmac_f32_dpp (a link of a chain)", "v_rcp_f32", "v_cndmask_b32"};
static const char* c_name[] = {"v_mov_b32_dpp row_shr:1", "v_max_f32_dpp row_newbcast:2", "v_fmac_f32_dpp row_newbcast:5", "v_mov_b32_dpp row_newbcast:15", "v_add_f32_dpp row_shl:4", "v_mov_b32_dpp quad_perm:[1,0,3,2]"};
template <int P, int C>
static void run(const float* A, const float* B, unsigned long long* bad, int iters, int neighbour, unsigned long long* tot) {
    hipMemset(bad, 0, 64);
    hipLaunchKernelGGL((k_dpp<P, C>), dim3(256), dim3(neighbour == 0 ? 256 : 512), 0, 0, A, B, bad, iters, neighbour);
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
