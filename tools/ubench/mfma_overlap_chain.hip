// The CONTEXT in which the round-4 defect of the per-rollout kernel appeared (dl_policy_pair.hpp; the isolated instruction is exact: mfma_overlap.hip):
// a dense chain of v_mfma_f32_4x4x1_16b_f32 over four accumulators (dependent instructions four apart, no wait states), in the middle of which FOUR
// back-to-back instructions RELOCATE their accumulators -- destination != source C, destination component 3 laid over the instruction's own B operand
// (the listing had `v_mfma ... v[144:147], v243, v147, v[198:201]`) -- while a second wave shares the SIMD.
//   test path:  block 1 tied on C[cb] -> block 2 relocating C[cb] -> N[cb] with B in N[cb][3] -> block 3 tied on N[cb], all back to back, fixed registers
//               (operands go in and out through LDS: an asm statement takes at most 30 operands)
//   reference:  the same twelve multiply-adds, every instruction tied, B operands in registers of their own, generous wait states
// Partner modes (the other wave of the SIMD): 0 = the same code (competes for the matrix pipe), 1 = VALU + LDS traffic only, 2 = no partner (one wave per SIMD).
// Output: differing components per accumulator row, per partner mode.  usage: hipcc --offload-arch=gfx950 -O2 mfma_overlap_chain.hip -o /tmp/moc && /tmp/moc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void chain_ref(f4 (&c)[4], const float (&a)[3], const float (&b)[3][4]) {
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) asm volatile("s_nop 7\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\ts_nop 7" : "+v"(c[cb]) : "v"(a[s]), "v"(b[s][cb]));
}

// slots of 512 floats in LDS, one float per thread each: 0..15 C[cb][i] (in, then N[cb][i] out), 16..18 a, 20..23 b[0], 24..27 b[1], 28..31 b[2]
// registers: C[cb] = v[32+4cb : 35+4cb], N[cb] = v[48+4cb : 51+4cb], a = v64..v66, b[0][cb] = v68+cb, b[2][cb] = v72+cb, b[1][cb] = N[cb][3]
__device__ __forceinline__ void chain_test(float* slot0) {
    asm volatile(
        "ds_read_b32 v32, %0 offset:0\n\t"
        "ds_read_b32 v33, %0 offset:2048\n\t"
        "ds_read_b32 v34, %0 offset:4096\n\t"
        "ds_read_b32 v35, %0 offset:6144\n\t"
        "ds_read_b32 v36, %0 offset:8192\n\t"
        "ds_read_b32 v37, %0 offset:10240\n\t"
        "ds_read_b32 v38, %0 offset:12288\n\t"
        "ds_read_b32 v39, %0 offset:14336\n\t"
        "ds_read_b32 v40, %0 offset:16384\n\t"
        "ds_read_b32 v41, %0 offset:18432\n\t"
        "ds_read_b32 v42, %0 offset:20480\n\t"
        "ds_read_b32 v43, %0 offset:22528\n\t"
        "ds_read_b32 v44, %0 offset:24576\n\t"
        "ds_read_b32 v45, %0 offset:26624\n\t"
        "ds_read_b32 v46, %0 offset:28672\n\t"
        "ds_read_b32 v47, %0 offset:30720\n\t"
        "ds_read_b32 v64, %0 offset:32768\n\t"
        "ds_read_b32 v65, %0 offset:34816\n\t"
        "ds_read_b32 v66, %0 offset:36864\n\t"
        "ds_read_b32 v68, %0 offset:40960\n\t"
        "ds_read_b32 v69, %0 offset:43008\n\t"
        "ds_read_b32 v70, %0 offset:45056\n\t"
        "ds_read_b32 v71, %0 offset:47104\n\t"
        "ds_read_b32 v51, %0 offset:49152\n\t"
        "ds_read_b32 v55, %0 offset:51200\n\t"
        "ds_read_b32 v59, %0 offset:53248\n\t"
        "ds_read_b32 v63, %0 offset:55296\n\t"
        "ds_read_b32 v72, %0 offset:57344\n\t"
        "ds_read_b32 v73, %0 offset:59392\n\t"
        "ds_read_b32 v74, %0 offset:61440\n\t"
        "ds_read_b32 v75, %0 offset:63488\n\t"
        "s_waitcnt lgkmcnt(0)\n\ts_nop 4\n\t"
        // block 1: tied
        "v_mfma_f32_4x4x1_16b_f32 v[32:35], v64, v68, v[32:35]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[36:39], v64, v69, v[36:39]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[40:43], v64, v70, v[40:43]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[44:47], v64, v71, v[44:47]\n\t"
        // block 2: four relocations back to back, the destination's last component ON the instruction's own B operand
        "v_mfma_f32_4x4x1_16b_f32 v[48:51], v65, v51, v[32:35]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[52:55], v65, v55, v[36:39]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[56:59], v65, v59, v[40:43]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[60:63], v65, v63, v[44:47]\n\t"
        // block 3: tied on the relocated accumulators
        "v_mfma_f32_4x4x1_16b_f32 v[48:51], v66, v72, v[48:51]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[52:55], v66, v73, v[52:55]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[56:59], v66, v74, v[56:59]\n\t"
        "v_mfma_f32_4x4x1_16b_f32 v[60:63], v66, v75, v[60:63]\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        "ds_write_b32 %0, v48 offset:0\n\t"
        "ds_write_b32 %0, v49 offset:2048\n\t"
        "ds_write_b32 %0, v50 offset:4096\n\t"
        "ds_write_b32 %0, v51 offset:6144\n\t"
        "ds_write_b32 %0, v52 offset:8192\n\t"
        "ds_write_b32 %0, v53 offset:10240\n\t"
        "ds_write_b32 %0, v54 offset:12288\n\t"
        "ds_write_b32 %0, v55 offset:14336\n\t"
        "ds_write_b32 %0, v56 offset:16384\n\t"
        "ds_write_b32 %0, v57 offset:18432\n\t"
        "ds_write_b32 %0, v58 offset:20480\n\t"
        "ds_write_b32 %0, v59 offset:22528\n\t"
        "ds_write_b32 %0, v60 offset:24576\n\t"
        "ds_write_b32 %0, v61 offset:26624\n\t"
        "ds_write_b32 %0, v62 offset:28672\n\t"
        "ds_write_b32 %0, v63 offset:30720\n\t"
        "s_waitcnt lgkmcnt(0)"
        :: "v"((unsigned)(size_t)slot0)
        : "memory", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");
}

__global__ __launch_bounds__(512) void k(const float* A, const float* B, unsigned long long* bad, int iters, int partner) {
    __shared__ float lds[32 * 512];
    const int t = blockIdx.x * 512 + threadIdx.x, wave = threadIdx.x >> 6;
    float x = A[t], y = B[t];
    unsigned long long cnt[4] = {0, 0, 0, 0};
    float* mine = lds + threadIdx.x;
    // waves 0..3 of a workgroup sit on SIMDs 0..3, waves 4..7 are their partners
    if (wave >= 4 && partner == 1) {          // VALU + LDS traffic only (its own slots)
        float s = x;
        for (int it = 0; it < iters * 24; it++) { mine[(it & 31) * 512] = s; s = s * 1.0001f + mine[((it * 7) & 31) * 512] * 1e-6f; }
        if (s == 12345.678f) bad[7] = 1;
        return;
    }
    for (int it = 0; it < iters; it++) {
        float a[3] = {x, x * 0.5f + 0.1f, y - x};
        float b[3][4];
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
            for (int cb = 0; cb < 4; cb++) b[s][cb] = y * (0.3f + 0.1f * cb) + 0.01f * s - x * 0.05f * cb;
        f4 c0[4];
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            c0[cb] = f4{x * 0.5f + cb, y * 0.25f, x + y, x - y * cb};
#pragma unroll
            for (int i = 0; i < 4; i++) mine[(cb * 4 + i) * 512] = c0[cb][i];
        }
#pragma unroll
        for (int s = 0; s < 3; s++) {
            mine[(16 + s) * 512] = a[s];
#pragma unroll
            for (int cb = 0; cb < 4; cb++) mine[(20 + 4 * s + cb) * 512] = b[s][cb];
        }
        chain_ref(c0, a, b);
        chain_test(mine);
#pragma unroll
        for (int cb = 0; cb < 4; cb++)
#pragma unroll
            for (int i = 0; i < 4; i++) cnt[i] += __float_as_uint(c0[cb][i]) != __float_as_uint(mine[(cb * 4 + i) * 512]);
        x = x * 1.0001f + 0.001f; y = y * 0.9999f - 0.002f;
    }
    for (int i = 0; i < 4; i++) if (cnt[i]) atomicAdd(&bad[i], cnt[i]);
}

int main() {
    const int wg = 256 * 2, n = wg * 512;
    float *A, *B; unsigned long long* bad;
    hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&bad, 64);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 977) / 977.f - 0.5f;
    hipMemcpy(A, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 613) / 613.f + 0.1f;
    hipMemcpy(B, h, n * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"partner wave runs the same chains", "partner wave runs VALU + LDS traffic", "no partner (256 threads per workgroup)"};
    for (int mode = 0; mode < 3; mode++) {
        hipMemset(bad, 0, 64);
        const int iters = 1500, threads = mode == 2 ? 256 : 512;
        hipLaunchKernelGGL(k, dim3(wg), dim3(threads), 0, 0, A, B, bad, iters, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned long long hb[8]; hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost);
        const long long lanes = (long long)wg * (mode == 0 ? 512 : 256);
        printf("%-40s: differing components per accumulator row 0..3 in %lld lane-chains x 4 accumulators: %llu %llu %llu %llu\n", names[mode], lanes * iters, hb[0], hb[1], hb[2], hb[3]);
    }
    return 0;
}
