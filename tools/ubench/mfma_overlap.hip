// Is a v_mfma_f32_4x4x1_16B_f32 whose DESTINATION overlaps its source B register safe?  (LLVM allows the overlap for the 4x4 shapes; the rollout kernel's
// register allocator produced it under pressure.)  Every lane runs the instruction twice on the same operands -- destination apart from the sources, and
// destination component 3 ON the source B register -- many times, with two waves per SIMD competing for the matrix core, and counts differing components.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const float* A, const float* B, unsigned long long* bad, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cnt[4] = {0, 0, 0, 0};
    float a = A[t], b = B[t];
    for (int it = 0; it < iters; it++) {
        f4 c = {a * 0.5f, b * 0.25f, a + b, a - b};
        f4 ref = c, ovl;
        asm volatile("s_nop 4\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+v"(ref) : "v"(a), "v"(b));
        float o0, o1, o2, o3;
        asm volatile("v_mov_b32 v24, %4\n\tv_mov_b32 v25, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %7\n\tv_mov_b32 v31, %9\n\ts_nop 4\n\t"
                     "v_mfma_f32_4x4x1_16b_f32 v[28:31], %8, v31, v[24:27]\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mov_b32 %0, v28\n\tv_mov_b32 %1, v29\n\tv_mov_b32 %2, v30\n\tv_mov_b32 %3, v31"
                     : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b) : "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        ovl = f4{o0, o1, o2, o3};
        for (int i = 0; i < 4; i++) cnt[i] += __float_as_uint(ref[i]) != __float_as_uint(ovl[i]);
        a = a * 1.0001f + 0.001f; b = b * 0.9999f - 0.002f;
    }
    for (int i = 0; i < 4; i++) if (cnt[i]) atomicAdd(&bad[i], cnt[i]);
}
int main() {
    const int n = 256 * 2 * 512;          // two workgroups of eight waves per CU: two waves per SIMD
    float *A, *B; unsigned long long* bad;
    hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&bad, 32); hipMemset(bad, 0, 32);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 977) / 977.f - 0.5f;
    hipMemcpy(A, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 613) / 613.f + 0.1f;
    hipMemcpy(B, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 512), dim3(512), 0, 0, A, B, bad, 2000);
    unsigned long long hb[4]; hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
    printf("destination component 3 on source B: differing results per row 0..3 of %lld lane-instructions: %llu %llu %llu %llu\n", (long long)n * 2000, hb[0], hb[1], hb[2], hb[3]);
    return 0;
}
