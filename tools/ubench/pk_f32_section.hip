// Would PACKED float32 (v_pk_fma_f32 over TWO walkers per 16-lane row, operands in register pairs) lower the instruction count per walker of the step kernel's hot
// sections?  (the benchmark line is flat in the walker count because both waves of a SIMD issue VALU instructions 72 % of the time.)
// The sections' inner patterns, timed per wave with s_memtime over many repetitions, at one and at two waves per SIMD:
//   A  the triangular substitution / factorisation step  x_j += l_jk * x_k (lane k broadcast):  scalar form = ONE v_fmac_f32_dpp row_newbcast (the product's fused form,
//      a dependent chain of 14 steps); packed form = v_mov_b64_dpp row_newbcast (DPP does not exist on VOP3P: the broadcast is its own instruction) + v_pk_fma_f32
//   B  the same with four INDEPENDENT chains interleaved (issue rate instead of latency)
//   C  lane-local 3-vector arithmetic (cross product + dot + axpy, 21 multiply-adds: the kinematics / RNE pattern): scalar v_fma_f32 vs v_pk_fma_f32 over two walkers
// Output: cycles per repetition and per WALKER-repetition (a scalar wave carries 4 walkers, a packed one 8).  usage: hipcc --offload-arch=gfx950 -O2 pk_f32_section.hip -o /tmp/pk && /tmp/pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

#define STEP_S(K) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(l));
#define STEP_P(K) asm volatile("v_mov_b64_dpp %1, %0 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_pk_fma_f32 %0, %1, %2, %0" : "+v"(x), "=&v"(t) : "v"(l));
#define CHAIN14(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13)

__global__ void k_chain_scalar(float* out, long long* cyc, int reps) {
    float x = threadIdx.x * 1e-3f + 1.0f, l = 1e-4f * (threadIdx.x & 15);
    asm volatile("s_nop 1" : "+v"(x));
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) { CHAIN14(STEP_S) }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
__global__ void k_chain_packed(float* out, long long* cyc, int reps) {
    f2 x = {threadIdx.x * 1e-3f + 1.0f, threadIdx.x * 2e-3f + 1.0f}, l = {1e-4f * (threadIdx.x & 15), 2e-4f * (threadIdx.x & 15)}, t;
    asm volatile("s_nop 1" : "+v"(x));
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) { CHAIN14(STEP_P) }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// four independent chains
#define STEP4_S(K) asm volatile("v_fmac_f32_dpp %0, %0, %4 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %1, %4 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
                                "v_fmac_f32_dpp %2, %2, %4 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %3, %4 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" \
                                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(l));
#define STEP4_P(K) asm volatile("v_mov_b64_dpp %4, %0 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %1 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
                                "v_mov_b64_dpp %6, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %3 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t" \
                                "v_pk_fma_f32 %0, %4, %8, %0\n\tv_pk_fma_f32 %1, %5, %8, %1\n\tv_pk_fma_f32 %2, %6, %8, %2\n\tv_pk_fma_f32 %3, %7, %8, %3" \
                                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"(l));
__global__ void k_chain4_scalar(float* out, long long* cyc, int reps) {
    float x0 = threadIdx.x * 1e-3f + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, l = 1e-4f * (threadIdx.x & 15);
    asm volatile("s_nop 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) { CHAIN14(STEP4_S) }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
__global__ void k_chain4_packed(float* out, long long* cyc, int reps) {
    f2 x0 = {threadIdx.x * 1e-3f + 1.0f, 1.5f}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, l = {1e-4f * (threadIdx.x & 15), 2e-4f}, t0_, t1_, t2_, t3_;
    asm volatile("s_nop 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) { CHAIN14(STEP4_P) }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0[0] + x1[1] + x2[0] + x3[1];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// lane-local vector arithmetic: c = a x b; d = c . a; a += d * b  (21 multiply-adds), the compiler's own scalar code vs v_pk_fma_f32 over (walker 0, walker 1) pairs
__global__ void k_vec_scalar(float* out, long long* cyc, int reps) {
    float ax = threadIdx.x * 1e-3f, ay = 0.5f, az = 0.25f, bx = 0.1f, by = 0.2f + ax, bz = 0.3f;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const float d = cx * ax + cy * ay + cz * az + 1e-3f;
        ax = fmaf(d, bx, ax); ay = fmaf(d, by, ay); az = fmaf(d, bz, az);
        bx = fmaf(cx, 1e-3f, bx); by = fmaf(cy, 1e-3f, by); bz = fmaf(cz, 1e-3f, bz);
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz));
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ax + ay + az + bx + by + bz;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { f2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { f2 d; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__global__ void k_vec_packed(float* out, long long* cyc, int reps) {
    f2 ax = {threadIdx.x * 1e-3f, threadIdx.x * 2e-3f}, ay = {0.5f, 0.6f}, az = {0.25f, 0.35f}, bx = {0.1f, 0.15f}, by = ax + 0.2f, bz = {0.3f, 0.31f};
    const f2 eps = {1e-3f, 1e-3f};
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        const f2 cx = pk_fma(ay, bz, -pk_mul(az, by)), cy = pk_fma(az, bx, -pk_mul(ax, bz)), cz = pk_fma(ax, by, -pk_mul(ay, bx));
        const f2 d = pk_fma(cx, ax, pk_fma(cy, ay, pk_fma(cz, az, eps)));
        ax = pk_fma(d, bx, ax); ay = pk_fma(d, by, ay); az = pk_fma(d, bz, az);
        bx = pk_fma(cx, eps, bx); by = pk_fma(cy, eps, by); bz = pk_fma(cz, eps, bz);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ax[0] + ay[1] + az[0] + bx[1] + by[0] + bz[1];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <typename K> static double run(K kern, int threads_per_wg, int reps, float* out, long long* cyc) {
    const int wgs = 256;          // one workgroup per CU: 256 threads = one wave per SIMD, 512 = two
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads_per_wg), 0, 0, out, cyc, reps);          // warm-up
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads_per_wg), 0, 0, out, cyc, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    const int nw = wgs * threads_per_wg / 64;
    long long* h = (long long*)malloc(nw * 8);
    (void)hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nw; i++) s += (double)h[i];
    free(h);
    return s / nw / reps;
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int reps = 20000;
    printf("cycles (s_memtime ticks) per repetition and wave; per walker = / 4 (scalar: one walker per 16-lane row) or / 8 (packed: two)\n");
    for (int tpw : {256, 512}) {
        const char* occ = tpw == 256 ? "1 wave / SIMD " : "2 waves / SIMD";
        const double a0 = run(k_chain_scalar, tpw, reps, out, cyc), a1 = run(k_chain_packed, tpw, reps, out, cyc);
        const double b0 = run(k_chain4_scalar, tpw, reps, out, cyc), b1 = run(k_chain4_packed, tpw, reps, out, cyc);
        const double c0 = run(k_vec_scalar, tpw, reps, out, cyc), c1 = run(k_vec_packed, tpw, reps, out, cyc);
        printf("%s  A dependent 14-step substitution chain:      scalar %7.1f  packed %7.1f   per walker %6.2f vs %6.2f  -> packed is %.2f x per walker\n", occ, a0, a1, a0 / 4, a1 / 8, (a0 / 4) / (a1 / 8));
        printf("%s  B four independent chains (56 steps):         scalar %7.1f  packed %7.1f   per walker %6.2f vs %6.2f  -> packed is %.2f x per walker\n", occ, b0, b1, b0 / 4, b1 / 8, (b0 / 4) / (b1 / 8));
        printf("%s  C lane-local vector arithmetic (21 fma):      scalar %7.1f  packed %7.1f   per walker %6.2f vs %6.2f  -> packed is %.2f x per walker\n", occ, c0, c1, c0 / 4, c1 / 8, (c0 / 4) / (c1 / 8));
    }
    return 0;
}
