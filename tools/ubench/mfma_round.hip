// Does v_mfma_f32_16x16x4_f32 round like four sequential fmas (k ascending)?  And like four v_mfma_f32_4x4x1_16B_f32 instructions?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D16, float* D4) {
    const int l = threadIdx.x, lm = l & 15, lk = l >> 4;
    // 16x16x4: lane (lk, lm) supplies A[row lm][k = lk], B[k = lk][col lm]; acc[i] = D[row 4 lk + i][col lm]
    f4 acc; for (int i = 0; i < 4; i++) acc[i] = C[(4 * lk + i) * 16 + lm];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[lm * 4 + lk], B[lk * 16 + lm], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) D16[(4 * lk + i) * 16 + lm] = acc[i];
    // 4x4x1, 16 blocks: lane l: block b = l / 4, j = l % 4: supplies A_b[i = j][0], B_b[0][j]; acc[i] = D_b[i][j].  Use blocks = 16 column groups of 4 over rows 0..3 only:
    // D[row i][col 4 b + j] = C + sum_k A[row i][k] B[k][col 4 b + j], one instruction per k
    f4 a2; for (int i = 0; i < 4; i++) a2[i] = (l < 16) ? C[i * 16 + l] : 0.f;        // only 16 columns exist: lanes 0..15 (blocks 0..3)
    for (int kk = 0; kk < 4; kk++) {
        const float av = A[(l & 3) * 4 + kk];                 // A[row = l % 4][k]
        const float bv = (l < 16) ? B[kk * 16 + l] : 0.f;     // B[k][col = l]
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, a2, 0, 0, 0);
    }
    if (l < 16) for (int i = 0; i < 4; i++) D4[i * 16 + l] = a2[i];
}
int main() {
    float hA[64], hB[64], hC[256], hD16[256], hD4[64];
    srand(7);
    long same_seq = 0, same_rev = 0, same_44 = 0, total = 0, total44 = 0;
    float *A, *B, *C, *D16, *D4;
    hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&C, 1024); hipMalloc(&D16, 1024); hipMalloc(&D4, 256);
    for (int rep = 0; rep < 2000; rep++) {
        for (int i = 0; i < 64; i++) { hA[i] = (float)rand() / RAND_MAX * 2 - 1; hB[i] = (float)rand() / RAND_MAX * 2 - 1; }
        for (int i = 0; i < 256; i++) hC[i] = ((float)rand() / RAND_MAX * 2 - 1) * (rep % 3 == 0 ? 1e-3f : 1.f);
        hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice); hipMemcpy(C, hC, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C, D16, D4);
        hipMemcpy(hD16, D16, 1024, hipMemcpyDeviceToHost); hipMemcpy(hD4, D4, 256, hipMemcpyDeviceToHost);
        for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
            float s = hC[r * 16 + c]; for (int kk = 0; kk < 4; kk++) s = fmaf(hA[r * 4 + kk], hB[kk * 16 + c], s);
            float t = hC[r * 16 + c]; for (int kk = 3; kk >= 0; kk--) t = fmaf(hA[r * 4 + kk], hB[kk * 16 + c], t);
            total++; same_seq += memcmp(&s, &hD16[r * 16 + c], 4) == 0; same_rev += memcmp(&t, &hD16[r * 16 + c], 4) == 0;
            if (r < 4) { total44++; same_44 += memcmp(&hD4[r * 16 + c], &hD16[r * 16 + c], 4) == 0; }
        }
    }
    printf("16x16x4 == sequential fma chain (k ascending): %ld of %ld; == descending: %ld; four 4x4x1 == one 16x16x4: %ld of %ld\n", same_seq, total, same_rev, same_44, total44);
    return 0;
}
