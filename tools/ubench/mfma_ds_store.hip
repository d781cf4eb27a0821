// How many wait states does a reader of an MFMA result need on gfx950, and what beside the wave can break a wait that is long enough?
// hipcc (ROCm 7.2) puts `s_nop 3` (4 wait states) between a v_mfma_f32_4x4x1_16B_f32 and a ds_write2_b32 whose data operands are its destination.  The
// round-4 per-rollout policy kernel, written with builtins, stored a stale row 2 of the heads' partial sums about 6.5e-4 of the time with exactly that
// sequence (row 2 = the first data register the store reads; EXPERIMENTS.md, round 5, "the 4x4x1 defect, found").  This test reproduces the signature outside
// the kernel:
//   the waves 0..3 of a workgroup (one per SIMD) run a chain of dependent MFMAs on one accumulator, read the result after WAIT wait states -- an LDS store
//   of rows 2,3 then 0,1 (the kernel's sequence) or four v_mov -- and compare with the same chain settled for 24 wait states;
//   the other waves of the workgroup (1 or 3 more per SIMD) run one of thirteen instruction streams beside them until the test waves are done.
// Result (profiles/r05_mfma_ds_store.txt): the need is 3 (LDS store) / 4 (VALU) states behind a 4x4x1 and 9 / 10 behind a 16x16x4, and NOTHING beside the wave
// moves it -- except the stream with s_wakeup in it: there a wait of ONE s_nop instruction fails whatever its count (W <= 8), a wait of two instructions
// (W = 12: s_nop 7 + s_nop 3) holds for the 4x4x1 and fails for the 16x16x4 (W = 14: s_nop 7 + s_nop 5 is worth 1 + 6 when the first is ended).
// tools/ubench/snop_wakeup.hip isolates it: s_wakeup, alone, ends the s_nop another wave of the workgroup is in.
// usage: mfma_ds_store [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
enum { O_NONE, O_SAME, O_VADD, O_TRANS, O_F64, O_DPP, O_LDS, O_MFMA16, O_MFMA4, O_MIX, O_LANE, O_F64T, O_SLEEP, O_VMEM, O_COUNT };
static const char* o_name[] = {"alone", "the same chain", "v_add_f32", "v_rcp/v_sqrt_f32", "v_fma_f64", "v_add_f32 dpp", "ds_read_b128", "mfma 16x16x4", "mfma 4x4x1 x4 accs", "dpp+trans+lds+pk mix", "v_readlane/v_writelane", "f64 rcp/rsq/div/cvt", "s_sleep/s_wakeup/s_load", "global load/store/atomic"};
#define M4 "v_mfma_f32_4x4x1_16b_f32 v[20:23], %5, %6, v[20:23]\n\t"
#define M16 "v_mfma_f32_16x16x4_f32 v[20:23], %5, %6, v[20:23]\n\t"
#define CHAIN4 M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4 "s_nop 1\n\t" M4
#define CHAIN16 M16 M16 M16 M16
#define CHAIN4L CHAIN4 "s_nop 1\n\t" CHAIN4 "s_nop 1\n\t" CHAIN4 "s_nop 1\n\t" CHAIN4 "s_nop 1\n\t" CHAIN4 "s_nop 1\n\t" CHAIN4 "s_nop 1\n\t" CHAIN4 "s_nop 1\n\t" CHAIN4          // 64 links, the length of the kernel's heads chain
#define INIT "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %4\n\ts_nop 4\n\t"
#define READ_DS "ds_write2_b32 %7, v22, v23 offset0:2 offset1:3\n\tds_write2_b32 %7, v20, v21 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\t" \
                "ds_read_b128 v[24:27], %7\n\ts_waitcnt lgkmcnt(0)\n\t"
#define READ_VALU "v_mov_b32 v26, v22\n\tv_mov_b32 v27, v23\n\tv_mov_b32 v24, v20\n\tv_mov_b32 v25, v21\n\ts_nop 7\n\ts_nop 7\n\t"
#define OUT "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27"
#define OPS : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]) : "v"(c0), "v"(a), "v"(b), "v"(lds_addr), "n"(W - 1) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory"
// W wait states between the last MFMA and the reader: W <= 8: s_nop W-1; above: s_nop 7 + s_nop W-9
template <int SHAPE, int READER, int W>
__device__ __forceinline__ void chain_then_read(unsigned lds_addr, float c0, float a, float b, float* o) {
    if constexpr (W <= 8) {
        if constexpr (SHAPE == 4 && READER == 0) asm volatile(INIT CHAIN4 "s_nop %8\n\t" READ_DS OUT OPS);
        if constexpr (SHAPE == 5 && READER == 0) asm volatile(INIT CHAIN4L "s_nop %8\n\t" READ_DS OUT OPS);
        if constexpr (SHAPE == 4 && READER == 1) asm volatile(INIT CHAIN4 "s_nop %8\n\t" READ_VALU OUT OPS);
        if constexpr (SHAPE == 16 && READER == 0) asm volatile(INIT CHAIN16 "s_nop %8\n\t" READ_DS OUT OPS);
        if constexpr (SHAPE == 16 && READER == 1) asm volatile(INIT CHAIN16 "s_nop %8\n\t" READ_VALU OUT OPS);
    } else {
        constexpr int W2 = W - 8;
#undef OPS
#define OPS : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]) : "v"(c0), "v"(a), "v"(b), "v"(lds_addr), "n"(W2 > 8 ? 7 : W2 - 1), "n"(W2 > 8 ? W2 - 9 : 0), "n"(W2 > 8 ? 1 : 0) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory"
        // s_nop 7; s_nop (W2<=8 ? W2-1 : 7); and a third s_nop for W2 > 8 -- the third is always emitted, as `s_nop 0` it costs one state: accounted below
        static_assert(W2 >= 1 && W2 <= 16, "");
        if constexpr (W2 <= 8) {
            if constexpr (SHAPE == 4 && READER == 0) asm volatile(INIT CHAIN4 "s_nop 7\n\ts_nop %8\n\t" READ_DS OUT OPS);
            if constexpr (SHAPE == 5 && READER == 0) asm volatile(INIT CHAIN4L "s_nop 7\n\ts_nop %8\n\t" READ_DS OUT OPS);
            if constexpr (SHAPE == 4 && READER == 1) asm volatile(INIT CHAIN4 "s_nop 7\n\ts_nop %8\n\t" READ_VALU OUT OPS);
            if constexpr (SHAPE == 16 && READER == 0) asm volatile(INIT CHAIN16 "s_nop 7\n\ts_nop %8\n\t" READ_DS OUT OPS);
            if constexpr (SHAPE == 16 && READER == 1) asm volatile(INIT CHAIN16 "s_nop 7\n\ts_nop %8\n\t" READ_VALU OUT OPS);
        } else {
            if constexpr (SHAPE == 4 && READER == 0) asm volatile(INIT CHAIN4 "s_nop 7\n\ts_nop 7\n\ts_nop %9\n\t" READ_DS OUT OPS);
            if constexpr (SHAPE == 5 && READER == 0) asm volatile(INIT CHAIN4L "s_nop 7\n\ts_nop 7\n\ts_nop %9\n\t" READ_DS OUT OPS);
            if constexpr (SHAPE == 4 && READER == 1) asm volatile(INIT CHAIN4 "s_nop 7\n\ts_nop 7\n\ts_nop %9\n\t" READ_VALU OUT OPS);
            if constexpr (SHAPE == 16 && READER == 0) asm volatile(INIT CHAIN16 "s_nop 7\n\ts_nop 7\n\ts_nop %9\n\t" READ_DS OUT OPS);
            if constexpr (SHAPE == 16 && READER == 1) asm volatile(INIT CHAIN16 "s_nop 7\n\ts_nop 7\n\ts_nop %9\n\t" READ_VALU OUT OPS);
        }
    }
}
__device__ __forceinline__ void other_stream(int other, float& x, float& y, double& d, unsigned lds_addr, float* gmem) {
    // about a hundred instructions of one kind per call
    typedef float f4 __attribute__((ext_vector_type(4)));
    switch (other) {
    case O_VADD: for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %0\n\tv_mul_f32 %0, %0, %1\n\tv_fma_f32 %1, %0, %1, %0\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0" : "+v"(x), "+v"(y)); break;
    case O_TRANS: for (int i = 0; i < 16; i++) asm volatile("v_rcp_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_rcp_f32 %1, %1\n\tv_exp_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_log_f32 %0, %0" : "+v"(x), "+v"(y)); break;
    case O_F64: for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %0, %0\n\tv_mul_f64 %0, %0, %0\n\tv_add_f64 %0, %0, %0\n\tv_fma_f64 %0, %0, %0, %0" : "+v"(d)); break;
    case O_DPP: for (int i = 0; i < 16; i++) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %1, %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(x), "+v"(y)); break;
    case O_LDS: for (int i = 0; i < 16; i++) { f4 t; asm volatile("ds_read_b128 %0, %1\n\tds_read_b128 %0, %1 offset:16\n\tds_write_b32 %1, %2 offset:32\n\tds_read_b128 %0, %1 offset:48\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(lds_addr), "v"(x) : "memory"); x += t[0]; } break;
    case O_MFMA16: { f4 a0 = {x, y, x, y}, a1 = a0; for (int i = 0; i < 16; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %2, %1\n\tv_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %2, %1" : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));
                     asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1)); x = a0[0] * 1e-30f + 0.5f; y = a1[1] * 1e-30f + 0.25f; } break;
    case O_MFMA4: { f4 a0 = {x, y, x, y}, a1 = a0, a2 = a0, a3 = a0; for (int i = 0; i < 16; i++) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %5, %4, %1\n\tv_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\n\tv_mfma_f32_4x4x1_16b_f32 %3, %5, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
                    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); x = (a0[0] + a2[2]) * 1e-30f + 0.5f; y = (a1[1] + a3[3]) * 1e-30f + 0.25f; } break;
    case O_MIX: for (int i = 0; i < 8; i++) { float t; asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_rcp_f32 %1, %1\n\tds_read_b32 %2, %3\n\tv_pk_fma_f32 %4, %4, %4, %4\n\tv_sqrt_f32 %1, %1\n\tv_fma_f32 %0, %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %2"
                                                                 : "+v"(x), "+v"(y), "=&v"(t), "+v"(lds_addr), "+v"(d) :: "memory"); } break;
    case O_LANE: for (int i = 0; i < 16; i++) { int sv; asm volatile("v_readlane_b32 %1, %0, 3\n\ts_nop 3\n\tv_writelane_b32 %0, %1, 5\n\tv_readlane_b32 %1, %0, 9\n\ts_nop 3\n\tv_writelane_b32 %0, %1, 11\n\tv_readfirstlane_b32 %1, %0\n\ts_nop 3\n\tv_writelane_b32 %0, %1, 17" : "+v"(x), "=&s"(sv)); } break;
    case O_F64T: for (int i = 0; i < 8; i++) { double e = d + 1.5, q; asm volatile("v_rcp_f64 %1, %0\n\tv_rsq_f64 %1, %1\n\tv_div_scale_f64 %1, vcc, %0, %0, %1\n\tv_div_fmas_f64 %1, %1, %0, %0\n\tv_div_fixup_f64 %1, %1, %0, %0\n\tv_ldexp_f64 %1, %1, 2\n\tv_cvt_f32_f64 %2, %1\n\tv_cvt_f64_f32 %0, %2"
                                                                            : "+v"(e), "=&v"(q), "+v"(x) :: "vcc"); d = e * 1e-300; } break;
    case O_SLEEP: for (int i = 0; i < 8; i++) { int sv; asm volatile("s_sleep 1\n\ts_wakeup\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_sleep 1\n\ts_wakeup" : "=&s"(sv) : "s"(gmem) : "memory"); } break;
    case O_VMEM: for (int i = 0; i < 8; i++) { float t; asm volatile("global_load_dword %0, %1, off\n\tglobal_load_dword %0, %1, off offset:256\n\ts_waitcnt vmcnt(0)\n\tglobal_store_dword %1, %0, off offset:512\n\tglobal_atomic_or %1, %2, off offset:1024\n\ts_waitcnt vmcnt(0)"
                                                                     : "=&v"(t) : "v"(gmem + (lds_addr >> 2)), "v"(0) : "memory"); x += t * 1e-30f; } break;
    default: break;
    }
}
template <int SHAPE, int READER, int W>
__global__ __launch_bounds__(1024) void k(const float* A, const float* B, unsigned long long* bad, int iters, int other, float* gmem) {
    extern __shared__ float lds[];
    volatile int* done = (volatile int*)(lds + blockDim.x * 16);
    if (threadIdx.x == 0) *done = 0;
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned addr = (unsigned)(threadIdx.x * 64);               // 16 words per lane
    const int wave = threadIdx.x >> 6;
    float a = A[t], b = B[t];
    if (wave < 4 || other == O_SAME) {
        unsigned long long cnt[4] = {0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
            float ref[4], got[4];
            const float c0 = a - b;
            chain_then_read<SHAPE, READER, 24>(addr, c0, a, b, ref);
            chain_then_read<SHAPE, READER, W>(addr, c0, a, b, got);
            for (int i = 0; i < 4; i++) cnt[i] += __float_as_uint(ref[i]) != __float_as_uint(got[i]);
            a = a * 1.0001f + 0.001f; b = b * 0.9999f - 0.002f;
            for (int d = 0; d < ((wave * 7 + it) & 15); d++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(0.0f));          // the waves of a SIMD drift against each other
        }
        for (int i = 0; i < 4; i++) if (cnt[i]) atomicAdd(&bad[i], cnt[i]);
        if ((threadIdx.x & 63) == 0) atomicAdd((int*)done, 1);
    } else {
        double d = a;
        float x = a + 1.5f, y = b + 1.5f;
        while (*done < 4) other_stream(other, x, y, d, addr, gmem + (size_t)blockIdx.x * 32768);
        if (x + y + (float)d == 123.456f) bad[7] = 1;
    }
}
static float* g_mem;
template <int SHAPE, int READER, int W>
static void run(const float* A, const float* B, unsigned long long* bad, int block, int iters, int other) {
    hipMemset(bad, 0, 64);
    hipLaunchKernelGGL((k<SHAPE, READER, W>), dim3(256), dim3(block), block * 64 + 64, 0, A, B, bad, iters, other, g_mem);
    unsigned long long hb[4]; hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
    printf(" %2d:%llu/%llu/%llu/%llu", W, hb[0], hb[1], hb[2], hb[3]);
}
template <int SHAPE, int READER>
static void sweep(const float* A, const float* B, unsigned long long* bad, int iters) {
    printf("\n%s -> %s; stale rows 0/1/2/3 per number of wait states, of %lld lane-reads\n", SHAPE == 4 ? "v_mfma_f32_4x4x1_16b_f32 (chain of 8, s_nop 1 between)" : SHAPE == 5 ? "v_mfma_f32_4x4x1_16b_f32 (chain of 64, s_nop 1 between)" : "v_mfma_f32_16x16x4_f32 (chain of 4)",
           READER == 0 ? "ds_write2_b32 rows 2,3 then 0,1" : "v_mov rows 2,3,0,1", 256LL * 256 * iters);
    for (int block : {512, 1024})
        for (int other = (block == 512 ? O_NONE : O_SAME); other < O_COUNT; other++) {
            const int blk = other == O_NONE ? 256 : block;
            printf("  %d wave(s) per SIMD, beside: %-22s", blk / 256, o_name[other]);
            if constexpr (SHAPE == 4 || SHAPE == 5) {
                run<SHAPE, READER, 2>(A, B, bad, blk, iters, other); run<SHAPE, READER, 3>(A, B, bad, blk, iters, other); run<SHAPE, READER, 4>(A, B, bad, blk, iters, other); run<SHAPE, READER, 5>(A, B, bad, blk, iters, other);
                run<SHAPE, READER, 6>(A, B, bad, blk, iters, other); run<SHAPE, READER, 7>(A, B, bad, blk, iters, other); run<SHAPE, READER, 8>(A, B, bad, blk, iters, other); run<SHAPE, READER, 12>(A, B, bad, blk, iters, other);
            } else {
                run<16, READER, 6>(A, B, bad, blk, iters, other); run<16, READER, 8>(A, B, bad, blk, iters, other); run<16, READER, 9>(A, B, bad, blk, iters, other); run<16, READER, 10>(A, B, bad, blk, iters, other);
                run<16, READER, 11>(A, B, bad, blk, iters, other); run<16, READER, 12>(A, B, bad, blk, iters, other); run<16, READER, 13>(A, B, bad, blk, iters, other); run<16, READER, 14>(A, B, bad, blk, iters, other);
            }
            printf("\n");
        }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int n = 256 * 1024;
    float *A, *B; unsigned long long* bad;
    hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&bad, 64); hipMalloc(&g_mem, (size_t)256 * 32768 * 4 + 8192); hipMemset(g_mem, 0, (size_t)256 * 32768 * 4 + 8192);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 977) / 977.f - 0.5f;
    hipMemcpy(A, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; i++) h[i] = (float)(i % 613) / 613.f + 0.1f;
    hipMemcpy(B, h, n * 4, hipMemcpyHostToDevice);
    printf("hipcc's own padding for these readers: 4 wait states behind the 4x4x1, 10 behind the 16x16x4 (ROCm 7.2, gfx950); the product kernels wait 12 behind either\n");
    sweep<4, 0>(A, B, bad, iters); sweep<5, 0>(A, B, bad, iters / 4); sweep<4, 1>(A, B, bad, iters); sweep<16, 0>(A, B, bad, iters); sweep<16, 1>(A, B, bad, iters);
    return 0;
}
