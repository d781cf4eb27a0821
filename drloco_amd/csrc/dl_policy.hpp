// dl_policy.hpp -- the policy forward pass of the rollout loop as one fused kernel.
//
// Restates SB3 1.0 ActorCriticPolicy.forward for the reference's CustomActorCriticPolicy
// (/root/reference/drloco/custom/policies.py:13-51; constructed at drloco/train.py:105-118):
//     latent  = tanh(W2 tanh(W1 obs + b1) + b2)          policy_net and value_net are built from the SAME layer
//                                                         objects (:33-41), i.e. one shared trunk
//     mean    = Wa latent + ba,  value = Wv latent + bv
//     action  = mean + exp(log_std) * eps,  eps ~ N(0, 1)   (DiagGaussianDistribution.sample)
//     log_prob = sum_a( -eps_a^2 / 2 - log_std_a - log(2 pi) / 2 )
// This is the one genuinely GEMM-shaped piece of the path: [N,29]x[29,512], [N,512]x[512,512], [N,512]x[512,9].
// One workgroup (4 or 8 waves) owns 16 walkers; the three layers run back to back on v_mfma_f32_16x16x4f32 (float32 in,
// float32 accumulate -- parity with the float32 torch module to ~1e-6), the activations never leave LDS.
//   A operand (activations): lane l supplies row l % 16, the reduction index is permuted so that a lane's four k
//     values of a 16-wide k block are contiguous: one ds_read_b128 feeds four MFMAs;
//   B operand (weights, torch layout [out][in]): lane l supplies output column n0 + l % 16 with the same four k:
//     one global_load_dwordx4 per tile and k block; two k blocks (one 128-byte line per weight row) per step, in two
//     ping-pong register sets loaded by inline asm one step ahead of the 64 MFMAs that consume them;
//   layer 2: each wave owns hidden/NW output columns (NTW accumulator tiles); heads: the reduction is split over
//     the waves and summed through LDS.
// (With 4 waves x 64 lanes and hidden < 256 `tid < 256` covers the whole workgroup; the epilogue uses 256 lanes.)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "drloco_hip.h"

namespace dl {

typedef float pf4 __attribute__((ext_vector_type(4)));

// EVERY MFMA of the policy kernels is inline asm with the accumulator TIED to the destination ("+v": source C and destination are the same
// registers) and every wait state around it written by hand, AS v_nop, NEVER AS s_nop.  Why (EXPERIMENTS.md, round 5, "the 4x4x1 defect, found";
// tools/ubench/snop_wakeup.hip, profiles/r05_snop_wakeup.txt): **an s_wakeup executed by another wave of the workgroup ends the s_nop this wave is
// in** -- whatever its count, it is over after one wait state (27 % of the readers 8 states behind a 4x4x1 saw the stale accumulator beside a wave
// that loops over s_wakeup; none beside s_sleep, s_load, VALU, LDS, MFMA or memory streams; none when the wait is v_nop or split over several
// s_nop).  The split workgroups hand over with s_sleep / s_wakeup (dl_group.hpp), so inside the rollout kernels any single s_nop can shrink to one
// state: hipcc's own `s_nop 3` between a v_mfma_f32_4x4x1 and the LDS store of its result did, about once in a thousand rows (the round-4 defect of
// dl_policy_pair.hpp; the relocated accumulators round 4 blamed are innocent).  hipcc pads nothing around inline asm, so the asm form is what keeps
// its s_nop out of these kernels; tools/check_mfma_overlap.py proves on the listing of every build that the waits are there, counting an s_nop as
// ONE state (rules R1 .. R5).  The need, measured (tools/ubench/mfma_ds_store.hip): a reader of a 16x16x4 result 9 (LDS store) / 10 (VALU) wait
// states, of a 4x4x1 result 3 / 4.
//   DL_MFMA16       the chain form: back to back on one accumulator (the 8-pass shape interlocks on an exactly matching source C) or interleaved;
//   DL_MFMA16_OPEN  the first instruction after a VALU write of an operand (the zero-initialised accumulator): two wait states in front;
//   DL_MFMA16_SETTLE before any other reader or writer of the accumulator: 16 x v_nop (three times per forward pass and tile).
#define DL_VNOP2 "v_nop\n\tv_nop"
#define DL_VNOP4 DL_VNOP2 "\n\t" DL_VNOP2
#define DL_VNOP16 DL_VNOP4 "\n\t" DL_VNOP4 "\n\t" DL_VNOP4 "\n\t" DL_VNOP4
#define DL_MFMA16(ACC, AV, BV) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(AV), "v"(BV))
#define DL_MFMA16_OPEN(ACC, AV, BV) asm volatile(DL_VNOP2 "\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(AV), "v"(BV))
#define DL_MFMA16_PAD(ACC) asm volatile(DL_VNOP2 : "+v"(ACC))          // behind a VALU write of an accumulator whose first MFMA is a plain DL_MFMA16
#define DL_MFMA16_SETTLE(ACC) asm volatile(DL_VNOP16 : "+v"(ACC))

__device__ __forceinline__ float pol_tanh(float x) {
    // 1 - 2 / (exp(2x) + 1); |error| < 2e-7 absolute
    const float xc = fminf(fmaxf(x, -15.0f), 15.0f);
    const float e = __expf(2.0f * xc);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ uint64_t pol_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// counter-based standard normal keyed by (seed, step counter, global walker index, action index): the sampled
// actions do not depend on how the walkers are sharded over GPUs
__device__ __forceinline__ float pol_gauss(uint64_t seed, uint64_t counter, uint32_t genv, uint32_t a) {
    const uint64_t r = pol_mix(seed ^ pol_mix(counter * 0x100000001B3ull + (((uint64_t)genv << 8) | a)));
    const float u1 = ((uint32_t)(r >> 40) + 1u) * (1.0f / 16777216.0f);       // (0, 1]
    const float u2 = (uint32_t)(r & 0xFFFFFFu) * (1.0f / 16777216.0f);         // [0, 1)
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

#ifndef DL_POL_SKEW
#define DL_POL_SKEW 0      // experiment switch: k > 0 lets wave w of the barrier-free form start its walk over the k blocks at block k w (measured: no gain; 0 keeps the forms bit-identical)
#endif
constexpr int POL_ROWS = 16;       // walkers per workgroup
constexpr int POL_MAXT = 8;        // accumulator tiles per wave in the hidden layer (hidden <= 512)
constexpr int POL_SLD = 36;        // row stride of a staged 16 x 32 activation block (floats)
constexpr int POL_PLD = 20;        // row stride of a wave's private 16 x 16 tile
// dynamic LDS of k_policy_forward<NTW, NW>: two staged activation blocks + per wave a private tile and a partial head tile
constexpr size_t pol_lds_bytes(int nw) { return ((size_t)2 * POL_ROWS * POL_SLD + (size_t)nw * POL_ROWS * POL_PLD + (size_t)nw * 256) * sizeof(float); }
// WHOLE_H1 form: the whole 16 x hidden activation block of the first layer is staged once (row stride hidden + 4), the hidden layer's
// reduction then runs WITHOUT workgroup barriers -- the two waves of a SIMD drift apart and one's weight-load latency falls under the other's
// MFMAs.  51 KB for hidden 512: for callers with the CU's LDS to themselves (<= 4096 rows, one handle; the persistent rollout kernel, whose
// env regions are idle during the policy phase).  Same accumulation order as the lean form: bit-identical results.
constexpr size_t pol_lds_bytes_whole(int nw, int hidden, int rb = 1) { return ((size_t)rb * POL_ROWS * (hidden + 4) + (size_t)nw * POL_ROWS * POL_PLD + (size_t)rb * nw * 256) * sizeof(float); }

// VecNormalize.step_wait's second half (k_vn_apply) folded into the policy's input stage: with raw_obs != NULL the kernel reads the
// raw observation / reward of the last env step, normalises them with the (already updated) moments exactly as k_vn_apply does,
// stores them where the rollout buffer wants them (obs_out = observations[t + 1], rew_out = rewards[t]) and runs the forward pass
// on the normalised observation: one launch less per control step, the same bits.
struct PolVnFuse {
    const float* raw_obs; const float* raw_rew; const uint8_t* done;
    const double* mean; const double* var; double* count;
    double* ret; const double* ret_var; double* ret_count;
    float* obs_out; float* rew_out;
    double eps, clip_obs, clip_rew;
    int flags;
};

// VecNormalize's two scalings, one definition for every kernel that applies them (k_vn_apply / k_vns_apply, the policy's folded input
// stage, the persistent rollout kernel): the same float64 expression -> the same bits wherever a value is normalised
__device__ __forceinline__ float vn_norm_obs(float x, double mean, double var, double eps, double clip) {
    double y = ((double)x - mean) / sqrt(var + eps);
    y = y < -clip ? -clip : (y > clip ? clip : y);
    return (float)y;
}
__device__ __forceinline__ float vn_norm_rew(float r, double ret_var, double eps, double clip) {
    double y = (double)r / sqrt(ret_var + eps);
    y = y < -clip ? -clip : (y > clip ? clip : y);
    return (float)y;
}

// The policy's weights in k-chunk-major order (k_pack_policy), or all NULL = read torch's layout directly:
//   w2p[(k / 4) * H + n][k % 4] = w2[n][k];   w1p[(k / 4) * H + n][k % 4] = w1[n][k] (k < obs_dim, else 0; 12 chunks = 48 inputs);
//   whp[(k / 4) * 16 + j][k % 4] = wa[j][k] (j < act_dim), wv[0][k] (j == act_dim), 0 (else)
struct PolPacked { const float* w2p; const float* w1p; const float* whp; };
constexpr size_t pol_packed_floats(int hidden) { return (size_t)hidden * hidden + (size_t)48 * hidden + (size_t)hidden * 16; }

// NTW = accumulator tiles per wave in the hidden layer, NW = waves per workgroup: hidden = 16 * NTW * NW (compile time, so
// that the tile loops are straight-line code).  hidden = 512 runs as 8 waves x 4 tiles: two waves per SIMD, so that the LDS /
// weight-load latency of one overlaps the MFMAs of the other.
// The activations never exist as whole [16, hidden] arrays in LDS: a wave keeps the 16 x 64 block of h1 (h2) it computed in
// registers (accumulator layout) and the hidden layer's reduction walks over h1 in blocks of 32 columns that their owner
// stages through a double-buffered 16 x 32 block (4.5 KB); the heads transpose h2 tile by tile through a private 16 x 16 tile
// per wave.  23 KB of LDS per workgroup instead of 74 KB: the kernel fits next to four resident workgroups of the env-step
// kernel on a CU (160 KB), which is what lets the policy of one half of the walkers run under the simulation of the other.
// The forward pass for the POL_ROWS rows starting at row0, executed by the 64 * NW lanes of a workgroup (device function: k_policy_forward
// wraps it one workgroup per 16 rows; the persistent rollout kernel k_rollout_persistent calls it once per control step for the rows of its
// own sixteen walkers).  sm: pol_lds_bytes(NW) bytes of LDS.  count_owner: this workgroup advances the moment counts of a folded
// VecNormalize step (exactly one workgroup of a launch does).  tid: threadIdx.x (a parameter so that a caller looping over control steps can
// pass it opaque per step: the lane's index arithmetic is then redone per step instead of being kept in registers across the other phases).
// PACKED: the hidden layer's weights are read from `w2p`, a copy of w2 in k-chunk-major order (k_pack_w2: w2p[(k / 4) * H + n][k % 4]).  In
// torch's [out][in] layout the 16 lanes of a quarter-wave (one output column each) read 16 bytes from 16 DIFFERENT 2 KB-apart rows: 64 cache
// lines per wave instruction, one tag lookup each -- the load path then delivers ~14 bytes per clock and CU and the kernel is bound by it
// (measured: 13 of 37 us).  Packed, a quarter-wave's 16 x 16 bytes are one contiguous 256-byte run.  Same values, same order of arithmetic.
#ifdef DL_EXP_POL_PROF          // diagnostics build: shader-clock stamps of workgroup 0's waves 0 and 7 at the section boundaries of the forward pass (tools/diag_policy.py, diag_policy_rollout.py)
__device__ long long g_pol_prof[2][8];
#define DL_POL_STAMP(k) do { if (row0 == 0 && l == 0 && (wave == 0 || wave == NW - 1)) g_pol_prof[wave ? 1 : 0][k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DL_POL_STAMP(k) ((void)0)
#endif
// RB = row blocks per call (barrier-free packed form only): RB x 16 rows share every weight register set -- the hidden layer's stream per row halves with RB = 2 and the
// layer is bound by the matrix pipe again (a caller that owns several blocks of sixteen walkers: the persistent rollout kernel).  Per row the arithmetic is that of RB = 1.
template <int NTW, int NW, bool WHOLE_H1 = false, bool PACKED = false, int RB = 1>
__device__ __forceinline__ void pol_forward_rows(const dl_policy_params& p, const float* __restrict__ obs, int n, const float* __restrict__ eps,
                                                 uint64_t seed, uint64_t counter, int index_base, int deterministic,
                                                 float* __restrict__ actions, float* __restrict__ values, float* __restrict__ logp, const PolVnFuse& vf,
                                                 float* sm, int row0, bool count_owner, int tid, const PolPacked pk = PolPacked{nullptr, nullptr, nullptr}) {
    constexpr int H = 16 * NTW * NW, ntw = NTW;
    static_assert(RB == 1 || (WHOLE_H1 && PACKED), "several row blocks per call exist for the barrier-free packed form");
    const int D = p.obs_dim, A = p.act_dim;
    const int wave = tid >> 6, l = tid & 63, lm = l & 15, lk = l >> 4;
    constexpr int HLD = H + 4;                                             // row stride of the whole-h1 block
    constexpr int STAGE_WORDS = WHOLE_H1 ? RB * POL_ROWS * HLD : 2 * POL_ROWS * POL_SLD;
    float* stage = sm;                                                     // [2][16][POL_SLD], or [RB][16][HLD]
    float* priv = sm + STAGE_WORDS + wave * (POL_ROWS * POL_PLD);          // this wave's [16][POL_PLD]
    float* part = sm + STAGE_WORDS + NW * (POL_ROWS * POL_PLD);            // [RB][NW][16][16] partial head tiles, then [RB][16][16] log-prob terms
    constexpr int ncw = H / NW;
    const int n0w = wave * ncw;
    auto wave_sync = [&]() {      // LDS operations of one wave execute in order: exchanging data inside the wave needs no s_barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    DL_POL_STAMP(0);
    // ---- reward normalisation / bookkeeping of the folded VecNormalize step (k_vn_apply's second half)
    if (vf.raw_obs) {
        if (tid < RB * POL_ROWS && row0 + tid < n) {
            const int r = row0 + tid;
            const float x = vf.raw_rew[r];
            vf.rew_out[r] = (vf.flags & 8) ? vn_norm_rew(x, *vf.ret_var, vf.eps, vf.clip_rew) : x;
            if ((vf.flags & 4) && vf.done[r]) { double zero = 0.0; asm volatile("" : "+v"(zero)); vf.ret[r] = zero; }          // (formed here: as a constant pair it is hoisted out of a caller's step loop and spilled)
        }
        if (count_owner && tid == 0) { if (vf.flags & 1) *vf.count += (double)n; if (vf.flags & 4) *vf.ret_count += (double)n; }
    }
    // ---- layer 1: [16, D] x [D, H].  D is small (29): all operand loads of the wave are issued before the first MFMA
    // (one memory latency for the layer instead of one per tile).  h1 stays in registers (accumulator layout).
    float h1r[RB][NTW][4];
    {
        constexpr int KB1 = 3;                                      // obs_dim <= 48 (checked by the host): 29 (straight walker), 47 (165 cm walker)
        constexpr int OLD = 52;                                     // row stride of the staged observation block (conflict-free ds_read_b128)
        float a1[RB][KB1 * 4], b1v[NTW][KB1 * 4];
        // the weights first (their latency covers the staging below) ...
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int ncol = n0w + t * 16 + lm;
            if constexpr (PACKED) {
#pragma unroll
                for (int kb = 0; kb < KB1; kb++) {
                    const pf4 v = *(const pf4*)(pk.w1p + ((size_t)(kb * 4 + lk) * H + ncol) * 4);
                    b1v[t][kb * 4] = v.x; b1v[t][kb * 4 + 1] = v.y; b1v[t][kb * 4 + 2] = v.z; b1v[t][kb * 4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < KB1 * 4; q++) {
                    const int k = (q >> 2) * 16 + lk * 4 + (q & 3);
                    b1v[t][q] = (k < D) ? p.w1[(size_t)ncol * D + k] : 0.0f;
                }
            }
        }
        // ... then the 16 observation rows, ONCE per workgroup: an element per lane (coalesced), normalised by the folded VecNormalize step
        // where there is one (float64 divide and square root: eight waves used to repeat them for all twelve of a lane's values), staged in
        // LDS (the space of the partial head tiles, unused until the heads), read back as the A fragments of every wave
        float* ostage = part;
        {
            const float* src = vf.raw_obs ? vf.raw_obs : obs;
            for (int e = tid; e < RB * POL_ROWS * 48; e += 64 * NW) {
                const int rr = e / 48, k = e % 48, r = row0 + rr;
                float x = 0.0f;
                if (k < D && r < n) {
                    x = src[(size_t)r * D + k];
                    if (vf.raw_obs) {
                        if (vf.flags & 2) x = vn_norm_obs(x, vf.mean[k], vf.var[k], vf.eps, vf.clip_obs);
                        vf.obs_out[(size_t)r * D + k] = x;
                    }
                }
                ostage[rr * OLD + k] = x;
            }
        }
        __syncthreads();
        DL_POL_STAMP(1);
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
#pragma unroll
            for (int kb = 0; kb < KB1; kb++) {
                const pf4 v = *(const pf4*)&ostage[(rb * POL_ROWS + lm) * OLD + kb * 16 + lk * 4];
                a1[rb][kb * 4] = v.x; a1[rb][kb * 4 + 1] = v.y; a1[rb][kb * 4 + 2] = v.z; a1[rb][kb * 4 + 3] = v.w;
            }
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int ncol = n0w + t * 16 + lm;
            const float bias = p.b1[ncol];
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                pf4 acc = {0.f, 0.f, 0.f, 0.f};
                DL_MFMA16_OPEN(acc, a1[rb][0], b1v[t][0]);
#pragma unroll
                for (int q = 1; q < KB1 * 4; q++) DL_MFMA16(acc, a1[rb][q], b1v[t][q]);
                DL_MFMA16_SETTLE(acc);
#pragma unroll
                for (int i = 0; i < 4; i++) h1r[rb][t][i] = pol_tanh(acc[i] + bias);
            }
        }
    }
    // the two 16-column tiles of k pair kp, staged by the wave(s) that own them
    auto publish = [&](int kp) {
        float* buf = stage + (kp & 1) * (POL_ROWS * POL_SLD);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int tt = 2 * kp + half;
            if (tt / NTW == wave) {
                const int lt = tt % NTW;
#pragma unroll
                for (int t = 0; t < NTW; t++)
                    if (t == lt) {
#pragma unroll
                        for (int i = 0; i < 4; i++) buf[(4 * lk + i) * POL_SLD + half * 16 + lm] = h1r[0][t][i];
                    }
            }
        }
    };
    // ---- layer 2: [16, H] x [H, H]; the weight rows of this wave's tiles stream from L2 one k block ahead
    float h2r[RB][NTW][4];
    {
        pf4 acc[RB][NTW];
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
#pragma unroll
            for (int t = 0; t < NTW; t++) { acc[rb][t] = pf4{0.f, 0.f, 0.f, 0.f}; DL_MFMA16_PAD(acc[rb][t]); }
        const float* wbase = PACKED ? pk.w2p + ((size_t)lk * H + n0w + lm) * 4 : p.w2 + (size_t)(n0w + lm) * H + lk * 4;
        // two k blocks (32 k = one 128-byte line per weight row) per step: both halves of every line a wave touches are
        // consumed together.  Explicit ping-pong register sets: the loads of the next step are issued BEFORE the 64 MFMAs
        // of the current one (with one buffer the compiler reuses the registers and every step waits a full L2 latency).
        const int npair = H / 32;                 // even for every supported hidden size
        // The loads are inline asm: LLVM sinks ordinary loads below the MFMA block to their uses (IR-level, sched_barrier
        // does not help), which serialises every step behind a full memory latency.  The compiler does not know that the
        // asm outputs are still in flight, so the wait is explicit and carries the registers as operands (the MFMAs depend
        // on the waited values).
        auto load_set = [&](pf4 (&b)[NTW][2], int kp) {
#pragma unroll
            for (int t = 0; t < NTW; t++) {
#ifndef DL_EXP_POL_NOLOAD
                if constexpr (PACKED) {
                    const float* q = wbase + (size_t)t * 64 + (size_t)kp * (8 * H * 4);
                    const float* q1 = q + (size_t)4 * H * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[t][0]) : "v"(q) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[t][1]) : "v"(q1) : "memory");
                } else {
                const float* q = wbase + (size_t)t * 16 * H + kp * 32;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[t][0]) : "v"(q) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(b[t][1]) : "v"(q) : "memory");
                }
#else
                const float* q = wbase;
                asm volatile("" : "=v"(b[t][0]) : "v"(q)); asm volatile("" : "=v"(b[t][1]) : "v"(q));
#endif
            }
        };
        // wait until the OLDEST set in flight has arrived while `newer` younger sets (2 * NTW loads each; loads return in order) stay
        // in flight; vmcnt takes an immediate, hence the ladder
        auto wait_set = [&](pf4 (&b)[NTW][2], int newer) {
            constexpr int L = 2 * NTW;
            if (newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L < 63 ? 3 * L : 63) : "memory");
            else if (newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
            else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < NTW; t++) asm volatile("" : "+v"(b[t][0]), "+v"(b[t][1]));
        };
        auto compute = [&](const pf4 (&b4)[NTW][2], int kp) {
            const float* buf = WHOLE_H1 ? stage + kp * 32 : stage + (kp & 1) * (POL_ROWS * POL_SLD);
            constexpr int LD = WHOLE_H1 ? HLD : POL_SLD;
            pf4 a4a[RB], a4b[RB];
#pragma unroll
            for (int rb = 0; rb < RB; rb++) { a4a[rb] = *(const pf4*)&buf[(rb * POL_ROWS + lm) * LD + lk * 4]; a4b[rb] = *(const pf4*)&buf[(rb * POL_ROWS + lm) * LD + 16 + lk * 4]; }
            // k step outermost: consecutive MFMAs go to different accumulator tiles (no back-to-back dependent issue)
#ifndef DL_EXP_POL_NOMFMA
#define DL_POL_KSTEP(AV, H2, C) _Pragma("unroll") for (int t = 0; t < NTW; t++) _Pragma("unroll") for (int rb = 0; rb < RB; rb++) DL_MFMA16(acc[rb][t], AV[rb].C, b4[t][H2].C);
#else
#define DL_POL_KSTEP(AV, H2, C) _Pragma("unroll") for (int t = 0; t < NTW; t++) _Pragma("unroll") for (int rb = 0; rb < RB; rb++) acc[rb][t].x += AV[rb].C * b4[t][H2].C;
#endif
            DL_POL_KSTEP(a4a, 0, x) DL_POL_KSTEP(a4a, 0, y) DL_POL_KSTEP(a4a, 0, z) DL_POL_KSTEP(a4a, 0, w)
            DL_POL_KSTEP(a4b, 1, x) DL_POL_KSTEP(a4b, 1, y) DL_POL_KSTEP(a4b, 1, z) DL_POL_KSTEP(a4b, 1, w)
#undef DL_POL_KSTEP
        };
        // DEPTH register sets, DEPTH - 1 of them in flight while one is consumed (4 sets were measured no faster than 2: the hidden
        // layer is bound by the matrix pipe plus the L2 -> CU stream of the weight matrix, not by the latency of a single load).
#ifndef DL_EXP_POL_DEPTH
#define DL_EXP_POL_DEPTH 2
#endif
        constexpr int DEPTH = ((H / 32) % DL_EXP_POL_DEPTH == 0) ? DL_EXP_POL_DEPTH : 2;
        static_assert((H / 32) % DEPTH == 0, "steps come in groups of DEPTH");
        pf4 bs[DEPTH][NTW][2];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing of the compiler's own loads may be counted against the sets
        // (experiment, DL_POL_SKEW > 0: the waves of the barrier-free form walk over the k blocks in rotated order -- a weight row is 2 KB long, so at
        // a given k block the rows' lines differ only in address bits >= 11 and all waves might ask the same L2 channels; measured: no effect,
        // what bound the weight stream was the number of cache lines per wave instruction, see PACKED)
        auto kp_at = [&](int i) { return WHOLE_H1 ? (i + DL_POL_SKEW * wave) % npair : i; };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; d++) load_set(bs[d], kp_at(d));
        if constexpr (WHOLE_H1) {
            // every wave stages the 64 columns of h1 it owns, ONE barrier, then the reduction runs free
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int t = 0; t < NTW; t++)
#pragma unroll
                    for (int i = 0; i < 4; i++) stage[(rb * POL_ROWS + 4 * lk + i) * HLD + n0w + t * 16 + lm] = h1r[rb][t][i];
            DL_POL_STAMP(2);
            __syncthreads();
            DL_POL_STAMP(3);
            for (int i0 = 0; i0 < npair; i0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; d++) {
                    const int i = i0 + d, ahead = i + DEPTH - 1;
                    if (ahead < npair) load_set(bs[(d + DEPTH - 1) % DEPTH], kp_at(ahead));
                    const int left = npair - 1 - i;
                    wait_set(bs[d], left < DEPTH - 1 ? left : DEPTH - 1);
                    compute(bs[d], kp_at(i));
                }
            }
        } else {
        publish(0);
        __syncthreads();
        for (int kp0 = 0; kp0 < npair; kp0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const int kp = kp0 + d, ahead = kp + DEPTH - 1;
                if (ahead < npair) load_set(bs[(d + DEPTH - 1) % DEPTH], ahead);
                if (kp + 1 < npair) publish(kp + 1);        // its buffer was last read in step kp - 1, which ended with the barrier below
                const int left = npair - 1 - kp;
                wait_set(bs[d], left < DEPTH - 1 ? left : DEPTH - 1);
                compute(bs[d], kp);
                __syncthreads();
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef DL_EXP_POL_NOMFMA
#pragma unroll
        for (int rb = 0; rb < RB; rb++)
#pragma unroll
            for (int t = 0; t < NTW; t++) DL_MFMA16_SETTLE(acc[rb][t]);          // (the first statement waits, the others find the pipe drained: 12 states each, once per forward pass)
#endif
        DL_POL_STAMP(4);
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const float bias = p.b2[n0w + t * 16 + lm];
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int i = 0; i < 4; i++) h2r[rb][t][i] = pol_tanh(acc[rb][t][i] + bias);
        }
    }
    // ---- heads: one 16 x 16 tile (columns 0..A-1 action means, column A the value); the reduction is split over the waves, each
    // over the h2 columns it owns: a tile goes from the accumulator layout to the A-operand layout through the wave's private tile
    {
        const float* wrow = lm < A ? p.wa + (size_t)lm * H : (lm == A ? p.wv : nullptr);
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            pf4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                wave_sync();
#pragma unroll
                for (int i = 0; i < 4; i++) priv[(4 * lk + i) * POL_PLD + lm] = h2r[rb][t][i];
                wave_sync();
                const int k0 = n0w + t * 16 + lk * 4;
                const pf4 a4 = *(const pf4*)&priv[lm * POL_PLD + lk * 4];
                pf4 b4;
                if constexpr (PACKED) b4 = *(const pf4*)(pk.whp + ((size_t)(k0 >> 2) * 16 + lm) * 4);
                else b4 = wrow ? *(const pf4*)(wrow + k0) : pf4{0.f, 0.f, 0.f, 0.f};
                if (t == 0) DL_MFMA16_OPEN(acc, a4.x, b4.x); else DL_MFMA16(acc, a4.x, b4.x);
                DL_MFMA16(acc, a4.y, b4.y);
                DL_MFMA16(acc, a4.z, b4.z);
                DL_MFMA16(acc, a4.w, b4.w);
            }
            DL_MFMA16_SETTLE(acc);
#pragma unroll
            for (int i = 0; i < 4; i++) part[((rb * NW + wave) * 16 + 4 * lk + i) * 16 + lm] = acc[i];
        }
    }
    DL_POL_STAMP(5);
    __syncthreads();
    DL_POL_STAMP(6);
    // ---- epilogue: sample, log-probability, value
    float lp = 0.0f;
    const int row = tid >> 4, col = tid & 15, r = row0 + row;          // row: 0 .. 16 RB - 1
    constexpr int EPI = 256 * RB;
    if (tid < EPI) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; w++) v += part[(((row >> 4) * NW + w) * 16 + (row & 15)) * 16 + col];
        if (r < n) {
            if (col < A) {
                // the two bases as VGPR addresses: an SGPR base that the allocator restores by v_readlane in front of the load rests on ONE `s_nop N` of hipcc's (5 wait states),
                // which an s_wakeup of another wave cuts to one state in the rollout kernels (tests/test_dpp_hazards.py: no multi-state s_nop there)
                const float* ba_ = p.ba; const float* ls_ = p.log_std;
                asm volatile("" : "+v"(ba_), "+v"(ls_));
                const float mean = v + ba_[col], ls = ls_[col];
                const float e = deterministic ? 0.0f : (eps ? eps[(size_t)r * A + col] : pol_gauss(seed, counter, (uint32_t)(index_base + r), (uint32_t)col));
                actions[(size_t)r * A + col] = mean + __expf(ls) * e;
                lp = -0.5f * e * e - ls - 0.91893853320467274178f;
            } else if (col == A) values[r] = v + p.bv[0];
        }
    }
    __syncthreads();
    if (tid < EPI) part[tid] = lp;
    __syncthreads();
    if (tid < EPI && col == 0 && r < n) {
        float s = 0.0f;
        for (int a = 0; a < A; a++) s += part[row * 16 + a];
        logp[r] = s;
    }
    DL_POL_STAMP(7);
}

template <int NTW, int NW, bool WHOLE_H1 = false, bool PACKED = false>
__global__ __launch_bounds__(64 * NW) void k_policy_forward(const dl_policy_params p, const float* __restrict__ obs, int n, const float* __restrict__ eps,
                                                        uint64_t seed, uint64_t counter, int index_base, int deterministic,
                                                        float* __restrict__ actions, float* __restrict__ values, float* __restrict__ logp, const PolVnFuse vf,
                                                        const PolPacked pk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    pol_forward_rows<NTW, NW, WHOLE_H1, PACKED>(p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, logp, vf, sm, (int)blockIdx.x * POL_ROWS, blockIdx.x == 0,
                                                (int)threadIdx.x, pk);
}
// torch's [out][in] matrices -> the k-chunk-major copies of PolPacked (one 16-byte chunk per thread); buf: pol_packed_floats(H) floats
__global__ __launch_bounds__(256) void k_pack_policy(const dl_policy_params p, float* __restrict__ buf) {
    const int H = p.hidden, D = p.obs_dim, A = p.act_dim;
    float* w2p = buf; float* w1p = buf + (size_t)H * H; float* whp = w1p + (size_t)48 * H;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n2 = H * (H / 4), n1 = 12 * H, nh = (H / 4) * 16;
    if (idx < n2) {
        const int n = idx / (H / 4), k4 = idx % (H / 4);          // reads coalesced along k
        *(pf4*)(w2p + ((size_t)k4 * H + n) * 4) = *(const pf4*)(p.w2 + (size_t)n * H + k4 * 4);
    } else if (idx < n2 + n1) {
        const int e = idx - n2, n = e / 12, k4 = e % 12;
        pf4 v;
        for (int i = 0; i < 4; i++) { const int k = k4 * 4 + i; v[i] = k < D ? p.w1[(size_t)n * D + k] : 0.0f; }
        *(pf4*)(w1p + ((size_t)k4 * H + n) * 4) = v;
    } else if (idx < n2 + n1 + nh) {
        const int e = idx - n2 - n1, j = e % 16, k4 = e / 16;
        const float* row = j < A ? p.wa + (size_t)j * H : (j == A ? p.wv : nullptr);
        *(pf4*)(whp + ((size_t)k4 * 16 + j) * 4) = row ? *(const pf4*)(row + k4 * 4) : pf4{0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace dl
