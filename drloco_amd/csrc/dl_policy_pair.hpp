// drloco_amd -- the policy forward for FOUR rows on two waves (a wave pair of a split workgroup), float32, hidden = 512 / 256 / 128, packed weights.
//
// Why: the 16-row MFMA tile of pol_forward_rows makes the four wave pairs of a persistent workgroup meet at every control step (they wait for the
// slowest pair: 17 % of the env phase).  v_mfma_f32_4x4x1_16B_f32 multiplies sixteen independent 4 x 4 blocks per instruction: with the four rows of
// ONE pair as the A operand of every block and 64 different columns as the B operands it runs four rows at the rate at which 16x16x4 runs sixteen, so a
// pair can evaluate the policy of its own four walkers on its own SIMD (drloco/custom/policies.py:13-51, the same network as dl_policy.hpp).
//
// Same bits as pol_forward_rows: v_mfma_f32_16x16x4_f32 rounds like four fused multiply-adds in ascending k, and one 4x4x1 instruction is one such
// multiply-add (tools/ubench/mfma_round.hip: 512 000 / 128 000 random cases identical).  pol_forward_rows hands the lanes of a quarter wave lk the
// k = 16 b + 4 lk + c of every block of sixteen, component c per instruction, so a sum runs over k = c, 4 + c, 8 + c, 12 + c for c = 0 .. 3, block after
// block: the single-k instructions below are issued in exactly that order.  The heads' partial sums per group of 64 columns and their final order are kept too.
//
// Lane layout of a 4x4x1 instruction: lane l belongs to block l / 4; it supplies A[row l % 4] and B[column l % 4 of the block] and receives, in its four
// accumulator registers, the block's column l % 4 for rows 0 .. 3.  With block b = columns 4 b .. 4 b + 3 of a group of 64, lane l owns column l of the group.
#pragma once

#include "dl_policy.hpp"

namespace dl {

constexpr int POLP_OLD = 52;
// LDS of one pair (floats): staged observation [4][52], activations [4][H + 4] (h1, then h2), head partials [8][4][16] (then the log-prob terms [4][16])
constexpr int POLP_XS = 0, POLP_HS = 4 * POLP_OLD;
constexpr int polp_part(int H) { return POLP_HS + 4 * (H + 4); }
constexpr int polp_words(int H) { return polp_part(H) + 8 * 4 * 16; }
constexpr int POLP_WORDS = polp_words(512);          // what a caller reserves: the largest hidden size

// H: the hidden size (512 / 256 / 128: each wave takes H / 2 columns in NCB = H / 128 groups of 64; the heads' eight partial sums run over H / 8 columns each, the partials of
// pol_forward_rows<H / 128, 8>).  h: which wave of the pair (0 / 1: columns (H / 2) h .. (H / 2) h + H / 2 - 1), l: lane.  sync(): a barrier of the pair's two waves that also orders their LDS and global
// accesses (stand-alone kernel: __syncthreads of a two-wave workgroup).  sm: POLP_WORDS floats of LDS owned by the pair.  Rows row0 .. row0 + 3 (those < n).
template <int H, typename SYNC>
__device__ __forceinline__ void pol_forward_pair(const dl_policy_params& p, const float* __restrict__ obs, int n, const float* __restrict__ eps, uint64_t seed, uint64_t counter, int index_base,
                                                 int deterministic, float* __restrict__ actions, float* __restrict__ values, float* __restrict__ logp, const PolVnFuse& vf,
                                                 float* sm, int row0, int h, int l, const PolPacked pk, SYNC sync) {
    static_assert(H == 512 || H == 256 || H == 128, "the pair form covers two waves x NCB groups of 64 columns");
    constexpr int HLD = H + 4, OLD = POLP_OLD, NCB = H / 128, KB2 = H / 16, PC = H / 8;
    const int D = p.obs_dim, A = p.act_dim;
    float* xs = sm + POLP_XS; float* hs = sm + POLP_HS; float* part = sm + polp_part(H);
    const int tid = h * 64 + l, rsel = l & 3;
    // ---- the folded VecNormalize step's reward half (as in pol_forward_rows)
    if (vf.raw_obs && tid < 4 && row0 + tid < n) {
        const int r = row0 + tid;
        const float x = vf.raw_rew[r];
        vf.rew_out[r] = (vf.flags & 8) ? vn_norm_rew(x, *vf.ret_var, vf.eps, vf.clip_rew) : x;
        if ((vf.flags & 4) && vf.done[r]) { double zero = 0.0; asm volatile("" : "+v"(zero)); vf.ret[r] = zero; }
    }
    // ---- the four observation rows, normalised where a VecNormalize step is folded in, staged for both waves
    {
        const float* src = vf.raw_obs ? vf.raw_obs : obs;
        for (int e = tid; e < 4 * 48; e += 128) {
            const int rr = e / 48, k = e % 48, r = row0 + rr;
            float x = 0.0f;
            if (k < D && r < n) {
                x = src[(size_t)r * D + k];
                if (vf.raw_obs) {
                    if (vf.flags & 2) x = vn_norm_obs(x, vf.mean[k], vf.var[k], vf.eps, vf.clip_obs);
                    vf.obs_out[(size_t)r * D + k] = x;
                }
            }
            xs[rr * OLD + k] = x;
        }
    }
    sync();
    const int colb = h * (H / 2) + l;             // this lane's column of column group cb: colb + 64 cb
    // one block of sixteen k: A fragments a[j] (k = 4 j .. 4 j + 3 of the lane's row), B fragments b[cb][j] (the same k of the lane's column in group cb)
    // The 4x4x1 instructions are inline asm with the accumulator tied to the destination ("+v") and EVERY wait state behind them is written by hand, as v_nop (dl_policy.hpp).
    // History: as builtins, inside the rollout kernel (not in the stand-alone one), about one row in a thousand -- always the third of the four -- came out wrong by ~1e-2 while
    // other pairs of the workgroup ran.  Round 4 blamed the relocated accumulators the listing showed and tied them.  Round 5 bisected the builtin build by patching its assembly
    // (EXPERIMENTS.md, "the 4x4x1 defect, found"): ONE site carries the defect -- the last MFMA of the heads' single-accumulator chain, hipcc's `s_nop 3`, then
    // `ds_write2_b32 v0, v8, v9` storing rows 2 and 3 of the heads' partial sums (row 2 = the first data register the store reads) -- and then found the mechanism
    // (tools/ubench/snop_wakeup.hip): another pair's s_wakeup ends the s_nop this wave is in after one wait state; the store needs three.  `s_nop 2` and `s_nop 3` failed alike, a
    // second s_nop instruction in front (of any length) never did.  What the tied form really changed: hipcc pads nothing around inline asm, so its s_nop left the kernel.
#ifdef DL_EXP_POLP_BUILTIN          // the round-4 form, for tools/asm_bisect.sh only: builtins, hipcc's own padding -- reproduces the stale row
#define DL_POLP_MFMA(ACC, AV, BV) ACC = __builtin_amdgcn_mfma_f32_4x4x1f32(AV, BV, ACC, 0, 0, 0)
#define DL_POLP_SETTLE(ACC) ((void)0)
#define DL_POLP_HEAD_MFMA(ACC, AV, BV) ACC = __builtin_amdgcn_mfma_f32_4x4x1f32(AV, BV, ACC, 0, 0, 0)
#define DL_POLP_HEAD_SETTLE(ACC) ((void)0)
#else
#define DL_POLP_MFMA(ACC, AV, BV) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(AV), "v"(BV))
#define DL_POLP_SETTLE(ACC) do { asm volatile(DL_VNOP16 : "+v"(ACC[0])); _Pragma("unroll") for (int cb_ = 1; cb_ < NCB; cb_++) asm volatile("" : "+v"(ACC[cb_])); } while (0)          // (volatile asm statements keep their order: every accumulator's readers come behind the wait)
#define DL_POLP_HEAD_MFMA(ACC, AV, BV) asm volatile(DL_VNOP4 "\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(AV), "v"(BV))          // one accumulator, back to back: the wait states a dependent instruction needs
#define DL_POLP_HEAD_SETTLE(ACC) asm volatile(DL_VNOP16 : "+v"(ACC))
#endif
    // (dependent instructions three wait states apart: with four accumulators the other three stand between, with two / one v_nop fill up -- rule R5 of the checker)
#ifdef DL_EXP_POLP_BUILTIN
#define DL_POLP_GAP() ((void)0)
#else
#define DL_POLP_GAP() do { if constexpr (NCB == 2) asm volatile(DL_VNOP2); else if constexpr (NCB == 1) asm volatile(DL_VNOP2 "\n\tv_nop"); } while (0)
#endif
#define DL_POLP_PAD(ACC) asm volatile(DL_VNOP2 : "+v"(ACC))          // behind the VALU write that zeroes an accumulator: two wait states before its first MFMA (rule R3; hipcc pads nothing in front of inline asm)
#define DL_POLP_BLOCK(ACC, AF, BF)                                                                                       \
    _Pragma("unroll") for (int c = 0; c < 4; c++) _Pragma("unroll") for (int j = 0; j < 4; j++) {                        \
        _Pragma("unroll") for (int cb = 0; cb < NCB; cb++) DL_POLP_MFMA(ACC[cb], AF[j][c], BF[cb][j][c]);                \
        DL_POLP_GAP(); }
    // ---- layer 1: three blocks of sixteen inputs (obs_dim <= 48)
    pf4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { acc[cb] = pf4{0.f, 0.f, 0.f, 0.f}; DL_POLP_PAD(acc[cb]); }
#pragma unroll
    for (int kb = 0; kb < 3; kb++) {
        pf4 a[4], b[NCB][4];
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = *(const pf4*)&xs[rsel * OLD + kb * 16 + 4 * j];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int j = 0; j < 4; j++) b[cb][j] = *(const pf4*)(pk.w1p + ((size_t)(kb * 4 + j) * H + colb + 64 * cb) * 4);
        DL_POLP_BLOCK(acc, a, b)
    }
    DL_POLP_SETTLE(acc);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const float bias = p.b1[colb + 64 * cb];
#pragma unroll
        for (int i = 0; i < 4; i++) hs[i * HLD + colb + 64 * cb] = pol_tanh(acc[cb][i] + bias);
    }
    sync();
    // ---- layer 2: H / 16 blocks of sixteen k; the weights of the next block are requested before the MFMAs of the current one (inline asm: see dl_policy.hpp)
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { acc[cb] = pf4{0.f, 0.f, 0.f, 0.f}; DL_POLP_PAD(acc[cb]); }
    {
        const float* wb = pk.w2p + (size_t)colb * 4;
        auto load_set = [&](pf4 (&b)[NCB][4], int kb) {
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float* q = wb + ((size_t)(kb * 4 + j) * H + 64 * cb) * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[cb][j]) : "v"(q) : "memory");
                }
        };
        auto wait_set = [&](pf4 (&b)[NCB][4], bool newer) {
            if (newer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NCB) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int j = 0; j < 4; j++) asm volatile("" : "+v"(b[cb][j]));
        };
        pf4 bs[2][NCB][4];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing of the compiler's own loads may be counted against the sets
        load_set(bs[0], 0);
        for (int kb0 = 0; kb0 < KB2; kb0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const int kb = kb0 + d;
                if (kb + 1 < KB2) load_set(bs[d ^ 1], kb + 1);
                pf4 a[4];
#pragma unroll
                for (int j = 0; j < 4; j++) a[j] = *(const pf4*)&hs[rsel * HLD + kb * 16 + 4 * j];
                wait_set(bs[d], kb + 1 < KB2);
                DL_POLP_BLOCK(acc, a, bs[d])
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DL_POLP_SETTLE(acc);
    sync();                                   // both waves have read all of h1: its space takes h2
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const float bias = p.b2[colb + 64 * cb];
#pragma unroll
        for (int i = 0; i < 4; i++) hs[i * HLD + colb + 64 * cb] = pol_tanh(acc[cb][i] + bias);
    }
    sync();
    // ---- heads: the partial sums of pol_forward_rows' eight waves (PC = H / 8 columns each), four per wave here: block b = 4 wq + jq is partial w = 4 h + wq,
    // outputs 4 jq .. 4 jq + 3 (columns 0 .. A - 1 action means, column A the value)
    {
        const int wq = l >> 4, jq = (l >> 2) & 3, w = 4 * h + wq, jo = 4 * jq + rsel;
        pf4 ha = {0.f, 0.f, 0.f, 0.f};
        DL_POLP_PAD(ha);
#pragma unroll
        for (int t = 0; t < PC / 16; t++) {
            pf4 a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k0 = w * PC + t * 16 + 4 * j;
                a[j] = *(const pf4*)&hs[rsel * HLD + k0];
                b[j] = *(const pf4*)(pk.whp + ((size_t)(k0 >> 2) * 16 + jo) * 4);
            }
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int j = 0; j < 4; j++) DL_POLP_HEAD_MFMA(ha, a[j][c], b[j][c]);
        }
        DL_POLP_HEAD_SETTLE(ha);          // <- the site of the round-4 defect: hipcc put ONE `s_nop 3` here, and an s_wakeup of another pair can end it after one state
#pragma unroll
        for (int i = 0; i < 4; i++) part[(w * 4 + i) * 16 + jo] = ha[i];
    }
#undef DL_POLP_BLOCK
#undef DL_POLP_PAD
#undef DL_POLP_GAP
#undef DL_POLP_MFMA
#undef DL_POLP_SETTLE
#undef DL_POLP_HEAD_MFMA
#undef DL_POLP_HEAD_SETTLE
    sync();
    // ---- epilogue: sample, log-probability, value (the order of pol_forward_rows: partials added w = 0 .. 7)
    float lp = 0.0f;
    const int row = tid >> 4, col = tid & 15, r = row0 + row;
    if (tid < 64) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; w++) v += part[(w * 4 + row) * 16 + col];
        if (r < n) {
            if (col < A) {
                const float mean = v + p.ba[col], ls = p.log_std[col];
                const float e = deterministic ? 0.0f : (eps ? eps[(size_t)r * A + col] : pol_gauss(seed, counter, (uint32_t)(index_base + r), (uint32_t)col));
                actions[(size_t)r * A + col] = mean + __expf(ls) * e;
                lp = -0.5f * e * e - ls - 0.91893853320467274178f;
            } else if (col == A) values[r] = v + p.bv[0];
        }
    }
    sync();
    if (tid < 64) part[tid] = lp;
    sync();
    if (tid < 64 && col == 0 && r < n) {
        float s = 0.0f;
        for (int a = 0; a < A; a++) s += part[row * 16 + a];
        logp[r] = s;
    }
}

// stand-alone form (tests, tools): a workgroup of two waves per four rows
template <int H>
__global__ __launch_bounds__(128) void k_policy_forward_pair(const dl_policy_params p, const float* __restrict__ obs, int n, const float* __restrict__ eps, uint64_t seed, uint64_t counter,
                                                             int index_base, int deterministic, float* __restrict__ actions, float* __restrict__ values, float* __restrict__ logp, const PolPacked pk) {
    __shared__ __attribute__((aligned(16))) float sm[polp_words(H)];
    PolVnFuse none{};
    pol_forward_pair<H>(p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, logp, none, sm, (int)blockIdx.x * 4, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, pk,
                     [] { __syncthreads(); });
}

}  // namespace dl
