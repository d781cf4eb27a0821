// dl_kernels.hip -- gfx950 kernels + the C-ABI of include/drloco_hip.h.
//
// Launch geometry of the environment kernels: one walker per lane, one 64-lane wave per
// workgroup (f64: 32 lanes) so that a workgroup's dynamic constraint storage ([slot][lane],
// conflict-free) fits a CU's LDS; with the benchmark's 4096 walkers that is 64 workgroups, each
// on its own CU.  All per-walker state is SoA [field][N] in HBM: every load/store of a wave is
// one coalesced 256-byte (f32) segment.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "dl_host.hpp"
#include "dl_hwprobe.hpp"
#include "dl_policy.hpp"
#include "dl_policy_pair.hpp"

using namespace dl;

// ------------------------------------------------------------------------------------------
// lanes per workgroup: as many as fit the per-lane LDS footprint into one CU's 160 KiB
template <typename T, typename TP> struct KernelGeom {
    static constexpr size_t PER_LANE = (size_t)MemLayout<TP>::TOTAL * sizeof(T);
    static constexpr int BLOCK = PER_LANE * 64 <= 163840 ? 64 : (PER_LANE * 32 <= 163840 ? 32 : 16);
};

template <typename T, typename TP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_env_step(const DevModel<T, TP>* __restrict__ mp, const DevCfg<T> c, const DevState<T> st, const float* __restrict__ actions,
                                                    float* obs, float* rew, uint8_t* done, float* term_obs, float* rew_terms,
                                                    const T* inj_q, const T* inj_v, const int32_t* inj_flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= st.n) return;
    LaneMem<T> mem{(DL_LDS T*)smem + threadIdx.x, BLOCK};
    const DL_CONST DevModel<T, TP>& m = *(const DL_CONST DevModel<T, TP>*)mp;
    env_step_lane<T, TP>(m, c, mem, st, i, actions, obs, rew, done, term_obs, rew_terms, inj_q, inj_v, inj_flags);
}

// mode 0: reset walkers whose need_reset > 0 (auto reset after a step); mode 1: reset walkers
// selected by mask (NULL = all)
template <typename T, typename TP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_env_reset(const DevModel<T, TP>* __restrict__ mp, const DevCfg<T> c, const DevState<T> st, int mode, const uint8_t* mask,
                                                     const int32_t* init_step, const int32_t* init_pos, float* obs, float* term_obs, int eval_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= st.n) return;
    int nrep;
    if (mode == 0) nrep = st.need_reset[i];
    else nrep = (mask == nullptr || mask[i]) ? 1 : 0;
    if (nrep == 0) return;
    LaneMem<T> mem{(DL_LDS T*)smem + threadIdx.x, BLOCK};
    const DL_CONST DevModel<T, TP>& m = *(const DL_CONST DevModel<T, TP>*)mp;
    env_reset_lane<T, TP>(m, c, mem, st, i, nrep, init_step, init_pos, obs, term_obs, eval_mode);
}

template <typename T, typename TP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_forward(const DevModel<T, TP>* __restrict__ mp, const DevState<T> st, const T* ctrl, T* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int i = blockIdx.x * BLOCK + threadIdx.x, n = st.n;
    if (i >= n) return;
    LaneMem<T> mem{(DL_LDS T*)smem + threadIdx.x, BLOCK};
    const DL_CONST DevModel<T, TP>& m = *(const DL_CONST DevModel<T, TP>*)mp;
    T q[TP::NV], v[TP::NV], w[TP::NV], u[TP::NU], a[TP::NV];
    static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; q[j] = st.qpos[(size_t)j * n + i]; v[j] = st.qvel[(size_t)j * n + i]; w[j] = st.warm[(size_t)j * n + i]; });
    static_for<TP::NU>([&](auto ai) { constexpr int k = ai.value; u[k] = ctrl ? ctrl[(size_t)k * n + i] : T(0); });
    const int info = forward_io<T, TP>(m, mem, q, v, u, w, a);
    static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; qacc[(size_t)j * n + i] = a[j]; });
    if (ncon) ncon[i] = info & 255;
    if (nefc) nefc[i] = (info >> 8) & 255;
    if (niter) niter[i] = info >> 16;
}

// forward dynamics with 16 lanes per walker (dl_group.hpp, dl_group_env.hpp): 4 walkers per 64-lane workgroup
template <typename T, typename TP, bool TIMED = false>
__global__ __launch_bounds__(64) void k_forward_g16(const GModel<T, TP>* __restrict__ gm, const DevState<T> st, const T* ctrl, T* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter, long long* tim = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    g_wave_forward<T, TP, TIMED>(threadIdx.x, g_block_of_workgroup(blockIdx.x, gridDim.x), blockIdx.x, gridDim.x, (DL_LDS T*)smem, gm, st, ctrl, qacc, ncon, nefc, niter, tim);
}

// control steps with 16 lanes per walker: action map, frame_skip x RK4 mj_step through g_forward, cursor / observation /
// reward / termination / Monitor, and the vec-env auto reset of finished walkers (RSI draw, mocap lookup, foot-site
// kinematics, first observation) in the same launch (dl_group_env.hpp).
template <typename T, typename TP, bool TIMED = false>
__global__ __launch_bounds__(64) void k_env_step_g16(const GModel<T, TP>* __restrict__ gm, const DevCfg<T> c, const DevState<T> st, const float* __restrict__ actions_all,
                                                     float* obs_all, float* rew_all, uint8_t* done_all, float* term_obs_all, float* rew_terms_all,
                                                     const T* inj_q, const T* inj_v, const int32_t* inj_flags, float* ctrl_out, int eval_mode, int nsteps, long long* tim = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    g_wave_env_step<T, TP, TIMED>(threadIdx.x, g_block_of_workgroup(blockIdx.x, gridDim.x), blockIdx.x, gridDim.x, (DL_LDS T*)smem, gm, c, st, actions_all, obs_all, rew_all, done_all,
                                  term_obs_all, rew_terms_all, inj_q, inj_v, inj_flags, ctrl_out, eval_mode, nsteps, tim);
}

// The same control steps with a SPLIT workgroup (lane-only walker, float32): eight waves serve sixteen walkers -- waves 0..3 are
// dynamics waves (the whole step, g_wave_env_step), waves 4..7 constraint waves (g_constraint_server), the hardware spreads the eight waves
// over the four SIMDs two by two, and constraint wave s serves dynamics wave s + 1 so that a SIMD holds two different groups of walkers
// (the constraint work of one falls into the solver stalls of the other).  Capped at 256 registers: two waves per SIMD.
template <typename T, typename TP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_env_step_g16_split(const GModel<T, TP>* __restrict__ gm, const DevCfg<T> c, const DevState<T> st, const float* __restrict__ actions_all,
                          float* obs_all, float* rew_all, uint8_t* done_all, float* term_obs_all, float* rew_terms_all,
                          const T* inj_q, const T* inj_v, const int32_t* inj_flags, float* ctrl_out, int eval_mode, int nsteps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Sp = GSplit<TP>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, role = wave >> 2, slot = wave & 3;
#ifndef DL_SPLIT_PAIR_OFFSET
#define DL_SPLIT_PAIR_OFFSET 1
#endif
    const int gslot = role == 0 ? slot : ((slot + DL_SPLIT_PAIR_OFFSET) & 3);   // the group of four walkers this wave works for
    DL_LDS T* base = (DL_LDS T*)smem + (size_t)gslot * GW * Sp::TOTAL;
    if (lane == 0) {               // both waves of a pair clear the pair's flags, then the one barrier of this kernel
        volatile DL_LDS int* f = (volatile DL_LDS int*)(base + Sp::MB);
        f[Sp::MB_CMDSEQ] = 0; f[Sp::MB_DONESEQ] = 0; f[Sp::MB_CMD] = 1; f[Sp::MB_MOK] = 0; f[Sp::MB_MFREE] = 0; f[Sp::MB_PRE] = -1;
    }
    __syncthreads();
    // the XCD-aware permutation acts on WORKGROUPS (workgroup b runs on XCD b % 8): a workgroup owns sixteen consecutive walkers = one full
    // 64-byte line of every SoA state row, and neighbouring walker blocks stay on one XCD
    const int vwg = blockIdx.x * 4 + gslot, nvwg = gridDim.x * 4;
#ifdef DL_EXP_OLD_XCD            // experiment switch: the round-2 mapping (permutation of the wave pairs)
    const int wblock = g_block_of_workgroup(vwg, nvwg);
#else
    const int wblock = g_block_of_workgroup(blockIdx.x, gridDim.x) * 4 + gslot;
#endif
    if (role == 0)
        g_wave_env_step<T, TP, false, true>(lane, wblock, vwg, nvwg, base, gm, c, st, actions_all, obs_all, rew_all, done_all, term_obs_all, rew_terms_all,
                                            inj_q, inj_v, inj_flags, ctrl_out, eval_mode, nsteps, nullptr);
    else
        g_constraint_server<T, TP>(lane, wblock, base, gm, st, nsteps > 1 ? actions_all : nullptr, nsteps);
}

// row primitives of dl_group.hpp on known data (tests/test_gpu_parity.py::test_row_primitives)
__global__ __launch_bounds__(64) void k_selftest(const float* in, float* out) {
    const int lane = threadIdx.x;
    const float a = in[lane], b = in[64 + lane];
    out[lane] = gsum(a * b);                  // product feeding the reduction: must stay lane-uniform
    out[64 + lane] = rbcast<3>(a);
    out[128 + lane] = rbcast<15>(a) + rbcast<0>(b);
    out[192 + lane] = (float)gsum((double)a * (double)b);
}

template <typename T> __global__ void k_fill(T* p, T val, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = val;
}
template <typename T> __global__ void k_copy_strided(T* dst, const float* src, int n, int stride, int offset) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (T)src[(size_t)i * stride + offset];
}
template <typename T> __global__ void k_copy_cast(T* dst, const double* src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (T)src[i];
}
// MimicEnv.do_terminate_early (mimic_env.py:652-702; straight walker, 3 trunk joints) for every walker:
// flags[i] = {any, COM height < 0.75, trunk angle (sagittal outside [-0.05, 0.3] or frontal deviation from the reference > 0.2), |COM y| > 0.2}
template <typename T>
__global__ void k_terminate_early(const DevCfg<T> c, const DevState<T> st, int32_t* flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = st.n;
    if (i >= n) return;
    int32_t cur[DL_CUR_WORDS];
#pragma unroll
    for (int k = 0; k < DL_CUR_WORDS; k++) cur[k] = st.cur[(size_t)k * n + i];
    const int base = c.step_off[cur[DL_CUR_READ_STEP]] + cur[DL_CUR_POS];
    const T q1 = st.qpos[(size_t)1 * n + i], q2 = st.qpos[(size_t)2 * n + i], q3 = st.qpos[(size_t)3 * n + i], q4 = st.qpos[(size_t)4 * n + i];
    const bool low = q2 < T(0.75), drunk = dl_abs(q1) > T(0.2);
    const bool front = dl_abs(q3 - ref_at(c, 3, base)) > T(0.2), sag = q4 > T(0.3) || q4 < T(-0.05);
    const bool trunk = sag || front;
    flags[4 * (size_t)i] = low || trunk || drunk; flags[4 * (size_t)i + 1] = low; flags[4 * (size_t)i + 2] = trunk; flags[4 * (size_t)i + 3] = drunk;
}
__global__ void k_mon_snapshot(const double* mon, int word, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mon[(size_t)word * n + i];
}

// ------------------------------------------------------------------------------------------
// SB3-layer reductions

// block reduction of a double over a workgroup (blockDim.x multiple of 64, <= 1024)
__device__ __forceinline__ double block_sum(double x, double* sh) {
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = x;
    __syncthreads();
    double r = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 0; k < nw; k++) r += sh[k];
    return r;
}

// RunningMeanStd.update for column k = blockIdx.x of x[B, D]: batch mean / population variance
// in float64, Chan merge with (mean, var, count).  count is updated by k_count_add afterwards.
template <typename X>
__global__ __launch_bounds__(256) void k_moments(double* mean, double* var, const double* count, const X* __restrict__ x, int B, int D) {
    __shared__ double sh[4];
    const int k = blockIdx.x;
    double s = 0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) s += (double)x[(size_t)i * D + k];
    const double bm = block_sum(s, sh) / B;
    double s2 = 0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { const double d = (double)x[(size_t)i * D + k] - bm; s2 += d * d; }
    const double bv = block_sum(s2, sh) / B;
    if (threadIdx.x == 0) {
        const double cnt = *count, tot = cnt + B, delta = bm - mean[k];
        const double M2 = var[k] * cnt + bv * B + delta * delta * cnt * B / tot;
        mean[k] = mean[k] + delta * B / tot;
        var[k] = M2 / tot;
    }
}
__global__ void k_count_add(double* count, double b) { *count += b; }

__global__ void k_normalize_obs(float* x, const double* mean, const double* var, int B, int D, double eps, double clip) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * D) return;
    const int k = (int)(idx % D);
    double y = ((double)x[idx] - mean[k]) / sqrt(var[k] + eps);
    y = y < -clip ? -clip : (y > clip ? clip : y);
    x[idx] = (float)y;
}
__global__ void k_ret_accumulate(double* ret, const float* rew, int B, double gamma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) ret[i] = ret[i] * gamma + (double)rew[i];
}
__global__ void k_reward_finish(float* rew, double* ret, const uint8_t* done, const double* ret_var, int B, double eps, double clip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double y = (double)rew[i] / sqrt(*ret_var + eps);
    y = y < -clip ? -clip : (y > clip ? clip : y);
    rew[i] = (float)y;
    if (done[i]) ret[i] = 0;
}

// ---- VecNormalize.step_wait fused into two launches -------------------------------------------------------
// k_vn_reduce: ONE workgroup of 1024 lanes.  The batch is small (4096 x 29 floats = 0.5 MB, L2 / MALL resident: the step kernel
// has just written it), so what a launch costs is latency, not bandwidth: a multi-block reduction pays two device-scope
// fences and contended atomics across the 8 XCDs' L2s (measured: 20 us with 32 or 128 blocks), one CU streams 0.5 MB in ~4 us
// and needs neither -- and the summation order is fixed, so the moments are the same bits on every run.
// Thread t always meets column t % D (the slab of a pass is rpb = 1024 / D whole rows, read with coalesced 256-byte wave
// loads); sums are of (x - K), K = the running mean (shift against cancellation), in float64.  The discounted returns
// (ret = ret * gamma + r is advanced here) are column D.  RunningMeanStd's Chan update follows; the counts are read here and
// advanced by k_vn_apply (stream order), so every merge sees the old count.
constexpr int VN_THREADS = 1024;
// one batch: the moments (mean, var, count of the observations / of the discounted returns) are read and updated through generic
// pointers -- global memory in the single-step kernel, LDS in the multi-step kernel, same arithmetic
__device__ __forceinline__ void vn_reduce_phase(const float* __restrict__ x, const float* __restrict__ rew, double* mean, double* var, const double* count,
                                                double* ret, double* ret_mean, double* ret_var, const double* ret_count,
                                                int B, int D, double gamma, int flags, double (*sh)[VN_THREADS], double* sh2) {
    const int t = threadIdx.x;
    double rs = 0, rss = 0;
    // ---- discounted returns first (their loads overlap the observation loads below)
    if (flags & 4) {
        const double K = *ret_mean;
        for (int i = t; i < B; i += VN_THREADS) { const double r = ret[i] * gamma + (double)rew[i]; ret[i] = r; const double d = r - K; rs += d; rss += d * d; }
    }
    // ---- observations: per-column partial sums over this thread's rows
    if (flags & 1) {
        const int rpb = VN_THREADS / D, nthr = rpb * D;       // rows per pass
        double s = 0, ss = 0;
        const int col = t % D, rsub = t / D;
        if (t < nthr) {
            const double K = mean[col];
            int row = rsub;
            for (; row + 15 * rpb < B; row += 16 * rpb) {          // sixteen independent loads in flight: one CU has to cover the latency alone
                float a[16];
#pragma unroll
                for (int k = 0; k < 16; k++) a[k] = x[(size_t)(row + k * rpb) * D + col];
#pragma unroll
                for (int k = 0; k < 16; k++) { const double d = (double)a[k] - K; s += d; ss += d * d; }
            }
            for (; row < B; row += rpb) { const double d = (double)x[(size_t)row * D + col] - K; s += d; ss += d * d; }
        }
        sh[0][t] = s; sh[1][t] = ss;
        __syncthreads();
        if (t < D) {
            for (int r = 1; r < rpb; r++) { s += sh[0][r * D + t]; ss += sh[1][r * D + t]; }
            const double K = mean[t], bm = K + s / B, bv = ss / B - (s / B) * (s / B);
            const double cnt = *count, tot = cnt + B, delta = bm - K;
            const double M2 = var[t] * cnt + bv * B + delta * delta * cnt * B / tot;
            mean[t] = K + delta * B / tot;
            var[t] = M2 / tot;
        }
    }
    if (flags & 4) {
        rs = block_sum(rs, sh2); rss = block_sum(rss, sh2);
        if (t == 0) {
            const double K = *ret_mean, bm = K + rs / B, bv = rss / B - (rs / B) * (rs / B);
            const double cnt = *ret_count, tot = cnt + B, delta = bm - K;
            const double M2 = *ret_var * cnt + bv * B + delta * delta * cnt * B / tot;
            *ret_mean = K + delta * B / tot;
            *ret_var = M2 / tot;
        }
    }
}
__global__ __launch_bounds__(VN_THREADS) void k_vn_reduce(const float* __restrict__ x, const float* __restrict__ rew, double* mean, double* var, const double* count,
                                                          double* ret, double* ret_mean, double* ret_var, const double* ret_count,
                                                          int B, int D, double gamma, int flags) {
    __shared__ double sh[2][VN_THREADS];
    __shared__ double sh2[VN_THREADS / 64];
    vn_reduce_phase(x, rew, mean, var, count, ret, ret_mean, ret_var, ret_count, B, D, gamma, flags, sh, sh2);
}
// The multi-block form of the reduction (flags bit 16): 32 blocks leave per-column partial sums, the last block to arrive merges them in
// block order (deterministic).  Two device-scope fences and an atomic per block make it slower than the one-workgroup kernel when
// a caller waits for it (20 vs 15 us), but its loads are spread over 32 CUs: on a side stream UNDER a running env-step kernel it
// keeps its pace where the single workgroup, sharing one CU with four env-step waves, falls to 39 us -- the choice of the
// overlapped fixed-action path (HipVecNormalize.enable_overlap).
constexpr int VN_BLOCKS = 32;
__global__ __launch_bounds__(256) void k_vn_reduce_mb(const float* __restrict__ x, const float* __restrict__ rew, double* mean, double* var, const double* count,
                                                   double* ret, double* ret_mean, double* ret_var, const double* ret_count,
                                                   int B, int D, double gamma, int flags, double* work, unsigned* arrive) {
    __shared__ double sh[2][256];
    __shared__ double sh2[4];
    __shared__ bool is_last;
    const int t = threadIdx.x, W = D + 1;
    // ---- observations: per-column partial sums of this block's rows
    if (flags & 1) {
        const int rpb = blockDim.x / D, nthr = rpb * D;       // rows per pass of this block
        double s = 0, ss = 0;
        const int col = t % D, rsub = t / D;
        if (t < nthr) {
            const double K = mean[col];
            const int stride = VN_BLOCKS * rpb;
            int row = blockIdx.x * rpb + rsub;
            for (; row + 3 * stride < B; row += 4 * stride) {       // four independent loads in flight, accumulated in row order
                const float a0 = x[(size_t)row * D + col], a1 = x[(size_t)(row + stride) * D + col], a2 = x[(size_t)(row + 2 * stride) * D + col], a3 = x[(size_t)(row + 3 * stride) * D + col];
                const double d0 = (double)a0 - K, d1 = (double)a1 - K, d2 = (double)a2 - K, d3 = (double)a3 - K;
                s += d0; ss += d0 * d0; s += d1; ss += d1 * d1; s += d2; ss += d2 * d2; s += d3; ss += d3 * d3;
            }
            for (; row < B; row += stride) { const double d = (double)x[(size_t)row * D + col] - K; s += d; ss += d * d; }
        }
        sh[0][t] = s; sh[1][t] = ss;
        __syncthreads();
        if (t < D) {
            for (int r = 1; r < rpb; r++) { s += sh[0][r * D + t]; ss += sh[1][r * D + t]; }
            work[((size_t)blockIdx.x * W + t) * 2] = s; work[((size_t)blockIdx.x * W + t) * 2 + 1] = ss;
        }
    }
    // ---- discounted returns: advance this block's slice, partial sums as column D
    if (flags & 4) {
        const int chunk = (B + VN_BLOCKS - 1) / VN_BLOCKS, lo = blockIdx.x * chunk, hi = lo + chunk < B ? lo + chunk : B;
        const double K = *ret_mean;
        double s = 0, ss = 0;
        for (int i = lo + t; i < hi; i += blockDim.x) { const double r = ret[i] * gamma + (double)rew[i]; ret[i] = r; const double d = r - K; s += d; ss += d * d; }
        s = block_sum(s, sh2); ss = block_sum(ss, sh2);
        if (t == 0) { work[((size_t)blockIdx.x * W + D) * 2] = s; work[((size_t)blockIdx.x * W + D) * 2 + 1] = ss; }
    }
    __threadfence();
    __syncthreads();
    if (t == 0) is_last = atomicAdd(arrive, 1u) == (unsigned)(VN_BLOCKS - 1);
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    const bool do_obs = t < D && (flags & 1), do_ret = t == D && (flags & 4);
    if (do_obs || do_ret) {
        double S = 0, SS = 0;
        for (int b = 0; b < VN_BLOCKS; b++) { S += __builtin_nontemporal_load(&work[((size_t)b * W + t) * 2]); SS += __builtin_nontemporal_load(&work[((size_t)b * W + t) * 2 + 1]); }
        double* m = do_obs ? mean + t : ret_mean;
        double* v = do_obs ? var + t : ret_var;
        const double K = *m, bm = K + S / B, bv = SS / B - (S / B) * (S / B);
        const double cnt = do_obs ? *count : *ret_count, tot = cnt + B, delta = bm - K;
        const double M2 = *v * cnt + bv * B + delta * delta * cnt * B / tot;
        *m = K + delta * B / tot;
        *v = M2 / tot;
    }
    if (t == 0) *arrive = 0;
}
// k_vn_apply: obs_out = clip((obs - mean)/sqrt(var + eps)); rew_out = clip(r/sqrt(ret_var + eps)); ret[done] = 0; counts += B
__global__ __launch_bounds__(256) void k_vn_apply(const float* __restrict__ x, const float* __restrict__ rew, const uint8_t* __restrict__ done,
                                                  const double* __restrict__ mean, const double* __restrict__ var, double* count,
                                                  double* ret, const double* __restrict__ ret_var, double* ret_count,
                                                  int B, int D, double eps, double clip_obs, double clip_rew, int flags, float* obs_out, float* rew_out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && !(flags & 64)) { if (flags & 1) *count += (double)B; if (flags & 4) *ret_count += (double)B; }
    if (idx < (size_t)B * D) {
        const int k = (int)(idx % D);
        obs_out[idx] = (flags & 2) ? vn_norm_obs(x[idx], mean[k], var[k], eps, clip_obs) : x[idx];
    }
    if (idx < (size_t)B) {
        rew_out[idx] = (flags & 8) ? vn_norm_rew(rew[idx], *ret_var, eps, clip_rew) : rew[idx];
        if ((flags & 4) && done[idx]) ret[idx] = 0;
    }
}

// RolloutBuffer.compute_returns_and_advantage.  The recurrence A_t = delta_t + c_t A_{t+1} (c_t = gamma lambda (1 - start_{t+1}))
// is affine in A_{t+1}, so the T axis is cut into chunks of GAE_CL steps that run in parallel.  ONE kernel, every input read once:
// a workgroup owns GAE_W = 32 neighbouring walkers (128-byte row segments) and GAE_CH = 32 chunks -- 512 steps per pass, longer
// rollouts take further passes from the end of the rollout towards its start, the advantage carried in LDS.  A thread
//   loads the GAE_CL steps of its (walker, chunk) into registers (all loads in flight at once) and scans them with A_in = 0,
//   publishes the chunk's affine map (offset a = its first advantage, slope b = prod c_t) in LDS,
//   composes the maps of the later chunks (<= 31 steps) into its A_in,
//   writes A_t = A_t(local) + B_t A_in with the running product B_t, returns = A + V.
// 17 bytes per sample of memory traffic (9 read, 8 written), no scratch buffer.  Float32; the grouping of the operations differs
// from SB3's single loop by rounding only (<= 1e-6 relative, tests/test_gpu_parity.py::test_sb3_reductions).
constexpr int GAE_CL = 16, GAE_CH = 32, GAE_W = 32;
__global__ __launch_bounds__(GAE_CH * GAE_W) void k_gae_fused(const float* __restrict__ rew, const float* __restrict__ val, const uint8_t* __restrict__ ep_start,
                                                              const float* __restrict__ last_val, const uint8_t* __restrict__ last_done, float gamma, float lam, int T, int N,
                                                              float* __restrict__ adv, float* __restrict__ ret) {
    __shared__ float2 ab[GAE_CH][GAE_W];
    __shared__ float carry[GAE_W];
    const int wi = threadIdx.x % GAE_W, k = threadIdx.x / GAE_W;       // k: chunk inside the pass, 0 = the latest
    const int i = blockIdx.x * GAE_W + wi;
    const bool valid = i < N;
    const size_t ii = valid ? i : N - 1;
    if (k == 0) carry[wi] = 0.f;
    const float gl = gamma * lam;
    for (int top = T - 1; top >= 0; top -= GAE_CL * GAE_CH) {
        const int t1 = top - k * GAE_CL;                               // this thread's chunk: steps t1 down to max(t1 - GAE_CL + 1, 0)
        float r[GAE_CL], v[GAE_CL], nt[GAE_CL];
        // value / non-terminal flag of step t1 + 1
        float nv = 0.f, nnt = 0.f;
        if (t1 >= 0) {
            if (t1 == T - 1) { nv = last_val[ii]; nnt = 1.0f - (float)last_done[ii]; }
            else { const size_t o1 = (size_t)(t1 + 1) * N + ii; nv = val[o1]; nnt = 1.0f - (float)ep_start[o1]; }
        }
#pragma unroll
        for (int s = 0; s < GAE_CL; s++) {
            const int t = t1 - s;
            const bool ok = t >= 0;
            const size_t o = (size_t)(ok ? t : 0) * N + ii;
            r[s] = ok ? rew[o] : 0.f; v[s] = ok ? val[o] : 0.f; nt[s] = ok ? 1.0f - (float)ep_start[o] : 0.f;
        }
        float last = 0.f, b = 1.f, loc[GAE_CL], Bc[GAE_CL];
#pragma unroll
        for (int s = 0; s < GAE_CL; s++) {
            if (t1 - s >= 0) {
                const float delta = r[s] + gamma * nv * nnt - v[s];
                const float c = gl * nnt;
                last = delta + c * last;
                b *= c;
                nnt = nt[s]; nv = v[s];
            }
            loc[s] = last; Bc[s] = b;
        }
        ab[k][wi] = make_float2(last, b);                               // a chunk without steps is the identity map (0, 1)
        __syncthreads();
        // advantage entering this chunk: compose the later chunks from the latest on, starting from the previous pass
        float ain = carry[wi];
        for (int kk = 0; kk < k; kk++) { const float2 m = ab[kk][wi]; ain = m.x + m.y * ain; }
        if (valid) {
#pragma unroll
            for (int s = 0; s < GAE_CL; s++) {
                const int t = t1 - s;
                if (t >= 0) {
                    const size_t o = (size_t)t * N + i;
                    const float a = loc[s] + Bc[s] * ain;
                    adv[o] = a;
                    ret[o] = a + v[s];
                }
            }
        }
        __syncthreads();
        if (k == GAE_CH - 1) carry[wi] = last + b * ain;               // advantage of the earliest step of this pass
        __syncthreads();
    }
}

// sums for the advantage normalisation, deterministic: every block leaves its partial sums in the workspace, the last
// block to arrive adds them up in block order (no floating-point atomics: the same bits on every run)
constexpr int ADV_MAXBLOCKS = 512;
__global__ __launch_bounds__(256) void k_adv_stats(const float* __restrict__ a, long long n, double* out3, double* work, unsigned* arrive) {
    __shared__ double sh[4];
    __shared__ bool is_last;
    double s = 0, s2 = 0;
    const long long n4 = n / 4, stride = (long long)gridDim.x * blockDim.x;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = a4[i];
        s += (double)x.x + (double)x.y + (double)x.z + (double)x.w;
        s2 += (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z + (double)x.w * x.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) { const double x = a[4 * n4 + threadIdx.x]; s += x; s2 += x * x; }
    s = block_sum(s, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { work[2 * blockIdx.x] = s; work[2 * blockIdx.x + 1] = s2; }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(arrive, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    double S = 0, S2 = 0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x) { S += __builtin_nontemporal_load(&work[2 * b]); S2 += __builtin_nontemporal_load(&work[2 * b + 1]); }
    S = block_sum(S, sh);
    S2 = block_sum(S2, sh);
    if (threadIdx.x == 0) { out3[0] = S; out3[1] = S2; out3[2] = (double)n; *arrive = 0; }
}
__global__ void k_adv_normalize(float* a, long long n, const double* s3) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double cnt = s3[2], mean = s3[0] / cnt;
    double var = (s3[1] - cnt * mean * mean) / (cnt - 1);      // torch.std: Bessel corrected
    if (var < 0) var = 0;
    a[i] = (float)(((double)a[i] - mean) / (sqrt(var) + 1e-8));
}


// ---- the "blocked" order of the moment reduction (flags bit 32 of dl_vecnormalize_step) ----------------------------------------
// One canonical summation order that a single launch AND the persistent rollout kernel (k_rollout_persistent: one workgroup per sixteen
// walkers, a grid-wide exchange per control step) can both follow, so that the two produce the same bits:
//   rows are cut into blocks of 16 (block b = rows 16 b ..), blocks into <= 8 contiguous groups of gsize = ceil(nblk / 8) blocks;
//   S_b   = the block's shifted sums, rows in order          (s += d; ss = fma(d, d, ss); d = x - K, K = the mean before the update)
//   X_g   = sum of S_b over the group's blocks in order,  total = sum of X_g over the groups in order;  then the Chan merge below.
// The discounted returns are column D: ret = fma(ret, gamma, rew) is advanced first.
struct VnBlk { int nblk, gsize, ngrp; };
__host__ __device__ inline VnBlk vn_blk(int B) { VnBlk v; v.nblk = (B + 15) / 16; v.gsize = (v.nblk + 7) / 8; v.ngrp = (v.nblk + v.gsize - 1) / v.gsize; return v; }
// RunningMeanStd.update_from_moments for one column from the shifted sums (S, SS) of a batch of B rows
__device__ __forceinline__ void vn_chan_merge_d(double& mean, double& var, double cnt, double S, double SS, double B) {
#pragma clang fp contract(off)
    const double K = mean, sb = S / B, bm = K + sb, bv = SS / B - sb * sb;
    const double tot = cnt + B, delta = bm - K;
    const double M2 = var * cnt + bv * B + delta * delta * cnt * B / tot;
    mean = K + delta * B / tot;
    var = M2 / tot;
}
__device__ __forceinline__ void vn_chan_merge(double& mean, double& var, double cnt, double S, double SS, int B) { vn_chan_merge_d(mean, var, cnt, S, SS, (double)B); }
// shifted sums of column k over rows [r0, r1) of x[B, D] (k < D) or of the advanced returns (k == D; also stores them)
__device__ __forceinline__ void vn_block_sums(const float* __restrict__ x, const float* __restrict__ rew, double* ret, int D, int k, int r0, int r1, double K, double gamma,
                                              double& s_out, double& ss_out) {
    double s = 0, ss = 0;
    // a full block (every block but a ragged last one): sixteen unconditional loads at constant offsets from one address, all in flight at once
    // (predicated, every load sits in its own branch and waits for its own round trip: measured 10 k cycles per step), then the sums in row order.
    // A ragged block takes the same sums in the same order from a plain loop.
    if (r1 - r0 == 16) {
        if (k < D) {
            const float* xb = x + (size_t)r0 * D + k;
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; r++) a[r] = xb[r * D];
#pragma unroll
            for (int r = 0; r < 16; r++) { const double d = (double)a[r] - K; s = s + d; ss = fma(d, d, ss); }
        } else {
            double o[16]; float w[16];
#pragma unroll
            for (int r = 0; r < 16; r++) { o[r] = ret[r0 + r]; w[r] = rew[r0 + r]; }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const double rn = fma(o[r], gamma, (double)w[r]);
                ret[r0 + r] = rn;
                const double d = rn - K;
                s = s + d; ss = fma(d, d, ss);
            }
        }
    } else {
#pragma unroll 1
        for (int r = r0; r < r1; r++) {
            double d;
            if (k < D) d = (double)x[(size_t)r * D + k] - K;
            else { const double rn = fma(ret[r], gamma, (double)rew[r]); ret[r] = rn; d = rn - K; }
            s = s + d; ss = fma(d, d, ss);
        }
    }
    s_out = s; ss_out = ss;
}
// single launch, one workgroup: a chunk of 32 blocks at a time (block sums in parallel -> LDS -> sequential accumulation per column)
__global__ __launch_bounds__(1024) void k_vn_reduce_blk(const float* __restrict__ x, const float* __restrict__ rew, double* mean, double* var, const double* count,
                                                        double* ret, double* ret_mean, double* ret_var, const double* ret_count, int B, int D, double gamma, int flags) {
    constexpr int CH = 32;
    __shared__ double sh[CH][2][32];           // [block of the chunk][s / ss][column] (D + 1 <= 32 columns per pass of 32)
    const int t = threadIdx.x, W = D + 1;
    const VnBlk vb = vn_blk(B);
    for (int k0 = 0; k0 < W; k0 += 32) {       // column passes (D + 1 <= 32: one pass for the straight walker; 48 columns: two)
        const int kl = t & 31, bl = t >> 5, k = k0 + kl;
        const bool colok = k < W && ((k < D) ? (flags & 1) != 0 : (flags & 4) != 0);
        const double K = colok ? (k < D ? mean[k] : *ret_mean) : 0.0;
        double tot_s = 0, tot_ss = 0;
        for (int g = 0; g < vb.ngrp; g++) {
            const int b0 = g * vb.gsize, b1 = b0 + vb.gsize < vb.nblk ? b0 + vb.gsize : vb.nblk;
            double xs = 0, xss = 0;
            for (int c0 = b0; c0 < b1; c0 += CH) {
                const int b = c0 + bl;
                if (colok && b < b1) vn_block_sums(x, rew, ret, D, k, 16 * b, 16 * b + 16 < B ? 16 * b + 16 : B, K, gamma, sh[bl][0][kl], sh[bl][1][kl]);
                __syncthreads();
                if (bl == 0 && colok) { const int nb = b1 - c0 < CH ? b1 - c0 : CH; for (int i = 0; i < nb; i++) { xs += sh[i][0][kl]; xss += sh[i][1][kl]; } }
                __syncthreads();
            }
            tot_s += xs; tot_ss += xss;
        }
        if (bl == 0 && colok) {
            double m = K, v = k < D ? var[k] : *ret_var;
            vn_chan_merge(m, v, k < D ? *count : *ret_count, tot_s, tot_ss, B);
            if (k < D) { mean[k] = m; var[k] = v; } else { *ret_mean = m; *ret_var = v; }
        }
    }
}


// ---- exact per-step moments across ranks (collective C3 in its exact form): the local half of the reduction and the merge are separate calls,
// the caller all-reduces the 2 (D + 1) sums in between.  k_vn_sums_blk: this rank's shifted sums in the blocked order (one workgroup; advances the
// discounted returns); k_vn_merge_sums: RunningMeanStd.update_from_moments with the GLOBAL batch size, counts advanced here.
__global__ __launch_bounds__(1024) void k_vn_sums_blk(const float* __restrict__ x, const float* __restrict__ rew, const double* mean, double* ret, const double* ret_mean,
                                                      int B, int D, double gamma, int flags, double* sums) {
    constexpr int CH = 32;
    __shared__ double sh[CH][2][32];
    const int t = threadIdx.x, W = D + 1;
    const VnBlk vb = vn_blk(B);
    for (int k0 = 0; k0 < W; k0 += 32) {
        const int kl = t & 31, bl = t >> 5, k = k0 + kl;
        const bool colok = k < W && ((k < D) ? (flags & 1) != 0 : (flags & 4) != 0);
        const double K = colok ? (k < D ? mean[k] : *ret_mean) : 0.0;
        double tot_s = 0, tot_ss = 0;
        for (int g = 0; g < vb.ngrp; g++) {
            const int b0 = g * vb.gsize, b1 = b0 + vb.gsize < vb.nblk ? b0 + vb.gsize : vb.nblk;
            double xs = 0, xss = 0;
            for (int c0 = b0; c0 < b1; c0 += CH) {
                const int b = c0 + bl;
                if (colok && b < b1) vn_block_sums(x, rew, ret, D, k, 16 * b, 16 * b + 16 < B ? 16 * b + 16 : B, K, gamma, sh[bl][0][kl], sh[bl][1][kl]);
                __syncthreads();
                if (bl == 0 && colok) { const int nb = b1 - c0 < CH ? b1 - c0 : CH; for (int i = 0; i < nb; i++) { xs += sh[i][0][kl]; xss += sh[i][1][kl]; } }
                __syncthreads();
            }
            tot_s += xs; tot_ss += xss;
        }
        if (bl == 0 && k < W) { sums[2 * k] = colok ? tot_s : 0.0; sums[2 * k + 1] = colok ? tot_ss : 0.0; }
    }
}
__global__ __launch_bounds__(128) void k_vn_merge_sums(const double* __restrict__ sums, double* mean, double* var, double* count, double* ret_mean, double* ret_var, double* ret_count,
                                                       long long B, int D, int flags) {
    const int k = threadIdx.x, W = D + 1;
    const bool mine = k < W && (k < D ? (flags & 1) != 0 : (flags & 4) != 0);
    double cnt = 0;
    if (mine) cnt = k < D ? *count : *ret_count;
    __syncthreads();
    if (!mine) return;
    double* mp = k < D ? mean + k : ret_mean;
    double* vp = k < D ? var + k : ret_var;
    double m = *mp, v = *vp;
    vn_chan_merge_d(m, v, cnt, sums[2 * k], sums[2 * k + 1], (double)B);
    *mp = m; *vp = v;
    if (k == 0) *count = cnt + (double)B;
    if (k == D) *ret_count = cnt + (double)B;
}

// ---- SB3 collect_rollouts as ONE launch (dl_collect_rollouts, DL_ROLLOUT_PERSISTENT) ---------------------------------------------
// A persistent workgroup of eight waves owns sixteen walkers for all T control steps of the rollout -- the split workgroup of the step
// kernel (4 dynamics + 4 constraint waves) and the policy workgroup (8 waves x 16 rows) have the same shape, and one such workgroup fits
// per CU (143 KB of LDS), so a grid of <= one workgroup per CU is co-resident by construction.  Per control step:
//   P  policy forward for the workgroup's own 16 rows (pol_forward_rows: MFMA), with the normalisation of the last step's raw outputs folded
//      into its input stage exactly as in dl_rollout_policy; the moments live in LDS, replicated per workgroup;
//   E  MimicEnv.step for the 16 walkers (g_wave_env_step / g_constraint_server, the code of k_env_step_g16_split);
//   R  VecNormalize's moment update: the workgroup's block sums (blocked order, above) go to HBM, ONE grid-wide exchange -- group
//      counter -> the group's last arriver adds the group's blocks -> top counter -> everybody reads the <= 8 group sums and performs the
//      same merge -- and every workgroup holds the moments SB3 would have after this step.  Exact SB3 semantics.
// per_rollout != 0 (DL_ROLLOUT_MOMENTS_PER_ROLLOUT, opt-in): the moments are frozen at their start-of-rollout values, every workgroup
// accumulates its shifted sums over the whole rollout, nothing is exchanged during the rollout (workgroups run free: the launch lasts as
// long as the slowest SUM, not the sum of every step's slowest workgroup) and k_vn_merge_rollout performs one exact Chan merge of all
// T x N samples afterwards -- the relaxation collective C3 already applies across ranks.
// Hand-offs follow /opt/skills/guides/cdna_hip_programming.md guideline 16 in its fence-free form: every exchanged word is an 8-byte agent-scope
// access on BOTH sides (stores write through, loads bypass L1), every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup's one lane
// bumps a relaxed agent-scope counter; consumers poll that counter relaxed, from one lane.  Counters are
// monotonic over the steps of a launch (epoch = step + 1) and zeroed by the host before every launch.  Every poll is bounded: a timeout
// sets the handle's fault word (DL_FAULT_GRID_TIMEOUT) and ends the workgroup.
struct RolloutP {
    dl_policy_params pol;
    uint64_t seed, counter0;
    double *obs_mean, *obs_var, *obs_count, *ret, *ret_mean, *ret_var, *ret_count;
    double gamma, eps, clip_obs, clip_rew;
    float *observations, *actions, *values, *log_probs, *rewards, *next_obs, *raw_obs, *raw_rew;
    uint8_t *episode_starts, *next_done;
    double *partial, *xpart;      // [nblk][W][2], [2][8][W][2] (group sums, double-buffered by step parity: a group may not overwrite what another group's workgroups still read)
    unsigned* sync;               // group counters at [16 g], top counter at [128]
    PolPacked pk;                 // the policy's weights in k-chunk-major order (k_pack_policy), packed by the host before the launch
    long long* prof;              // diagnostics (DL_EXP_ROLLOUT_PROF builds): [nblk][4] shader-clock cycles in P, E, R (sums + exchange), waiting in the exchange
    int32_t index_base, flags, T, per_rollout, spin_grid, kblocks, deterministic;
};
constexpr int RP_SYNC_WORDS = 160;
constexpr int RP_PROF_STEPS = 512;        // DL_EXP_ROLLOUT_PROF builds: control steps with per-step section records
constexpr int RP_MAX_KBLOCKS = 8;          // blocks of sixteen walkers per workgroup: <= 128 walkers per CU, 32768 on 256 CUs
constexpr int RP_WS = 64;                  // lanes per block in the moment sums (a column per lane: OBS + 1 <= 64)
// blocks per workgroup a walker type allows: the 19-dof walker's sixteen regions leave 3 KB of the CU's LDS -- room for the moments and ONE block's rollout sums
template <typename TP> constexpr int rp_kblocks() { return GD<TP>::NX > 0 ? 1 : RP_MAX_KBLOCKS; }
template <typename TP> constexpr size_t rollout_lds_extra() { return (size_t)(2 * (TP::OBS + 1) + 2 + rp_kblocks<TP>() * 2 * (TP::OBS + 1)) * sizeof(double) + 64; }

// All arguments travel as ONE by-value struct: the kernel reads them through the kernarg segment pointer, made opaque at the start of every
// phase of every control step.  Passed as separate by-value parameters the ~150 uniform words (reference table, state arrays, rollout-buffer
// pointers, policy parameters) are values the compiler keeps alive in SGPRs across the WHOLE step loop -- policy phase, env phase and
// exchange -- and pays for with v_writelane / v_readlane spills inside the solver loops (measured: 471 SGPR + 188 VGPR spills, 40 us per
// control step); loaded per phase they live exactly as long as the phase.
template <typename TP> struct RolloutArgs {
    const GModel<float, TP>* gm;
    DevCfg<float> c;
    DevState<float> st;
    RolloutP a;
    int32_t eval_mode;
};
// MULTI: the instantiation for workgroups that own several blocks (more than sixteen walkers per CU): the policy phase takes TWO blocks per pass (32 rows share
// every weight register set, dl_policy.hpp RB), the env phase one block after the other.  A kernel of its own so that each stays under the instruction cache.
template <typename TP, bool MULTI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_rollout_persistent(const RolloutArgs<TP> args_by_value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using T = float;
    using Sp = GSplit<TP>;
    using Args = RolloutArgs<TP>;
    constexpr int D = TP::OBS, W = D + 1, NU = TP::NU;
    constexpr size_t ENV_LDS = (size_t)4 * GW * Sp::TOTAL * sizeof(T);
    constexpr int RBK = MULTI ? 2 : 1;          // blocks per pass of the policy phase
    static_assert(pol_lds_bytes_whole(8, 512, RBK) <= ENV_LDS, "the policy's LDS aliases the walkers' regions between two env steps");
    // the first (only) explicit argument sits at offset 0 of the kernarg segment (HSA ABI)
    const DL_CONST Args* const ap0 = (const DL_CONST Args*)__builtin_amdgcn_kernarg_segment_ptr();
    auto args = [&]() { const DL_CONST Args* q = ap0; DL_SPIN(q); return q; };       // a fresh, opaque view: loads through it are not merged with earlier ones
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, role = wave >> 2, slot = wave & 3;      // (wave: a scalar, so that what derives from it -- role, slot, the wave's block -- is too)
    const int gslot = role == 0 ? slot : ((slot + DL_SPLIT_PAIR_OFFSET) & 3);
    DL_LDS T* base = (DL_LDS T*)smem + (size_t)gslot * GW * Sp::TOTAL;
    double* vm = (double*)(smem + ENV_LDS);                 // mean[W] (column D: the returns'), var[W], count, ret_count
    double* accm = vm + 2 * W + 2;                          // per_rollout: [kb][W][2] the blocks' column sums over the whole rollout (in LDS: nothing of the moment update lives in registers across the env phase)
    constexpr int KBMAX = rp_kblocks<TP>();
    static_assert(!MULTI || KBMAX > 1, "several blocks per workgroup: not for this walker");
    int* shf = (int*)(accm + KBMAX * 2 * W);       // [0] group-last flags, [1] exchange ok
    static_assert(KBMAX * RP_WS <= 512 && W <= RP_WS, "a group of RP_WS lanes per block");
    const int wgi = g_block_of_workgroup(blockIdx.x, gridDim.x);
    int n, nT, flags, per_rollout, kb;
    {
        const DL_CONST Args* p = args();
        n = p->st.n; nT = p->a.T; flags = p->a.flags; per_rollout = p->a.per_rollout; kb = p->a.kblocks;
        if (tid < D) { vm[tid] = p->a.obs_mean[tid]; vm[W + tid] = p->a.obs_var[tid]; }
        if (tid == D) { vm[D] = *p->a.ret_mean; vm[W + D] = *p->a.ret_var; vm[2 * W] = *p->a.obs_count; vm[2 * W + 1] = *p->a.ret_count; }
        for (int i = tid; i < KBMAX * 2 * W; i += 512) accm[i] = 0.0;
    }
    // the workgroup owns kb consecutive blocks of sixteen walkers (one on <= 4096 walkers; more walkers than 16 x CUs: the grid stays co-resident and a
    // workgroup takes its blocks one after the other through the policy and env phases of a step)
    const VnBlk vb = vn_blk(n);
    const int nblk = vb.nblk;
    const int b0 = wgi * kb, b1 = b0 + kb < nblk ? b0 + kb : nblk;
    const bool upd_obs = (flags & 1) != 0, upd_ret = (flags & 4) != 0, exchange = (upd_obs || upd_ret) && !per_rollout;
#ifdef DL_EXP_ROLLOUT_PROF
    long long prof_acc[4] = {0, 0, 0, 0}, prof_t = DL_CLOCK();
    int t_prof = 0;
    long long* const pstep = args()->a.prof ? args()->a.prof + (size_t)nblk * 4 * 11 : nullptr;          // [T][workgroups][4]: the same sections per control step (tools/diag_rollout_floor.py)
    long long prof_rt = (long long)__builtin_amdgcn_s_memrealtime();          // the per-step records use the constant 100 MHz counter: the XCDs' shader clocks differ, and these records are compared ACROSS workgroups
#define DL_RP_TICK(k) do { const long long now_ = DL_CLOCK(); prof_acc[k] += now_ - prof_t; prof_t = now_; \
        if (pstep) { const long long rt_ = (long long)__builtin_amdgcn_s_memrealtime(); if (tid == 0) pstep[((size_t)t_prof * gridDim.x + wgi) * 4 + (k)] += rt_ - prof_rt; prof_rt = rt_; } } while (0)
#else
#define DL_RP_TICK(k) ((void)0)
#endif
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < nT; t++) {
#ifdef DL_EXP_ROLLOUT_PROF
        t_prof = t;
#endif
        DL_RP_TICK(2);
#pragma unroll 1
        for (int blk_i = b0; blk_i < b1; blk_i += RBK) {
        int blk = blk_i;
        DL_SPIN(blk);             // (opaque per block: +0.7 % -- what is derived from the block index is formed in the phase that uses it)
        const int row0 = blk * 16;
        // ---- P: actions, values, log-probs of step t (and observations[t], rewards[t - 1] from the raw outputs of step t - 1)
        {
            const DL_CONST Args* p = args();
            const RolloutP a = p->a;
            int tid_t = tid;
            DL_VPIN(tid_t);           // per-step opaque: the policy's per-lane index arithmetic is not kept alive across the env phase
            PolVnFuse vf{};
            if (t > 0) {
                vf.raw_obs = a.raw_obs; vf.raw_rew = a.raw_rew; vf.done = a.episode_starts + (size_t)t * n;
                vf.mean = vm; vf.var = vm + W; vf.count = nullptr; vf.ret = a.ret; vf.ret_var = vm + W + D; vf.ret_count = nullptr;
                vf.obs_out = a.observations + (size_t)t * n * D; vf.rew_out = a.rewards + (size_t)(t - 1) * n;
                vf.eps = a.eps; vf.clip_obs = a.clip_obs; vf.clip_rew = a.clip_rew; vf.flags = flags;
            }
            const int nlim = b1 * 16 < n ? b1 * 16 : n;          // rows of THIS workgroup's blocks only (a pass of two blocks may reach beyond its last one)
#define DL_RP_ROWS(NTW) pol_forward_rows<NTW, 8, true, true, RBK>(a.pol, a.observations + (size_t)t * n * D, nlim, nullptr, a.seed, a.counter0 + (uint64_t)t, a.index_base, a.deterministic, \
                                   a.actions + (size_t)t * n * NU, a.values + (size_t)t * n, a.log_probs + (size_t)t * n, vf, (float*)smem, row0, false, tid_t, a.pk)
            const int hid = a.pol.hidden;          // (uniform; checked by the host: 512 / 256 / 128 = eight waves x 4 / 2 / 1 tiles -- drloco/config/hypers.py:98-99 makes the hidden sizes a config)
            if (hid == 512) DL_RP_ROWS(4); else if (hid == 256) DL_RP_ROWS(2); else DL_RP_ROWS(1);
#undef DL_RP_ROWS
        }
        __syncthreads();          // the actions of the pass's rows are in memory (workgroup scope); the policy's LDS is free again
        DL_RP_TICK(0);
#pragma unroll 1
        for (int sub = 0; sub < RBK; sub++) {
        if (sub > 0) { blk += 1; if (blk >= b1) break; }
        // ---- E: one control step of the sixteen walkers
        if (lane == 0) {
            volatile DL_LDS int* f = (volatile DL_LDS int*)(base + Sp::MB);
            f[Sp::MB_CMDSEQ] = 0; f[Sp::MB_DONESEQ] = 0; f[Sp::MB_CMD] = 1; f[Sp::MB_MOK] = 0; f[Sp::MB_MFREE] = 0; f[Sp::MB_PRE] = -1;
        }
        __syncthreads();
        {
            const DL_CONST Args* p = args();
            DevState<T> st = p->st;
            st.push_step0 += t;
            const DevCfg<T> c = p->c;
            uint8_t* done = t + 1 == nT ? p->a.next_done : p->a.episode_starts + (size_t)(t + 1) * n;
            const int wblock = blk * 4 + gslot;
            int lane_t = lane;
            DL_VPIN(lane_t);          // per-step opaque (as above): lane topology, lane records and pinned constants are this phase's only
            if (role == 0)
#if defined(DL_EXP_ROLLOUT_PROF) && DL_EXP_ROLLOUT_PROF != 2        // (2: the per-step phase records only, the env step itself uninstrumented -- tools/diag_rollout_floor.py)  per-section cycles of the LAST control step's env phase, per dynamics wave: prof[nblk * 4 + k * (nblk * 4) + wave block], k as in g_wave_env_step<TIMED>
                g_wave_env_step<T, TP, true, true>(lane_t, wblock, wblock, nblk * 4, base, p->gm, c, st, p->a.actions + (size_t)t * n * NU, p->a.raw_obs, p->a.raw_rew, done, (float*)nullptr, (float*)nullptr,
                                                   (const T*)nullptr, (const T*)nullptr, (const int32_t*)nullptr, (float*)nullptr, p->eval_mode, 1, p->a.prof + (size_t)nblk * 4);
#else
                g_wave_env_step<T, TP, false, true>(lane_t, wblock, 0, 1, base, p->gm, c, st, p->a.actions + (size_t)t * n * NU, p->a.raw_obs, p->a.raw_rew, done, (float*)nullptr, (float*)nullptr,
                                                    (const T*)nullptr, (const T*)nullptr, (const int32_t*)nullptr, (float*)nullptr, p->eval_mode, 1, nullptr);
#endif
            else
                g_constraint_server<T, TP>(lane_t, wblock, base, p->gm, st);
        }
        __syncthreads();          // raw observation / reward / done of the block's rows are in memory
        DL_RP_TICK(1);
        }   // the blocks of the pass
        }   // the workgroup's blocks
        // ---- R: VecNormalize's moment update
        int tid_r = tid;
        DL_VPIN(tid_r);               // per-step opaque (as in the other phases): the per-thread addresses of the exchange are recomputed every step, not hoisted out of the step loop and spilled
        int n_r = n;
        DL_SPIN(n_r);                 // (opaque: `(double)n` is otherwise a register pair hoisted out of the step loop and spilled; the other uniform words stay as they are -- all of them opaque measured 2 % slower)
        const int ngrp_r = vb.ngrp, gsize_r = vb.gsize;
        if (upd_obs || upd_ret) {     // the block sums: a group of WS lanes per block of the workgroup, a column per lane
            const DL_CONST Args* p = args();
            const int bi = tid_r / RP_WS, col = tid_r % RP_WS, blk = b0 + bi;
            const bool mine = col < W && blk < b1 && (col < D ? upd_obs : upd_ret);
            if (mine) {
                double s, ss;
                const int row0 = blk * 16, row1 = row0 + 16 < n_r ? row0 + 16 : n_r;
                vn_block_sums(p->a.raw_obs, p->a.raw_rew, p->a.ret, D, col, row0, row1, vm[col], p->a.gamma, s, ss);
                if (exchange) {       // 8-byte agent-scope stores: write-through, so that the hand-over needs no release fence (guideline 16, R1)
                    __hip_atomic_store(&p->a.partial[((size_t)blk * W + col) * 2], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&p->a.partial[((size_t)blk * W + col) * 2 + 1], ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains before the flag
                }
                else { accm[(bi * W + col) * 2] += s; accm[(bi * W + col) * 2 + 1] += ss; }
            }
        }
        if (exchange) {
            const DL_CONST Args* p = args();
            // every word of the exchange is an 8-byte agent-scope access on both sides (stores write through, loads bypass L1): the valid
            // fence-free form of the guide -- an agent-scope release would write back the whole L2's dirty lines (the step's raw outputs) first
            unsigned* sync = p->a.sync;
            double* partial = p->a.partial;
            double* xpart = p->a.xpart + (size_t)(t & 1) * 8 * W * 2;
            const int g_lo = b0 / gsize_r, g_hi = (b1 - 1) / gsize_r;          // the groups this workgroup's blocks belong to (one; two where its range crosses a group boundary)
            __syncthreads();
            if (tid_r == 0) {         // a group's counter counts BLOCKS: complete when all of the group's blocks of this step have been delivered
                int last = 0;
                for (int g = g_lo; g <= g_hi; g++) {
                    const int gb0 = g * gsize_r, gb1 = gb0 + gsize_r < nblk ? gb0 + gsize_r : nblk;
                    const unsigned mine_n = (unsigned)((b1 < gb1 ? b1 : gb1) - (b0 > gb0 ? b0 : gb0));
                    const unsigned old = __hip_atomic_fetch_add(sync + 16 * g, mine_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + mine_n == (unsigned)(gb1 - gb0) * (unsigned)(t + 1)) last |= 1 << (g - g_lo);
                }
                shf[0] = last;
            }
            __syncthreads();
            const int last = shf[0];
            for (int g = g_lo; g <= g_hi; g++) {
                if (!((last >> (g - g_lo)) & 1)) continue;          // (uniform)
                // the deliverer of a group's last block adds the group's block sums, in block order, 32 blocks at a time: every load of a chunk in flight at once
                // (eight lanes per column pair fetch four blocks each into LDS -- the idle env regions --, then the column's lane adds them in order)
                double* gs = (double*)smem;                         // [32][2 W]
                const int gb0 = g * gsize_r, gb1 = gb0 + gsize_r < nblk ? gb0 + gsize_r : nblk, gn = gb1 - gb0;
                double x = 0;
                for (int c0 = 0; c0 < gn; c0 += 32) {
                    for (int e = tid_r; e < 8 * 2 * W; e += 512) {          // (one trip for the straight walker: 8 x 60 <= 512 lanes; two for the 19-dof walker's 8 x 96)
                        const int col = e % (2 * W), part = e / (2 * W);
                        double v4[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) { const int b = c0 + part * 4 + i; v4[i] = b < gn ? __hip_atomic_load(&partial[(size_t)(gb0 + b) * W * 2 + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; }
#pragma unroll
                        for (int i = 0; i < 4; i++) gs[(part * 4 + i) * (2 * W) + col] = v4[i];
                    }
                    __syncthreads();
                    if (tid_r < 2 * W) { const int cn = gn - c0 < 32 ? gn - c0 : 32; for (int b = 0; b < cn; b++) x += gs[b * (2 * W) + tid_r]; }
                    __syncthreads();
                }
                if (tid_r < 2 * W) {
                    __hip_atomic_store(&xpart[(size_t)g * W * 2 + tid_r], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                if (tid_r == 0) __hip_atomic_fetch_add(sync + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            DL_RP_TICK(2);
            if (tid_r == 0) {
                const unsigned want = (unsigned)ngrp_r * (unsigned)(t + 1);
                const int budget = p->a.spin_grid;
                bool ok = false;
                for (int it = 0; it < budget; it++) {
                    if (__hip_atomic_load(sync + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!ok && p->st.fault) DL_FAULT_OR(p->st.fault, DL_FAULT_GRID_TIMEOUT);
                shf[1] = ok ? 1 : 0;
            }
            __syncthreads();
            DL_RP_TICK(3);
            if (!shf[1]) return;          // (uniform) the grid never completed this step: fault word set, nothing further is written
            if (tid_r < W && (tid_r < D ? upd_obs : upd_ret)) {
                double S = 0, SS = 0, xs[8], xq[8];
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    xs[g] = g < ngrp_r ? __hip_atomic_load(&xpart[((size_t)g * W + tid_r) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                    xq[g] = g < ngrp_r ? __hip_atomic_load(&xpart[((size_t)g * W + tid_r) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                }
#pragma unroll
                for (int g = 0; g < 8; g++) if (g < ngrp_r) { S += xs[g]; SS += xq[g]; }
                double m = vm[tid_r], v = vm[W + tid_r];
                vn_chan_merge(m, v, tid_r < D ? vm[2 * W] : vm[2 * W + 1], S, SS, n_r);
                vm[tid_r] = m; vm[W + tid_r] = v;
            }
            __syncthreads();
            if (tid_r == 0) { if (upd_obs) vm[2 * W] += (double)n_r; if (upd_ret) vm[2 * W + 1] += (double)n_r; }
        }
        __syncthreads();
    }
    // ---- VecNormalize of the last step's outputs (k_vn_apply's work for the workgroup's rows): next_obs, rewards[T - 1], ret
    const DL_CONST Args* p = args();
#ifdef DL_EXP_ROLLOUT_PROF
    if (tid == 0 && p->a.prof) for (int k = 0; k < 4; k++) p->a.prof[(size_t)wgi * 4 + k] = prof_acc[k];
#endif
    const RolloutP a = p->a;
    const int row0 = b0 * 16, row1 = b1 * 16 < n ? b1 * 16 : n;          // all rows of the workgroup's blocks
    for (int idx = tid; idx < (row1 - row0) * D; idx += 512) {
        const size_t e = (size_t)row0 * D + idx;
        const int k = idx % D;
        a.next_obs[e] = (flags & 2) ? vn_norm_obs(a.raw_obs[e], vm[k], vm[W + k], a.eps, a.clip_obs) : a.raw_obs[e];
    }
    if (tid < row1 - row0) {
        const int r = row0 + tid;
        a.rewards[(size_t)(nT - 1) * n + r] = (flags & 8) ? vn_norm_rew(a.raw_rew[r], vm[W + D], a.eps, a.clip_rew) : a.raw_rew[r];
        if (upd_ret && a.next_done[r]) a.ret[r] = 0;
    }
    if (per_rollout) {            // the blocks' sums over the whole rollout, merged by k_vn_merge_rollout
        const int bi = tid / RP_WS, col = tid % RP_WS, blk = b0 + bi;
        if (col < W && blk < b1) { a.partial[((size_t)blk * W + col) * 2] = accm[(bi * W + col) * 2]; a.partial[((size_t)blk * W + col) * 2 + 1] = accm[(bi * W + col) * 2 + 1]; }
    } else if (wgi == 0) {        // every workgroup holds the same moments: one of them hands them back
        if (tid < D) { a.obs_mean[tid] = vm[tid]; a.obs_var[tid] = vm[W + tid]; }
        if (tid == D) { *a.ret_mean = vm[D]; *a.ret_var = vm[W + D]; *a.obs_count = vm[2 * W]; *a.ret_count = vm[2 * W + 1]; }
    }
}
// ---- per-rollout moments (DL_ROLLOUT_MOMENTS_PER_ROLLOUT), every wave PAIR on its own.  With the moments frozen nothing couples the walkers of a rollout, and the only
// reason the four pairs of a workgroup met at every control step was the policy's 16-row tile (17 % of the env phase went into waiting for the slowest pair).  Here a pair
// evaluates the policy of its own four walkers (pol_forward_pair: v_mfma_f32_4x4x1, the bits of dl_policy_forward), steps them, adds their samples to its own sums and
// goes on: a pair takes its four walkers of a block through all T steps, then those of the workgroup's next block.  The two waves of a pair meet through two counters in
// their mailbox (pair_sync); no s_barrier after the kernel's prologue.  The pair's shifted sums live in memory (partial[(4 block + pair)][W][2], zeroed by the host): k_vn_merge_rollout
// adds them in row order.
template <typename TP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_rollout_pairs(const RolloutArgs<TP> args_by_value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using T = float;
    using Sp = GSplit<TP>;
    using Args = RolloutArgs<TP>;
    constexpr int D = TP::OBS, W = D + 1, NU = TP::NU;
    constexpr size_t ENV_LDS = (size_t)4 * GW * Sp::TOTAL * sizeof(T);
    static_assert(POLP_WORDS <= 3 * Sp::TOTAL, "the pair's policy works in the regions of its walkers 1 .. 3 (walker 0's mailbox holds the pair's counters)");
    const DL_CONST Args* const ap0 = (const DL_CONST Args*)__builtin_amdgcn_kernarg_segment_ptr();
    auto args = [&]() { const DL_CONST Args* q = ap0; DL_SPIN(q); return q; };
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, role = wave >> 2, slot = wave & 3;
    const int gslot = role == 0 ? slot : ((slot + DL_SPLIT_PAIR_OFFSET) & 3);          // the pair: the dynamics wave and the partner wave of one group of four walkers
    DL_LDS T* base = (DL_LDS T*)smem + (size_t)gslot * GW * Sp::TOTAL;
    double* vm = (double*)(smem + ENV_LDS);                 // mean[W] (column D: the returns'), var[W]: frozen for the whole rollout
    const int wgi = g_block_of_workgroup(blockIdx.x, gridDim.x);
    int n, nT, flags, kb;
    {
        const DL_CONST Args* p = args();
        n = p->st.n; nT = p->a.T; flags = p->a.flags; kb = p->a.kblocks;
        if (tid < D) { vm[tid] = p->a.obs_mean[tid]; vm[W + tid] = p->a.obs_var[tid]; }
        if (tid == D) { vm[D] = *p->a.ret_mean; vm[W + D] = *p->a.ret_var; }
        if (lane == 0 && role == 0) { volatile DL_LDS int* ps = (volatile DL_LDS int*)(base + Sp::MB); ps[Sp::MB_PAIR0] = 0; ps[Sp::MB_PAIR1] = 0; }
    }
    const int nblk = (n + 15) / 16;
    const int b0 = wgi * kb, b1 = b0 + kb < nblk ? b0 + kb : nblk;
    const bool upd_obs = (flags & 1) != 0, upd_ret = (flags & 4) != 0;
    __syncthreads();
    int epoch = 0;
    int32_t* const fault_w = args()->st.fault;
    const int pair_spin = args()->a.spin_grid;          // polls before a wave gives its partner up: the budget of the persistent kernels' waits, shared with the grid exchange of k_rollout_persistent (dl_debug_set_grid_spin sets both: include/drloco_hip.h; 1 << 22 ~ seconds)
    // The two waves of the pair meet: everything either has written (LDS, memory) is visible to the other afterwards.  Bounded like every poll of the split form -- an
    // error, never a hung GPU -- and, like there, a wave never carries on with stale data: a wave whose partner did not arrive sets the handle's fault word
    // (DL_FAULT_SRV_TIMEOUT), the pair is dead from then on (every later pair_sync returns at once) and the wave LEAVES the kernel at the next phase boundary
    // (`if (!pair_alive) return` below; no s_barrier follows the prologue, so a wave may end early).  The rows of the rollout buffer the pair owned stay incomplete, as
    // a workgroup's do after DL_FAULT_GRID_TIMEOUT: the host must see the fault word before it reads the buffer (include/drloco_hip.h).
    bool pair_alive = true;
    auto pair_sync = [&]() {
        if (!pair_alive) return;
        volatile DL_LDS int* ps = (volatile DL_LDS int*)(base + Sp::MB);
        DL_WG_RELEASE();
        epoch += 1;
        if (lane == 0) ps[Sp::MB_PAIR0 + role] = epoch;
        int budget = pair_spin;
        while (DL_UNIFORM(ps[Sp::MB_PAIR1 - role]) < epoch && --budget > 0) __builtin_amdgcn_s_sleep(1);
        // the verdict comes from a LAST read of the partner's counter, not from the budget: a partner that arrives on the very last poll (or a budget of 0 with the
        // partner already there) has arrived
        if (DL_UNIFORM(ps[Sp::MB_PAIR1 - role]) < epoch) {
            pair_alive = false;
            if (fault_w && lane == 0) DL_FAULT_OR(fault_w, DL_FAULT_SRV_TIMEOUT);
        }
        DL_WG_ACQUIRE();
    };
#pragma unroll 1
    for (int blk_i = b0; blk_i < b1; blk_i++) {
    const int row0 = blk_i * 16 + gslot * 4, row1 = row0 + 4 < n ? row0 + 4 : n;
#pragma unroll 1
    for (int t = 0; t < nT; t++) {
        if (row0 >= n) break;         // (uniform per pair) a pair beyond the last walker has nothing to do
        int blk = blk_i;
        DL_SPIN(blk);
        // ---- P: the policy of the pair's four rows (and observations[t], rewards[t - 1] from the raw outputs of step t - 1)
        {
            const DL_CONST Args* p = args();
            const RolloutP a = p->a;
            int lane_p = lane;
            DL_VPIN(lane_p);
            PolVnFuse vf{};
            if (t > 0) {
                vf.raw_obs = a.raw_obs; vf.raw_rew = a.raw_rew; vf.done = a.episode_starts + (size_t)t * n;
                vf.mean = vm; vf.var = vm + W; vf.count = nullptr; vf.ret = a.ret; vf.ret_var = vm + W + D; vf.ret_count = nullptr;
                vf.obs_out = a.observations + (size_t)t * n * D; vf.rew_out = a.rewards + (size_t)(t - 1) * n;
                vf.eps = a.eps; vf.clip_obs = a.clip_obs; vf.clip_rew = a.clip_rew; vf.flags = flags;
            }
#define DL_RP_PAIR(HH) pol_forward_pair<HH>(a.pol, a.observations + (size_t)t * n * D, n, nullptr, a.seed, a.counter0 + (uint64_t)t, a.index_base, a.deterministic, a.actions + (size_t)t * n * NU, \
                             a.values + (size_t)t * n, a.log_probs + (size_t)t * n, vf, (float*)(base + Sp::TOTAL), blk * 16 + gslot * 4, role, lane_p, a.pk, pair_sync)
            const int hid = a.pol.hidden;          // (uniform; checked by the host: 512 / 256 / 128 -- the reference's hidden sizes are a config, drloco/config/hypers.py:98-99)
            if (hid == 512) DL_RP_PAIR(512); else if (hid == 256) DL_RP_PAIR(256); else DL_RP_PAIR(128);
#undef DL_RP_PAIR
        }
        // ---- E: one control step of the four walkers
        if (lane == 0 && role == 0) {
            volatile DL_LDS int* f = (volatile DL_LDS int*)(base + Sp::MB);
            f[Sp::MB_CMDSEQ] = 0; f[Sp::MB_DONESEQ] = 0; f[Sp::MB_CMD] = 1; f[Sp::MB_MOK] = 0; f[Sp::MB_MFREE] = 0; f[Sp::MB_PRE] = -1;
        }
        pair_sync();                  // the actions are in memory, the policy's LDS is free, the mailbox is reset
        if (!pair_alive) return;
        {
            const DL_CONST Args* p = args();
            DevState<T> st = p->st;
            st.push_step0 += t;
            const DevCfg<T> c = p->c;
            uint8_t* done = t + 1 == nT ? p->a.next_done : p->a.episode_starts + (size_t)(t + 1) * n;
            const int wblock = blk * 4 + gslot;
            int lane_t = lane;
            DL_VPIN(lane_t);
            if (role == 0)
                g_wave_env_step<T, TP, false, true>(lane_t, wblock, 0, 1, base, p->gm, c, st, p->a.actions + (size_t)t * n * NU, p->a.raw_obs, p->a.raw_rew, done, (float*)nullptr, (float*)nullptr,
                                                    (const T*)nullptr, (const T*)nullptr, (const int32_t*)nullptr, (float*)nullptr, p->eval_mode, 1, nullptr);
            else
                g_constraint_server<T, TP>(lane_t, wblock, base, p->gm, st);
        }
        pair_sync();                  // raw observation / reward / done of the pair's rows are in memory
        if (!pair_alive) return;
        // ---- R: the pair's samples of this step join its shifted sums (and the discounted returns advance)
        if (role == 0 && (upd_obs || upd_ret)) {
            const DL_CONST Args* p = args();
            int col = lane;
            DL_VPIN(col);
            if (col < W && (col < D ? upd_obs : upd_ret)) {
                double s, ss;
                vn_block_sums(p->a.raw_obs, p->a.raw_rew, p->a.ret, D, col, row0, row1, vm[col], p->a.gamma, s, ss);
                double* acc = p->a.partial + ((size_t)(blk * 4 + gslot) * W + col) * 2;
                acc[0] += s; acc[1] += ss;
            }
        }
    }
    // ---- VecNormalize of the last step's outputs for the pair's rows: next_obs, rewards[T - 1], ret
    if (row0 < n) {
        const DL_CONST Args* p = args();
        const RolloutP a = p->a;
        const int t2 = role * 64 + lane;
        for (int idx = t2; idx < (row1 - row0) * D; idx += 128) {
            const size_t e = (size_t)row0 * D + idx;
            const int k = idx % D;
            a.next_obs[e] = (flags & 2) ? vn_norm_obs(a.raw_obs[e], vm[k], vm[W + k], a.eps, a.clip_obs) : a.raw_obs[e];
        }
        if (t2 < row1 - row0) {
            const int r = row0 + t2;
            a.rewards[(size_t)(nT - 1) * n + r] = (flags & 8) ? vn_norm_rew(a.raw_rew[r], vm[W + D], a.eps, a.clip_rew) : a.raw_rew[r];
            if (upd_ret && a.next_done[r]) a.ret[r] = 0;
        }
        pair_sync();                  // (the next block's first policy pass reuses the regions)
        if (!pair_alive) return;
    }
    }
}
// per_rollout: one exact Chan merge of the T x N samples of a rollout from the workgroups' shifted sums (groups, then blocks, in order)
__global__ __launch_bounds__(64) void k_vn_merge_rollout(const double* __restrict__ partial, double* mean, double* var, double* count, double* ret_mean, double* ret_var, double* ret_count,
                                                       int nblk, int D, long long samples, int flags) {
    const int k = threadIdx.x, W = D + 1;
    const bool mine = k < W && (k < D ? (flags & 1) != 0 : (flags & 4) != 0);
    double cnt = 0;
    if (mine) cnt = k < D ? *count : *ret_count;
    __syncthreads();
    if (!mine) return;
    double S = 0, SS = 0;
    for (int b = 0; b < nblk; b++) { S += partial[((size_t)b * W + k) * 2]; SS += partial[((size_t)b * W + k) * 2 + 1]; }
    double* mp = k < D ? mean + k : ret_mean;
    double* vp = k < D ? var + k : ret_var;
    double m = *mp, v = *vp;
    vn_chan_merge_d(m, v, cnt, S, SS, (double)samples);
    *mp = m; *vp = v;
    if (k == 0) *count = cnt + (double)samples;
    if (k == D) *ret_count = cnt + (double)samples;
}

// ------------------------------------------------------------------------------------------
// handle
static thread_local std::string g_err;
static int fail(int code, const std::string& what) { g_err = what; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(DL_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct dl_env_s {
    virtual ~dl_env_s() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); if (fault_host) (void)hipHostFree(fault_host); }
    // fault word: host-pinned, mapped into the device; a wave of a split workgroup that leaves a bounded poll by timeout ors its reason in
    // (dl_group.hpp DL_FAULT_*).  Sticky until dl_fault_clear: every entry point that launches or reads results returns DL_E_FAULT while set.
    int32_t* fault_host = nullptr;
    int32_t* fault_dev = nullptr;
    int fault_alloc() {
        if (hipHostMalloc((void**)&fault_host, sizeof(int32_t), hipHostMallocMapped) != hipSuccess) { fault_host = nullptr; return DL_E_NOMEM; }
        *fault_host = 0;
        if (hipHostGetDevicePointer((void**)&fault_dev, fault_host, 0) != hipSuccess) return DL_E_HIP;
        return DL_OK;
    }
    int fault_code() const { return fault_host ? __atomic_load_n(fault_host, __ATOMIC_RELAXED) : 0; }
    virtual void set_spin_limits(int dyn, int srv) = 0;
    virtual void set_grid_spin(int polls) = 0;
    int n = 0, device = 0, real_size = 4, eval_mode = 0, obs_dim = 0, act_dim = 0, variant = 0;
    virtual int init(const dl_model_desc&, const dl_refs_desc&, const dl_config&, int n, int device) = 0;
    virtual int reset(const uint8_t*, const int32_t*, const int32_t*, float*, hipStream_t) = 0;
    virtual int step(const float*, float*, float*, uint8_t*, float*, float*, hipStream_t) = 0;
    // up to `nsteps` control steps of a time-major action tape in one launch; returns the number of steps taken (>= 1) or a negative error
    virtual int steps_fixed(int nsteps, const float*, float*, float*, uint8_t*, hipStream_t) = 0;
    int prof_steps = 0, prof_last_steps = 0;   // control steps covered by the bracketed launches (since / at the last dl_profile_read)
    virtual int get_state(void*, void*, void*, int32_t*, double*, hipStream_t) = 0;
    virtual int set_state(const void*, const void*, const void*, const int32_t*, const double*, hipStream_t) = 0;
    virtual int ref_offsets(void* get, const void* set, hipStream_t) = 0;
    virtual int forward(const void*, void*, int32_t*, int32_t*, int32_t*, hipStream_t) = 0;
    virtual int snapshot(int word, double* out, hipStream_t) = 0;
    virtual int terminate_early(int32_t* flags, hipStream_t) = 0;
    virtual int randomize(const float* mass_scale, const float* floor_friction, const float* push, bool set_push, hipStream_t) = 0;
    virtual int push_schedule(const float* force, const int32_t* phase, int period, int duration, hipStream_t) = 0;
    virtual int inject(const void* q, const void* v, const int32_t* flags, const int32_t* rsi, hipStream_t) = 0;
    virtual int counters(int32_t* out, int clear, hipStream_t) = 0;
    virtual int capstate(float* out, hipStream_t) = 0;
    virtual int eval_iters(float* out, hipStream_t) = 0;
    virtual int last_ctrl(float* out, hipStream_t) = 0;
    virtual int set_split(int on) = 0;
    virtual int rollout_prof(long long* out, hipStream_t s) = 0;
    virtual int persistent_ok(int hidden, std::string* why) = 0;
    virtual int pack_policy(const dl_policy_params& pol, PolPacked* out, hipStream_t s) = 0;
    virtual int collect_persistent(const dl_policy_params& pol, uint64_t seed, uint64_t counter0, int32_t index_base, const dl_vecnorm_state& vn, int32_t T, float* observations,
                                   float* actions, float* values, float* log_probs, float* rewards, uint8_t* episode_starts, float* next_obs, uint8_t* next_done, float* raw_obs,
                                   float* raw_rew, int per_rollout, int deterministic, hipStream_t s) = 0;
    virtual int forward_timed(const void*, void*, long long*, hipStream_t) = 0;
    virtual int step_timed(const float*, float*, float*, uint8_t*, long long*, hipStream_t) = 0;
    // per-launch timing of the dominant kernel (k_env_step) with HIP events on the launch stream
    int prof = 0;                // 0 off, k > 0: bracket every k-th launch (events between kernels cost a few us of launch gap each)
    int prof_tick = 0;
    bool prof_open = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    // launch geometry of the last bracketed launch of the dominant kernel: threads of the grid, threads per workgroup, dynamic LDS bytes (dl_profile_launch_config:
    // what a committed PMC pass is compared with before its counters are replayed, next to the device code's hash)
    int launch_cfg[3] = {0, 0, 0};
    void note_launch(long long blocks, int block, size_t lds) { if (prof_open) { launch_cfg[0] = (int)(blocks * block); launch_cfg[1] = block; launch_cfg[2] = (int)lds; } }
    void prof_begin(hipStream_t s) {
        prof_open = false;
        if (!prof || (prof_tick++ % prof) != 0) return;
        prof_open = true;
        if (ev_used + 2 > ev.size()) { hipEvent_t a, b; if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return; ev.push_back(a); ev.push_back(b); }
        (void)hipEventRecord(ev[ev_used], s);
    }
    void prof_end(hipStream_t s) {
        if (!prof_open || ev_used + 2 > ev.size()) return;
        (void)hipEventRecord(ev[ev_used + 1], s);
        ev_used += 2;
    }
};

template <typename T, typename TP> struct EnvImpl final : dl_env_s {
    static constexpr int BLOCK = KernelGeom<T, TP>::BLOCK;
    static constexpr size_t LDS = (size_t)MemLayout<TP>::TOTAL * BLOCK * sizeof(T);
    DevModel<T, TP> m;          // host copy
    DevModel<T, TP>* md = nullptr;   // device copy (read through the constant address space by the kernels)
    DevCfg<T> c;
    DevState<T> st{};
    std::vector<void*> allocs;
    GModel<T, TP>* gmd = nullptr;    // table-driven model of the 16-lane kernels
    static constexpr size_t GLDS = (size_t)GW * GLds<TP>::TOTAL * sizeof(T);      // LDS of one wave (four walkers) of the 16-lane kernels
    static constexpr size_t SLDS_ = (size_t)4 * GW * GSplit<TP>::TOTAL * sizeof(T);
    static constexpr bool CAN_SPLIT = sizeof(T) == 4 && SLDS_ <= 160 * 1024;         // the split workgroup (step kernel): float32, sixteen walkers' regions within a CU's LDS
    static constexpr bool CAN_PERSIST = CAN_SPLIT && SLDS_ + rollout_lds_extra<TP>() <= 160 * 1024;          // the persistent rollout kernels: the split regions + the moments' block fit the CU's LDS
    static constexpr bool CAN_PERSIST_MULTI = CAN_PERSIST && rp_kblocks<TP>() > 1;
    static constexpr size_t SLDS = (size_t)4 * GW * GSplit<TP>::TOTAL * sizeof(T);   // LDS of a split workgroup (four wave pairs, sixteen walkers)
    bool split = false;              // dl_set_split: step launches use k_env_step_g16_split
    float* ctrl_dbg = nullptr;       // test hook (dl_debug_last_ctrl): sim.data.ctrl of the last single-step launch, float[N, nu]
    T *inj_q = nullptr, *inj_v = nullptr;
    int32_t* inj_flags = nullptr;
    bool inj_armed = false;
    double floor_friction_model = 0;
    float* scratch_obs = nullptr;

    ~EnvImpl() override { for (void* p : allocs) (void)hipFree(p); }
    template <typename U> int dalloc(U** p, size_t count) {
        HIPCHK(hipMalloc((void**)p, count * sizeof(U)));
        allocs.push_back(*p);
        HIPCHK(hipMemset(*p, 0, count * sizeof(U)));
        // hipMemset of device memory returns before the fill has run, and the fill sits on the NULL stream: a caller's non-blocking stream (torch's side streams) does not wait for
        // it, so the first kernel that WRITES a lazily allocated buffer could be overtaken by its zeroing (round 6: the packed policy weights of a HipEnvGroup handle's first
        // chunk -- wrong actions at step 0 in one of thirteen runs of test_env_group_handles_are_shards).  Allocation is rare: wait for the fill here.
        HIPCHK(hipStreamSynchronize(nullptr));
        return DL_OK;
    }
    int grid() const { return (n + BLOCK - 1) / BLOCK; }

    int init(const dl_model_desc& d, const dl_refs_desc& r, const dl_config& cfg, int n_, int device_) override {
        n = n_; device = device_; real_size = sizeof(T); obs_dim = TP::OBS; act_dim = TP::NU;
        HIPCHK(hipSetDevice(device));
        fill_dev_model<T, TP>(d, m);
        fill_dev_cfg<T>(cfg, r, c);
        floor_friction_model = d.floor_friction;
        {
            int rc0;
            if ((rc0 = dalloc(&md, 1))) return rc0;
            HIPCHK(hipMemcpy(md, &m, sizeof m, hipMemcpyHostToDevice));
        }
        if (r.n_rows != 2 * TP::NV) return fail(DL_E_INVAL, "refs.n_rows must be 2*nv");
        int rc;
        // reference table -> device, in the arithmetic type of the kernels
        const size_t tn = (size_t)r.n_rows * r.total_len;
        T* table; double* stage; T* svel; int32_t *soff, *sleft;
        if ((rc = dalloc(&table, tn))) return rc;
        if ((rc = dalloc(&stage, tn))) return rc;
        {
            std::vector<double> tr;
            transpose_refs(r, tr);              // sample-major: one walker's 2*nv reference values are contiguous
            HIPCHK(hipMemcpy(stage, tr.data(), tn * sizeof(double), hipMemcpyHostToDevice));
        }
        k_copy_cast<T><<<(unsigned)((tn + 255) / 256), 256>>>(table, stage, tn);
        HIPCHK(hipDeviceSynchronize());            // `stage` is reused below
        if ((rc = dalloc(&svel, r.n_steps))) return rc;
        HIPCHK(hipMemcpy(stage, r.step_vel, r.n_steps * sizeof(double), hipMemcpyHostToDevice));
        k_copy_cast<T><<<1, 256>>>(svel, stage, r.n_steps);
        HIPCHK(hipDeviceSynchronize());
        if ((rc = dalloc(&soff, r.n_steps + 1))) return rc;
        if ((rc = dalloc(&sleft, r.n_steps))) return rc;
        HIPCHK(hipMemcpy(soff, r.step_off, (r.n_steps + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sleft, r.step_is_left, r.n_steps * sizeof(int32_t), hipMemcpyHostToDevice));
        c.table = table; c.step_off = soff; c.step_is_left = sleft; c.step_vel = svel;
        if (TP::ENV_KIND == 1) {
            if (r.n_steps != 1) return fail(DL_E_INVAL, "loco3d reference tables are one continuous trajectory (n_steps must be 1)");
            std::vector<double> pref;
            loco3d_prefix_sums(r, TP::NV, pref);
            double* dpref;
            if ((rc = dalloc(&dpref, pref.size()))) return rc;
            HIPCHK(hipMemcpy(dpref, pref.data(), pref.size() * sizeof(double), hipMemcpyHostToDevice));
            c.pref = dpref;
        }
        // per-walker state
        st.n = n;
        if ((rc = fault_alloc())) return fail(rc, "dl_create: cannot allocate the fault word (pinned host memory)");
        st.fault = fault_dev; st.spin_dyn = GSplit<TP>::SPIN_LIMIT; st.spin_srv = GSplit<TP>::SPIN_LIMIT;
        if ((rc = dalloc(&st.qpos, (size_t)TP::NV * n))) return rc;
        if ((rc = dalloc(&st.qvel, (size_t)TP::NV * n))) return rc;
        if ((rc = dalloc(&st.warm, (size_t)TP::NV * n))) return rc;
        if ((rc = dalloc(&st.comz_off, n))) return rc;
        if ((rc = dalloc(&st.zacc, (size_t)r.n_steps * n))) return rc;          // quirk Q4: zero = the pristine data set
        st.strict_solver = cfg.strict_solver ? 1 : 0;
        if ((rc = dalloc(&st.cur, (size_t)DL_CUR_WORDS * n))) return rc;
        if ((rc = dalloc(&st.walked, n))) return rc;
        if ((rc = dalloc(&st.mon, (size_t)MON_WORDS * n))) return rc;
        if ((rc = dalloc(&st.need_reset, n))) return rc;
        if ((rc = dalloc(&st.inj_rsi, (size_t)2 * n))) return rc;
        if ((rc = dalloc(&st.work, (size_t)4 * TP::NV * n))) return rc;
        if ((rc = dalloc(&inj_q, (size_t)TP::NV * n))) return rc;
        if ((rc = dalloc(&inj_v, (size_t)TP::NV * n))) return rc;
        if ((rc = dalloc(&inj_flags, n))) return rc;
        if ((rc = dalloc(&scratch_obs, (size_t)TP::OBS * n))) return rc;
        if (cfg.lanes_per_walker != 0 && cfg.lanes_per_walker != 1 && cfg.lanes_per_walker != GL) return fail(DL_E_INVAL, "lanes_per_walker must be 0 (auto), 1 or 16");
        variant = cfg.lanes_per_walker == 1 ? 0 : 1;      // auto prefers the 16-lane kernels
        {
            GModel<T, TP> gmh;
            std::string why;
            if (!fill_group_model<T, TP>(d, gmh, why)) { if (cfg.lanes_per_walker == GL) return fail(DL_E_INVAL, why); variant = 0; }
            else {
                if ((rc = dalloc(&gmd, 1))) return rc;
                HIPCHK(hipMemcpy(gmd, &gmh, sizeof gmh, hipMemcpyHostToDevice));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward_g16<T, TP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLDS));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_env_step_g16<T, TP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLDS));
                if constexpr (CAN_SPLIT) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_env_step_g16_split<T, TP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SLDS));
            }
        }
        const unsigned g256 = (unsigned)((n + 255) / 256);
        for (int j = 0; j < TP::NV; j++) k_fill<T><<<g256, 256>>>(st.qpos + (size_t)j * n, (T)d.jnt_qpos0[j], (size_t)n);
        k_fill<int32_t><<<g256, 256>>>(st.cur + (size_t)DL_CUR_COUNT * n, 1, (size_t)n);   // count_steps_same_vel = 1
        k_fill<int32_t><<<(unsigned)((2 * n + 255) / 256), 256>>>(st.inj_rsi, -1, (size_t)2 * n);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_env_step<T, TP, BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_env_reset<T, TP, BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward<T, TP, BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipFree(stage));
        allocs.erase(std::find(allocs.begin(), allocs.end(), (void*)stage));
        return DL_OK;
    }
    int reset(const uint8_t* mask, const int32_t* is, const int32_t* ip, float* obs, hipStream_t s) override {
        if ((is == nullptr) != (ip == nullptr)) return fail(DL_E_INVAL, "init_step and init_pos must be given together");
        if (!mask) HIPCHK(hipMemsetAsync(st.mon + (size_t)MON_FIRST_LEN * n, 0, (size_t)6 * n * sizeof(double), s));          // a reset of ALL walkers opens a new "first episode" record (MON_FIRST_*, MON_WALKED_LAST, and the record's own step / reward counters)
        static_assert(MON_FIRST_MOVED == MON_FIRST_LEN + 1 && MON_FIRST_RET == MON_FIRST_LEN + 2 && MON_WALKED_LAST == MON_FIRST_LEN + 3 && MON_FIRST_CUR_LEN == MON_FIRST_LEN + 4 && MON_FIRST_CUR_RET == MON_FIRST_LEN + 5, "cleared as one run");
        hipLaunchKernelGGL((k_env_reset<T, TP, BLOCK>), dim3(grid()), dim3(BLOCK), LDS, s, (const DevModel<T, TP>*)md, c, st, 1, mask, is, ip, obs ? obs : scratch_obs, (float*)nullptr, eval_mode);
        HIPCHK(hipGetLastError());
        return DL_OK;
    }
    // dl_rollout_fixed: the 16-lane step kernel takes several control steps per launch (state in registers in between, and the
    // launch lasts as long as the wave with the largest SUM over the steps instead of paying every step's slowest wave)
    static constexpr int MULTI = 512;
    int steps_fixed(int nsteps, const float* act, float* obs, float* rew, uint8_t* done, hipStream_t s) override {
        if (!(variant == 1 && gmd) || inj_armed || nsteps <= 1) { const int rc = step(act, obs, rew, done, nullptr, nullptr, s); return rc == DL_OK ? 1 : rc; }
        const int k = nsteps < MULTI ? nsteps : MULTI;
        st.push_step0 = push_step; push_step += k;
        prof_begin(s);
        if (split) note_launch(((n + GW - 1) / GW + 3) / 4, 512, CAN_SPLIT ? SLDS : 0); else note_launch((n + GW - 1) / GW, 64, GLDS);
        if constexpr (CAN_SPLIT) {
            if (split) hipLaunchKernelGGL((k_env_step_g16_split<T, TP>), dim3(((n + GW - 1) / GW + 3) / 4), dim3(512), SLDS, s, (const GModel<T, TP>*)gmd, c, st, act, obs, rew, done, (float*)nullptr,
                                          (float*)nullptr, (const T*)inj_q, (const T*)inj_v, (const int32_t*)nullptr, (float*)nullptr, eval_mode, k);
        }
        if (!split)
        hipLaunchKernelGGL((k_env_step_g16<T, TP>), dim3((n + GW - 1) / GW), dim3(64), GLDS, s, (const GModel<T, TP>*)gmd, c, st, act, obs, rew, done, (float*)nullptr, (float*)nullptr,
                           (const T*)inj_q, (const T*)inj_v, (const int32_t*)nullptr, (float*)nullptr, eval_mode, k);
        if (prof_open) prof_steps += k;
        prof_end(s);
        HIPCHK(hipGetLastError());
        return k;
    }
    int step(const float* act, float* obs, float* rew, uint8_t* done, float* term, float* terms, hipStream_t s) override {
        if (!act || !obs || !rew || !done) return fail(DL_E_INVAL, "actions/obs/rew/done must not be NULL");
        if (variant == 1 && gmd) {
            st.push_step0 = push_step; push_step += 1;
            prof_begin(s);
            if (split) note_launch(((n + GW - 1) / GW + 3) / 4, 512, CAN_SPLIT ? SLDS : 0); else note_launch((n + GW - 1) / GW, 64, GLDS);
            if constexpr (CAN_SPLIT) {
                if (split) hipLaunchKernelGGL((k_env_step_g16_split<T, TP>), dim3(((n + GW - 1) / GW + 3) / 4), dim3(512), SLDS, s, (const GModel<T, TP>*)gmd, c, st, act, obs, rew, done, term, terms,
                                              (const T*)inj_q, (const T*)inj_v, (const int32_t*)(inj_armed ? inj_flags : nullptr), ctrl_dbg, eval_mode, 1);
            }
            if (!split)
            hipLaunchKernelGGL((k_env_step_g16<T, TP>), dim3((n + GW - 1) / GW), dim3(64), GLDS, s, (const GModel<T, TP>*)gmd, c, st, act, obs, rew, done, term, terms,
                               (const T*)inj_q, (const T*)inj_v, (const int32_t*)(inj_armed ? inj_flags : nullptr), ctrl_dbg, eval_mode, 1);
            if (prof_open) prof_steps += 1;
            prof_end(s);
            HIPCHK(hipGetLastError());
            if (inj_armed) { HIPCHK(hipMemsetAsync(inj_flags, 0, (size_t)n * sizeof(int32_t), s)); inj_armed = false; }
            return DL_OK;                 // finished walkers were re-initialised inside the launch
        } else {
        prof_begin(s);
        if (prof_open) prof_steps += 1;
        hipLaunchKernelGGL((k_env_step<T, TP, BLOCK>), dim3(grid()), dim3(BLOCK), LDS, s, (const DevModel<T, TP>*)md, c, st, act, obs, rew, done, term, terms,
                           (const T*)inj_q, (const T*)inj_v, (const int32_t*)(inj_armed ? inj_flags : nullptr));
        prof_end(s);
        }
        HIPCHK(hipGetLastError());
        if (inj_armed) { HIPCHK(hipMemsetAsync(inj_flags, 0, (size_t)n * sizeof(int32_t), s)); inj_armed = false; }
        hipLaunchKernelGGL((k_env_reset<T, TP, BLOCK>), dim3(grid()), dim3(BLOCK), LDS, s, (const DevModel<T, TP>*)md, c, st, 0, (const uint8_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, obs, term, eval_mode);
        HIPCHK(hipGetLastError());
        return DL_OK;
    }
    int get_state(void* q, void* v, void* w, int32_t* cur, double* walked, hipStream_t s) override {
        const size_t b = (size_t)TP::NV * n * sizeof(T);
        if (q) HIPCHK(hipMemcpyAsync(q, st.qpos, b, hipMemcpyDeviceToDevice, s));
        if (v) HIPCHK(hipMemcpyAsync(v, st.qvel, b, hipMemcpyDeviceToDevice, s));
        if (w) HIPCHK(hipMemcpyAsync(w, st.warm, b, hipMemcpyDeviceToDevice, s));
        if (cur) HIPCHK(hipMemcpyAsync(cur, st.cur, (size_t)DL_CUR_WORDS * n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (walked) HIPCHK(hipMemcpyAsync(walked, st.walked, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        return DL_OK;
    }
    int set_state(const void* q, const void* v, const void* w, const int32_t* cur, const double* walked, hipStream_t s) override {
        const size_t b = (size_t)TP::NV * n * sizeof(T);
        if (q) HIPCHK(hipMemcpyAsync(st.qpos, q, b, hipMemcpyDeviceToDevice, s));
        if (v) HIPCHK(hipMemcpyAsync(st.qvel, v, b, hipMemcpyDeviceToDevice, s));
        if (w) HIPCHK(hipMemcpyAsync(st.warm, w, b, hipMemcpyDeviceToDevice, s));
        if (cur) HIPCHK(hipMemcpyAsync(st.cur, cur, (size_t)DL_CUR_WORDS * n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (walked) HIPCHK(hipMemcpyAsync(st.walked, walked, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        return DL_OK;
    }
    int ref_offsets(void* get, const void* set, hipStream_t s) override {
        const size_t b = (size_t)c.n_steps * n * sizeof(T);
        if (get) HIPCHK(hipMemcpyAsync(get, st.zacc, b, hipMemcpyDeviceToDevice, s));
        if (set) HIPCHK(hipMemcpyAsync(st.zacc, set, b, hipMemcpyDeviceToDevice, s));
        return DL_OK;
    }
    int forward(const void* ctrl, void* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter, hipStream_t s) override {
        if (!qacc) return fail(DL_E_INVAL, "qacc must not be NULL");
        if (variant == 1 && gmd) {
            hipLaunchKernelGGL((k_forward_g16<T, TP>), dim3((n + GW - 1) / GW), dim3(64), GLDS, s, (const GModel<T, TP>*)gmd, st, (const T*)ctrl, (T*)qacc, ncon, nefc, niter);
            HIPCHK(hipGetLastError());
            return DL_OK;
        }
        hipLaunchKernelGGL((k_forward<T, TP, BLOCK>), dim3(grid()), dim3(BLOCK), LDS, s, (const DevModel<T, TP>*)md, st, (const T*)ctrl, (T*)qacc, ncon, nefc, niter);
        HIPCHK(hipGetLastError());
        return DL_OK;
    }
    int randomize(const float* mass_scale, const float* floor_friction, const float* push, bool set_push, hipStream_t s) override {
        if (!(variant == 1 && gmd)) return fail(DL_E_INVAL, "dynamics randomisation / pushes are implemented by the 16-lane kernels (lanes_per_walker = 16)");
        if (set_push) st.push_phase = nullptr;           // a plain dl_set_push ends a push schedule (push_schedule re-arms it afterwards)
        const unsigned g256 = (unsigned)((n + 255) / 256);
        if (!st.rnd) {
            int rc;
            if ((rc = dalloc(&st.rnd, (size_t)5 * n))) return rc;
            k_fill<T><<<g256, 256, 0, s>>>(st.rnd, T(1), (size_t)n);
            k_fill<T><<<g256, 256, 0, s>>>(st.rnd + n, (T)floor_friction_model, (size_t)n);
        }
        if (mass_scale) k_copy_strided<T><<<g256, 256, 0, s>>>(st.rnd, mass_scale, n, 1, 0);
        if (floor_friction) k_copy_strided<T><<<g256, 256, 0, s>>>(st.rnd + n, floor_friction, n, 1, 0);
        if (set_push) {
            for (int k = 0; k < 3; k++) {
                if (push) k_copy_strided<T><<<g256, 256, 0, s>>>(st.rnd + (size_t)(2 + k) * n, push, n, 3, k);
                else k_fill<T><<<g256, 256, 0, s>>>(st.rnd + (size_t)(2 + k) * n, T(0), (size_t)n);
            }
        }
        HIPCHK(hipGetLastError());
        return DL_OK;
    }
    // dl_set_push_schedule: force float[N, 3] (NULL switches the schedule off and clears the push), phase int32[N]
    int32_t* push_phase = nullptr;
    int push_step = 0;               // control steps taken since the schedule was set
    int push_schedule(const float* force, const int32_t* phase, int period, int duration, hipStream_t s) override {
        if (!force) { st.push_phase = nullptr; return randomize(nullptr, nullptr, nullptr, true, s); }
        if (!phase || period <= 0 || duration < 0 || duration > period) return fail(DL_E_INVAL, "dl_set_push_schedule: needs phase, period > 0 and 0 <= duration <= period");
        int rc = randomize(nullptr, nullptr, force, true, s);
        if (rc) return rc;
        if (!push_phase && (rc = dalloc(&push_phase, (size_t)n))) return rc;
        HIPCHK(hipMemcpyAsync(push_phase, phase, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        st.push_phase = push_phase; st.push_period = period; st.push_dur = duration;
        push_step = 0;
        return DL_OK;
    }
    int terminate_early(int32_t* flags, hipStream_t s) override {
        if (!flags) return fail(DL_E_INVAL, "dl_terminate_early: flags must not be NULL");
        if (TP::ENV_KIND != 0) return fail(DL_E_INVAL, "dl_terminate_early: defined for the straight walker only (mimic_env.py:654)");
        hipLaunchKernelGGL((k_terminate_early<T>), dim3((n + 255) / 256), dim3(256), 0, s, c, st, flags);
        HIPCHK(hipGetLastError());
        return DL_OK;
    }
    int snapshot(int word, double* out, hipStream_t s) override {
        hipLaunchKernelGGL(k_mon_snapshot, dim3((n + 255) / 256), dim3(256), 0, s, (const double*)st.mon, word, out, n);
        HIPCHK(hipGetLastError());
        return DL_OK;
    }
    int counters(int32_t* out, int clear, hipStream_t s) override {
        if (!st.dbg) {
            int rc; if ((rc = dalloc(&st.dbg, (size_t)4 * n))) return rc; if ((rc = dalloc(&st.dbgf, (size_t)(48 + DL_DBG_EVALS) * n))) return rc;
            const char* e = getenv("DL_DEBUG_CAP_ITERS");
            st.dbg_cap = e ? atoi(e) : (int)m.iterations;
        }
        if (out) HIPCHK(hipMemcpyAsync(out, st.dbg, (size_t)4 * n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (clear) HIPCHK(hipMemsetAsync(st.dbg, 0, (size_t)4 * n * sizeof(int32_t), s));
        return DL_OK;
    }
    int forward_timed(const void* ctrl, void* qacc, long long* tim, hipStream_t s) override {
        if (!gmd || !qacc || !tim) return fail(DL_E_INVAL, "dl_debug_forward_timed: needs the 16-lane kernels, qacc and tim");
        if constexpr (sizeof(T) == 4) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_forward_g16<T, TP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLDS));
            hipLaunchKernelGGL((k_forward_g16<T, TP, true>), dim3((n + GW - 1) / GW), dim3(64), GLDS, s, (const GModel<T, TP>*)gmd, st, (const T*)ctrl, (T*)qacc, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, tim);
            HIPCHK(hipGetLastError());
            return DL_OK;
        }
        return fail(DL_E_INVAL, "dl_debug_forward_timed: float32 only");
    }
    int step_timed(const float* act, float* obs, float* rew, uint8_t* done, long long* tim, hipStream_t s) override {
        if (!gmd || !act || !obs || !rew || !done || !tim) return fail(DL_E_INVAL, "dl_debug_step_timed: needs the 16-lane kernels and all arrays");
        if constexpr (sizeof(T) == 4) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_env_step_g16<T, TP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GLDS));
            hipLaunchKernelGGL((k_env_step_g16<T, TP, true>), dim3((n + GW - 1) / GW), dim3(64), GLDS, s, (const GModel<T, TP>*)gmd, c, st, act, obs, rew, done, (float*)nullptr, (float*)nullptr,
                               (const T*)inj_q, (const T*)inj_v, (const int32_t*)nullptr, (float*)nullptr, eval_mode, 1, tim);
            HIPCHK(hipGetLastError());
            return DL_OK;
        }
        return fail(DL_E_INVAL, "dl_debug_step_timed: float32 only");
    }
    void set_grid_spin(int polls) override { spin_grid = polls >= 0 ? polls : (1 << 22); }
    void set_spin_limits(int dyn, int srv) override { st.spin_dyn = dyn >= 0 ? dyn : GSplit<TP>::SPIN_LIMIT; st.spin_srv = srv >= 0 ? srv : GSplit<TP>::SPIN_LIMIT; }
    int set_split(int on) override {
        if (on && !(CAN_SPLIT && variant == 1 && gmd)) return fail(DL_E_INVAL, "dl_set_split: the split workgroup exists for the 16-lane float32 kernels");
        if (on && st.strict_solver) return fail(DL_E_INVAL, "dl_set_split: dl_config.strict_solver selects the one-wave form (the split workgroups run the product's solver path only)");
        split = on != 0;
        return DL_OK;
    }

    // ---- dl_collect_rollouts(DL_ROLLOUT_PERSISTENT): the whole rollout as one launch of k_rollout_persistent
    double* rp_partial = nullptr; double* rp_xpart = nullptr; unsigned* rp_sync = nullptr; long long* rp_prof = nullptr;
    float* pol_packed = nullptr;      // pol_packed_floats(512) (room for any supported hidden size): k-chunk-major copies of the policy's weights, refreshed by every rollout call
    int n_cus = 0;
    int spin_grid = 1 << 22;      // polls of the grid exchange before a workgroup gives up (~2 s)
    int pack_policy(const dl_policy_params& pol, PolPacked* out, hipStream_t s) override {
        *out = PolPacked{nullptr, nullptr, nullptr};
        const int H = pol.hidden;
        if (H != 512 && H != 256 && H != 128) return DL_OK;                  // the packed form is built for the eight-wave kernels
        int rc;
        if (!pol_packed && (rc = dalloc(&pol_packed, pol_packed_floats(512)))) return rc;
        const int chunks = H * (H / 4) + 12 * H + (H / 4) * 16;
        hipLaunchKernelGGL(k_pack_policy, dim3((chunks + 255) / 256), dim3(256), 0, s, pol, pol_packed);
        HIPCHK(hipGetLastError());
        out->w2p = pol_packed; out->w1p = pol_packed + (size_t)H * H; out->whp = out->w1p + (size_t)48 * H;
        return DL_OK;
    }
    int persistent_ok(int hidden, std::string* why) override {
        auto no = [&](const char* w) { if (why) *why = w; return 0; };
        if constexpr (!CAN_PERSIST) return no("the persistent rollout kernels exist for the 16-lane float32 kernels of walkers whose split workgroup + moments block fit the LDS of a CU (both walkers of the reference do)");
        else {
            if (!(variant == 1 && gmd)) return no("the persistent rollout kernel needs the 16-lane kernels (lanes_per_walker = 16)");
            if (hidden != 512 && hidden != 256 && hidden != 128) return no("the persistent rollout kernels are built for hidden = 512, 256 or 128 (eight waves per workgroup x 4 / 2 / 1 tiles)");
            if (inj_armed) return no("injected states are pending");
            if (st.strict_solver) return no("dl_config.strict_solver selects the one-wave step kernel (the persistent kernels run the split workgroups' solver path)");
            if (!n_cus) { if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) n_cus = 0; }
            if (n_cus <= 0 || (n + 15) / 16 > n_cus * rp_kblocks<TP>()) return no(rp_kblocks<TP>() > 1 ? "more than 128 walkers per CU: a workgroup of the persistent rollout kernel takes at most eight blocks of sixteen walkers" : "more than 16 walkers per CU: this walker's workgroups take one block of sixteen walkers");
            return 1;
        }
    }
    int collect_persistent(const dl_policy_params& pol, uint64_t seed, uint64_t counter0, int32_t index_base, const dl_vecnorm_state& vn, int32_t nT, float* observations,
                           float* actions, float* values, float* log_probs, float* rewards, uint8_t* episode_starts, float* next_obs, uint8_t* next_done, float* raw_obs,
                           float* raw_rew, int per_rollout, int deterministic, hipStream_t s) override {
        if constexpr (!CAN_PERSIST) return fail(DL_E_INVAL, "dl_collect_rollouts: no persistent form for this walker / precision");
        else {
            std::string why;
            if (!persistent_ok(pol.hidden, &why)) return fail(DL_E_INVAL, "dl_collect_rollouts: " + why);
            constexpr int W = TP::OBS + 1;
            const int nblk = (n + 15) / 16;
            int rc;
            if (!rp_partial) {
                if ((rc = dalloc(&rp_partial, (size_t)nblk * 4 * W * 2))) return rc;          // (per-rollout moments: a slot per wave pair)
                if ((rc = dalloc(&rp_xpart, (size_t)2 * 8 * W * 2))) return rc;
                if ((rc = dalloc(&rp_sync, (size_t)RP_SYNC_WORDS))) return rc;
#ifdef DL_EXP_ROLLOUT_PROF
                if ((rc = dalloc(&rp_prof, (size_t)nblk * 4 * 11 + (size_t)RP_PROF_STEPS * nblk * 4))) return rc;
#else
                if ((rc = dalloc(&rp_prof, (size_t)nblk * 4 * 11))) return rc;
#endif
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_persistent<TP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SLDS + rollout_lds_extra<TP>())));
                if constexpr (CAN_PERSIST_MULTI) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_persistent<TP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SLDS + rollout_lds_extra<TP>())));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_pairs<TP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SLDS + rollout_lds_extra<TP>())));
            }
            RolloutP a{};
            a.pol = pol; a.seed = seed; a.counter0 = counter0; a.index_base = index_base;
            a.obs_mean = vn.obs_mean; a.obs_var = vn.obs_var; a.obs_count = vn.obs_count; a.ret = vn.ret; a.ret_mean = vn.ret_mean; a.ret_var = vn.ret_var; a.ret_count = vn.ret_count;
            a.gamma = vn.gamma; a.eps = vn.eps; a.clip_obs = vn.clip_obs; a.clip_rew = vn.clip_rew; a.flags = vn.flags;
            a.observations = observations; a.actions = actions; a.values = values; a.log_probs = log_probs; a.rewards = rewards; a.next_obs = next_obs; a.raw_obs = raw_obs; a.raw_rew = raw_rew;
            a.episode_starts = episode_starts; a.next_done = next_done;
            a.partial = rp_partial; a.xpart = rp_xpart; a.sync = rp_sync; a.prof = rp_prof;
            if ((rc = pack_policy(pol, &a.pk, s))) return rc;
            a.T = nT; a.per_rollout = per_rollout ? 1 : 0; a.spin_grid = spin_grid; a.deterministic = deterministic ? 1 : 0;
            // at most one workgroup per CU (co-resident by construction): with more than sixteen walkers per CU a workgroup takes kblocks consecutive blocks
            a.kblocks = (nblk + n_cus - 1) / n_cus;
            const int nwg = (nblk + a.kblocks - 1) / a.kblocks;
            HIPCHK(hipMemsetAsync(rp_sync, 0, RP_SYNC_WORDS * sizeof(unsigned), s));
#ifdef DL_EXP_ROLLOUT_PROF
            if (nT > RP_PROF_STEPS) return fail(DL_E_INVAL, "DL_EXP_ROLLOUT_PROF build: at most 512 control steps per rollout");
            HIPCHK(hipMemsetAsync(rp_prof + (size_t)nblk * 4 * 11, 0, (size_t)RP_PROF_STEPS * nblk * 4 * sizeof(long long), s));
#endif
            if (per_rollout == 1) HIPCHK(hipMemsetAsync(rp_partial, 0, (size_t)nblk * 4 * W * 2 * sizeof(double), s));          // the pairs' sums start at zero (the workgroup form writes every slot it owns)
            st.push_step0 = push_step; push_step += nT;
            prof_begin(s);
            note_launch(nwg, 512, SLDS + rollout_lds_extra<TP>());
            RolloutArgs<TP> ra{};
            ra.gm = gmd; ra.c = c; ra.st = st; ra.a = a; ra.eval_mode = eval_mode;
            if (per_rollout == 1) hipLaunchKernelGGL((k_rollout_pairs<TP>), dim3(nwg), dim3(512), SLDS + rollout_lds_extra<TP>(), s, ra);
            else if (a.kblocks > 1) { if constexpr (CAN_PERSIST_MULTI) hipLaunchKernelGGL((k_rollout_persistent<TP, true>), dim3(nwg), dim3(512), SLDS + rollout_lds_extra<TP>(), s, ra); }
            else hipLaunchKernelGGL((k_rollout_persistent<TP, false>), dim3(nwg), dim3(512), SLDS + rollout_lds_extra<TP>(), s, ra);
            if (prof_open) prof_steps += nT;
            prof_end(s);
            HIPCHK(hipGetLastError());
            if (per_rollout && (vn.flags & 5))
                hipLaunchKernelGGL(k_vn_merge_rollout, dim3(1), dim3(64), 0, s, (const double*)rp_partial, vn.obs_mean, vn.obs_var, vn.obs_count, vn.ret_mean, vn.ret_var, vn.ret_count,
                                   per_rollout == 1 ? nblk * 4 : nblk, (int)TP::OBS, (long long)n * nT, vn.flags);          // a slot per wave pair (k_rollout_pairs) or per block of sixteen (k_rollout_persistent)
            HIPCHK(hipGetLastError());
            return DL_OK;
        }
    }
    int rollout_prof(long long* out, hipStream_t s) override {
        if (!rp_prof || !out) return fail(DL_E_INVAL, "dl_debug_rollout_prof: no persistent rollout has run on this handle");
#ifdef DL_EXP_ROLLOUT_PROF          // + int64[512][workgroups][4]: P, E, R, exchange wait per control step (workgroup-major inside a step; rows of workgroups that do not exist stay 0)
        HIPCHK(hipMemcpyAsync(out, rp_prof, ((size_t)((n + 15) / 16) * 4 * 11 + (size_t)RP_PROF_STEPS * ((n + 15) / 16) * 4) * sizeof(long long), hipMemcpyDeviceToDevice, s));
#else
        HIPCHK(hipMemcpyAsync(out, rp_prof, (size_t)((n + 15) / 16) * 4 * 11 * sizeof(long long), hipMemcpyDeviceToDevice, s));
#endif
        return DL_OK;
    }
    int last_ctrl(float* out, hipStream_t s) override {
        if (!(variant == 1 && gmd)) return fail(DL_E_INVAL, "dl_debug_last_ctrl: implemented by the 16-lane kernels");
        if (!ctrl_dbg) { const int rc = dalloc(&ctrl_dbg, (size_t)n * TP::NU); if (rc) return rc; }      // the first call enables the record
        if (out) HIPCHK(hipMemcpyAsync(out, ctrl_dbg, (size_t)n * TP::NU * sizeof(float), hipMemcpyDeviceToDevice, s));
        return DL_OK;
    }
    int capstate(float* out, hipStream_t s) override {
        if (!st.dbgf || !out) return fail(DL_E_INVAL, "dl_debug_capstate: enable the counters first");
        HIPCHK(hipMemcpyAsync(out, st.dbgf, (size_t)48 * n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return DL_OK;
    }
    int eval_iters(float* out, hipStream_t s) override {
        if (!st.dbgf || !out) return fail(DL_E_INVAL, "dl_debug_eval_iters: enable the counters first");
        HIPCHK(hipMemcpyAsync(out, st.dbgf + (size_t)48 * n, (size_t)DL_DBG_EVALS * n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return DL_OK;
    }
    int inject(const void* q, const void* v, const int32_t* flags, const int32_t* rsi, hipStream_t s) override {
        const size_t b = (size_t)TP::NV * n * sizeof(T);
        if (q) HIPCHK(hipMemcpyAsync(inj_q, q, b, hipMemcpyDeviceToDevice, s));
        if (v) HIPCHK(hipMemcpyAsync(inj_v, v, b, hipMemcpyDeviceToDevice, s));
        if (flags) { HIPCHK(hipMemcpyAsync(inj_flags, flags, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s)); inj_armed = true; }
        if (rsi) HIPCHK(hipMemcpyAsync(st.inj_rsi, rsi, (size_t)2 * n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        return DL_OK;
    }
};

// ------------------------------------------------------------------------------------------
extern "C" {

const char* dl_last_error(void) { return g_err.c_str(); }
int dl_abi_version(void) { return DL_ABI_VERSION; }
int dl_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(dl_model_desc);
        case 1: return (int)sizeof(dl_refs_desc);
        case 2: return (int)sizeof(dl_config);
        case 3: return (int)sizeof(dl_policy_params);
        case 4: return (int)sizeof(dl_vecnorm_state);
        default: return -1;
    }
}

int dl_dpp_wait_states(void) { return DL_DPP_WAIT; }
int dl_hw_probe(int32_t device, int32_t iters, uint64_t* out) {
    if (!out) return fail(DL_E_INVAL, "dl_hw_probe: out must not be NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DL_E_NODEVICE, "dl_hw_probe: no HIP device");
    if (device >= ndev) return fail(DL_E_INVAL, "dl_hw_probe: no such device");
    int prev = 0;
    HIPCHK(hipGetDevice(&prev));
    if (device >= 0) HIPCHK(hipSetDevice(device));
    unsigned long long o[8];
    const int e = hwprobe::probe(iters > 0 ? iters : 64, o);
    if (device >= 0) (void)hipSetDevice(prev);
    if (e != 0) return fail(DL_E_HIP, std::string("dl_hw_probe: ") + hipGetErrorString((hipError_t)e));
    for (int i = 0; i < 8; i++) out[i] = o[i];
    return DL_OK;
}
#if DL_DPP_WAIT < 2
// The one-wait-state build runs only on a device that has shown, in this process, that one state is enough (and that the test can tell): once per device.
static int dpp_one_state_proven(int device) {
    static std::mutex mu;
    static int verdict[64] = {0};          // 0 unknown, 1 proven, -1 refuted
    static unsigned long long seen[64][3];
    std::lock_guard<std::mutex> lock(mu);
    const int d = device < 0 || device >= 64 ? 0 : device;
    if (!verdict[d]) {
        uint64_t o[8];
        const int rc = dl_hw_probe(device, 32, o);
        if (rc != DL_OK) return rc;
        seen[d][0] = o[0]; seen[d][1] = o[1]; seen[d][2] = o[2];
        verdict[d] = (o[1] == 0 && o[2] == 0) ? 1 : -1;
    }
    if (verdict[d] < 0)
        return fail(DL_E_HIP, "dl_create: this library (libdrloco_hip_dpp1.so) pads its hand-written DPP reads with ONE wait state, and device " + std::to_string(device) + " shows stale DPP reads with one state (" +
                    std::to_string(seen[d][0]) + " / " + std::to_string(seen[d][1]) + " / " + std::to_string(seen[d][2]) + " with 0 / 1 / 2 states, dl_hw_probe): load libdrloco_hip.so, the build with the ISA manual's two states");
    return DL_OK;
}
#endif
int dl_create(const dl_model_desc* model, const dl_refs_desc* refs, const dl_config* cfg, int32_t n_envs, int32_t device, dl_handle* out) {
    if (!model || !refs || !cfg || !out || n_envs <= 0) return fail(DL_E_INVAL, "dl_create: bad arguments");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        // the reference's train.py hides the GPUs from its own process before it builds the environments (use_cpu(): CUDA_VISIBLE_DEVICES = "", drloco/train.py:35,80, whenever
        // drloco/config/config.py:10 USE_CPU = True, the default) and HIP honours that variable: say so instead of leaving the caller to guess
        std::string why = "dl_create: no HIP device (this library has no CPU path)";
        for (const char* var : {"CUDA_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"}) {
            const char* v = getenv(var);
            if (v && !*v) why += std::string("; ") + var + " is set to the empty string in this process, which hides every GPU -- the reference's train.py does that in use_cpu() "
                                 "(drloco/train.py:35) when drloco/config/config.py:10 USE_CPU is True: set USE_CPU = False (INTEGRATION.md 1)";
            else if (v) why += std::string("; ") + var + "=" + v;
        }
        return fail(DL_E_NODEVICE, why);
    }
    if (device < 0 || device >= ndev) return fail(DL_E_INVAL, "dl_create: device ordinal out of range");
#if DL_DPP_WAIT < 2
    { const int rc = dpp_one_state_proven(device); if (rc != DL_OK) return rc; }
#endif
    if (cfg->intended_semantics & ~(DL_INTENDED_COUNT_PER_EPISODE | DL_INTENDED_EVAL_OWN_STEP | DL_INTENDED_COMZ_PER_EPISODE)) return fail(DL_E_INVAL, "dl_create: unknown bits in intended_semantics");
    std::string why;
    if (cfg->precision != 32 && cfg->precision != 64 && cfg->precision != 0) return fail(DL_E_INVAL, "dl_create: precision must be 32 or 64");
    const bool f64 = cfg->precision == 64;
    dl_env_s* h = nullptr;
    if (cfg->env_kind == DL_ENV_STRAIGHT) {
        if (!check_topology<TopoStraight>(*model, why)) return fail(DL_E_INVAL, why);
        h = f64 ? static_cast<dl_env_s*>(new (std::nothrow) EnvImpl<double, TopoStraight>()) : static_cast<dl_env_s*>(new (std::nothrow) EnvImpl<float, TopoStraight>());
    } else if (cfg->env_kind == DL_ENV_LOCO3D) {
        if (!check_topology<TopoWalker165>(*model, why)) return fail(DL_E_INVAL, why);
        h = f64 ? static_cast<dl_env_s*>(new (std::nothrow) EnvImpl<double, TopoWalker165>()) : static_cast<dl_env_s*>(new (std::nothrow) EnvImpl<float, TopoWalker165>());
    } else return fail(DL_E_INVAL, "dl_create: unknown env_kind");
    if (!h) return fail(DL_E_NOMEM, "dl_create: out of host memory");
    const int rc = h->init(*model, *refs, *cfg, n_envs, device);
    if (rc != DL_OK) { delete h; return rc; }
    *out = h;
    return DL_OK;
}
int dl_destroy(dl_handle h) { delete h; return DL_OK; }
int32_t dl_num_envs(dl_handle h) { return h ? h->n : 0; }
int32_t dl_obs_dim(dl_handle h) { return h ? h->obs_dim : 0; }
int32_t dl_act_dim(dl_handle h) { return h ? h->act_dim : 0; }
int32_t dl_real_size(dl_handle h) { return h ? h->real_size : 0; }

#define NEED(h) do { if (!(h)) return fail(DL_E_INVAL, "null handle"); } while (0)
static int fault_error(dl_handle h) {
    const int code = h->fault_code();
    std::string why = "device fault " + std::to_string(code) + ":";
    if (code & DL_FAULT_DYN_TIMEOUT) why += " a dynamics wave of a split workgroup gave up waiting for its constraint wave;";
    if (code & DL_FAULT_SRV_TIMEOUT) why += " a constraint wave of a split workgroup gave up waiting for a request;";
    if (code & DL_FAULT_GRID_TIMEOUT) why += " a workgroup of the persistent rollout kernel gave up waiting for the grid-wide moment exchange (the rollout buffer is incomplete);";
    return fail(DL_E_FAULT, why + " the walkers of the affected waves took the exception path (reward 0, episode ended, reset) -- results since the fault are "
                "not a valid rollout; dl_fault_clear + dl_reset to continue");
}
// a fault raised by an EARLIER launch that has completed is reported by the next call (the word is host memory: no synchronisation needed)
#define NOFAULT(h) do { if ((h)->fault_code()) return fault_error(h); } while (0)
int dl_fault_check(dl_handle h, int32_t* code) {
    NEED(h);
    if (code) *code = h->fault_code();
    NOFAULT(h);
    return DL_OK;
}
int dl_fault_clear(dl_handle h) {
    NEED(h);
    if (h->fault_host) __atomic_store_n(h->fault_host, 0, __ATOMIC_RELAXED);
    return DL_OK;
}
/* test hook: poll budgets of the split workgroup's two wave roles (negative = default); 0 for the constraint waves makes them leave at once,
 * so that every dynamics wave's first request times out */
int dl_debug_set_spin_limit(dl_handle h, int32_t dyn, int32_t srv) {
    NEED(h);
    h->set_spin_limits(dyn, srv);
    return DL_OK;
}
/* test hook: poll budget of the persistent rollout kernel's grid exchange (negative = default); 0 makes every workgroup give up at its first exchange */
int dl_debug_set_grid_spin(dl_handle h, int32_t polls) {
    NEED(h);
    h->set_grid_spin(polls);
    return DL_OK;
}

int dl_reset(dl_handle h, const uint8_t* mask, const int32_t* init_step, const int32_t* init_pos, float* obs_out, void* stream) {
    NEED(h);
    return h->reset(mask, init_step, init_pos, obs_out, (hipStream_t)stream);
}
int dl_set_eval(dl_handle h, int32_t on) {
    NEED(h);
    h->eval_mode = on != 0;
    return DL_OK;
}
int dl_step(dl_handle h, const float* actions, float* obs, float* rew, uint8_t* done, float* term_obs, float* rew_terms, void* stream) {
    NEED(h);
    NOFAULT(h);
    return h->step(actions, obs, rew, done, term_obs, rew_terms, (hipStream_t)stream);
}
int dl_rollout_fixed(dl_handle h, int32_t T, const float* actions, float* obs, float* rew, uint8_t* done, void* stream) {
    NEED(h);
    NOFAULT(h);
    if (T <= 0 || !actions || !obs || !rew || !done) return fail(DL_E_INVAL, "dl_rollout_fixed: bad arguments");
    const size_t n = (size_t)h->n;
    for (int32_t t = 0; t < T;) {
        const int k = h->steps_fixed(T - t, actions + (size_t)t * n * h->act_dim, obs + (size_t)t * n * h->obs_dim, rew + (size_t)t * n, done + (size_t)t * n, (hipStream_t)stream);
        if (k <= 0) return k < 0 ? k : fail(DL_E_INVAL, "dl_rollout_fixed: no progress");
        t += k;
    }
    return DL_OK;
}
int dl_get_state(dl_handle h, void* qpos, void* qvel, void* qacc_warm, int32_t* cursor, double* walked, void* stream) {
    NEED(h);
    NOFAULT(h);
    return h->get_state(qpos, qvel, qacc_warm, cursor, walked, (hipStream_t)stream);
}
int dl_set_state(dl_handle h, const void* qpos, const void* qvel, const void* qacc_warm, const int32_t* cursor, const double* walked, void* stream) {
    NEED(h);
    return h->set_state(qpos, qvel, qacc_warm, cursor, walked, (hipStream_t)stream);
}
int dl_get_ref_offsets(dl_handle h, void* z_offsets, void* stream) {
    NEED(h);
    NOFAULT(h);
    if (!z_offsets) return fail(DL_E_INVAL, "dl_get_ref_offsets: z_offsets must not be NULL");
    return h->ref_offsets(z_offsets, nullptr, (hipStream_t)stream);
}
int dl_set_ref_offsets(dl_handle h, const void* z_offsets, void* stream) {
    NEED(h);
    if (!z_offsets) return fail(DL_E_INVAL, "dl_set_ref_offsets: z_offsets must not be NULL");
    return h->ref_offsets(nullptr, z_offsets, (hipStream_t)stream);
}
int dl_forward(dl_handle h, const void* ctrl, void* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter, void* stream) {
    NEED(h);
    return h->forward(ctrl, qacc, ncon, nefc, niter, (hipStream_t)stream);
}
/* test hooks (not part of the reference surface): inject end states / exceptions for the next
 * step (flags int32[N]: 1 = use inj state, 2 = diverge) and a fixed RSI draw (int32[2,N], -1 = off) */
int dl_debug_inject(dl_handle h, const void* qpos, const void* qvel, const int32_t* flags, const int32_t* rsi, void* stream) {
    NEED(h);
    return h->inject(qpos, qvel, flags, rsi, (hipStream_t)stream);
}
/* diagnostics of the 16-lane step kernel (enabled by the first call): out int32[4, N] device or NULL =
 * per walker {sum of Newton iterations, max iterations of the last step, sum of constraint rows, diverged steps} */
int dl_debug_counters(dl_handle h, int32_t* out, int32_t clear, void* stream) {
    NEED(h);
    return h->counters(out, clear, (hipStream_t)stream);
}
/* one forward evaluation (16-lane f32 kernels) with per-section cycle counts: tim int64[8, ceil(N/4)] device */
int dl_debug_forward_timed(dl_handle h, const void* ctrl, void* qacc, long long* tim, void* stream) {
    NEED(h);
    return h->forward_timed(ctrl, qacc, tim, (hipStream_t)stream);
}
/* one control step (16-lane f32 kernels) with per-section cycle counts: tim int64[10, ceil(N/4)] device; [7] = whole kernel, [8] before / [9] after the physics */
int dl_debug_step_timed(dl_handle h, const float* actions, float* obs, float* rew, uint8_t* done, long long* tim, void* stream) {
    NEED(h);
    return h->step_timed(actions, obs, rew, done, tim, (hipStream_t)stream);
}
/* in: float[128] device, out: float[256] device (see k_selftest) */
int dl_debug_selftest(const float* in, float* out, void* stream) {
    if (!in || !out) return fail(DL_E_INVAL, "dl_debug_selftest: bad arguments");
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
/* float[48, N] device: (q, v, warmstart)[16] of the last evaluation per walker that hit the iteration cap */
int dl_debug_capstate(dl_handle h, float* out, void* stream) {
    NEED(h);
    return h->capstate(out, (hipStream_t)stream);
}
/* float[DL_DBG_EVALS = 40, N] device: Newton iterations + 128 x constraint rows of every walker in the forward evaluations (4 per mj_step) of the
 * last control step (16-lane step kernels, after dl_debug_counters has enabled the diagnostics) */
int dl_debug_eval_iters(dl_handle h, float* out, void* stream) {
    NEED(h);
    return h->eval_iters(out, (hipStream_t)stream);
}
/* sim.data.ctrl as the last dl_step set it (after _rescale_actions and mirror_action): float[N, nu] device; the first call
 * (out may be NULL) enables the record */
int dl_set_split(dl_handle h, int32_t on) {
    NEED(h);
    return h->set_split(on);
}
/* diagnostics of the persistent rollout kernel (builds with -DDL_EXP_ROLLOUT_PROF; zeros otherwise): int64[ceil(N/16), 4] device = per workgroup
 * the shader-clock cycles spent in the policy phase, the env phase, the moment sums + exchange, and waiting inside the exchange; followed by
 * int64[10, 4 ceil(N/16)]: the sections of the last control step's env phase per dynamics wave (g_wave_env_step<TIMED>) */
int dl_debug_rollout_prof(dl_handle h, long long* out, void* stream) {
    NEED(h);
    return h->rollout_prof(out, (hipStream_t)stream);
}
int dl_debug_last_ctrl(dl_handle h, float* out, void* stream) {
    NEED(h);
    return h->last_ctrl(out, (hipStream_t)stream);
}
int dl_set_randomization(dl_handle h, const float* mass_scale, const float* floor_friction, void* stream) {
    NEED(h);
    return h->randomize(mass_scale, floor_friction, nullptr, false, (hipStream_t)stream);
}
int dl_set_push(dl_handle h, const float* force, void* stream) {
    NEED(h);
    return h->randomize(nullptr, nullptr, force, true, (hipStream_t)stream);
}
int dl_set_push_schedule(dl_handle h, const float* force, const int32_t* phase, int32_t period, int32_t duration, void* stream) {
    NEED(h);
    return h->push_schedule(force, phase, period, duration, (hipStream_t)stream);
}
int dl_terminate_early(dl_handle h, int32_t* flags, void* stream) {
    NEED(h);
    return h->terminate_early(flags, (hipStream_t)stream);
}
int dl_stats_snapshot(dl_handle h, const char* name, double* out, void* stream) {
    NEED(h);
    static const struct { const char* name; int word; } tab[] = {
        {"ep_len_smoothed", MON_S_EP_LEN}, {"ep_ret_smoothed", MON_S_EP_RET}, {"mean_reward_smoothed", MON_S_MEAN_REW},
        {"moved_distance", MON_MOVED}, {"mean_ep_pos_rew_smoothed", MON_S_POS}, {"mean_ep_vel_rew_smoothed", MON_S_VEL},
        {"mean_ep_com_rew_smoothed", MON_S_COM}, {"mean_abs_ep_torque_smoothed", MON_S_TOR}, {"ep_len", MON_EP_LEN},
        {"diverged_steps", MON_DIVERGED}, {"first_ep_len", MON_FIRST_LEN}, {"first_ep_moved", MON_FIRST_MOVED}, {"first_ep_ret", MON_FIRST_RET}, {"init_pos", MON_INIT_POS}, {"et_pos", MON_ET_POS}, {"last_abs_torque", MON_TOR_LAST}, {"difficult", MON_DIFFICULT}};
    for (const auto& t : tab)
        if (!strcmp(name, t.name)) return h->snapshot(t.word, out, (hipStream_t)stream);
    return fail(DL_E_INVAL, std::string("dl_stats_snapshot: unknown attribute ") + name);
}

int dl_profile(dl_handle h, int32_t enable) {
    NEED(h);
    h->prof = enable > 0 ? enable : 0;
    h->prof_tick = 0;
    h->prof_steps = 0;
    h->ev_used = 0;
    return DL_OK;
}
int dl_profile_read(dl_handle h, double* total_ms, int32_t* launches) {
    NEED(h);
    double tot = 0;
    int cnt = 0;
    for (size_t k = 0; k + 1 < h->ev_used; k += 2) {
        HIPCHK(hipEventSynchronize(h->ev[k + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h->ev[k], h->ev[k + 1]));
        tot += ms;
        cnt++;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    h->prof_last_steps = h->prof_steps;
    h->prof_steps = 0;
    h->ev_used = 0;
    NOFAULT(h);                 // a host sync point: the bracketed launches have completed
    return DL_OK;
}

int dl_profile_steps(dl_handle h) {
    NEED(h);
    return h->prof_last_steps;
}
int dl_profile_launch_config(dl_handle h, int32_t* out3) {
    NEED(h);
    if (!out3) return fail(DL_E_INVAL, "dl_profile_launch_config: out must not be NULL");
    for (int k = 0; k < 3; k++) out3[k] = h->launch_cfg[k];
    return DL_OK;
}

int dl_moments_update(double* mean, double* var, double* count, const float* x, int32_t B, int32_t D, void* stream) {
    if (!mean || !var || !count || !x || B <= 0 || D <= 0) return fail(DL_E_INVAL, "dl_moments_update: bad arguments");
    hipLaunchKernelGGL((k_moments<float>), dim3(D), dim3(256), 0, (hipStream_t)stream, mean, var, (const double*)count, x, B, D);
    hipLaunchKernelGGL(k_count_add, dim3(1), dim3(1), 0, (hipStream_t)stream, count, (double)B);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_normalize_obs(float* x, const double* mean, const double* var, int32_t B, int32_t D, double eps, double clip, void* stream) {
    if (!x || !mean || !var || B <= 0 || D <= 0) return fail(DL_E_INVAL, "dl_normalize_obs: bad arguments");
    const size_t n = (size_t)B * D;
    hipLaunchKernelGGL(k_normalize_obs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mean, var, B, D, eps, clip);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_normalize_reward(float* rew, double* ret, const uint8_t* done, double* ret_mean, double* ret_var, double* ret_count, int32_t B, double gamma, double eps, double clip, void* stream) {
    if (!rew || !ret || !done || !ret_mean || !ret_var || !ret_count || B <= 0) return fail(DL_E_INVAL, "dl_normalize_reward: bad arguments");
    const unsigned g = (unsigned)((B + 255) / 256);
    hipLaunchKernelGGL(k_ret_accumulate, dim3(g), dim3(256), 0, (hipStream_t)stream, ret, (const float*)rew, B, gamma);
    hipLaunchKernelGGL((k_moments<double>), dim3(1), dim3(256), 0, (hipStream_t)stream, ret_mean, ret_var, (const double*)ret_count, (const double*)ret, B, 1);
    hipLaunchKernelGGL(k_count_add, dim3(1), dim3(1), 0, (hipStream_t)stream, ret_count, (double)B);
    hipLaunchKernelGGL(k_reward_finish, dim3(g), dim3(256), 0, (hipStream_t)stream, rew, ret, done, (const double*)ret_var, B, eps, clip);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
// ---- K consecutive VecNormalize steps in five launches (dl_vecnormalize_steps): the per-step batch sums do not depend on each other once the
// shift of the sums is fixed (the moments at the start of the run), only the Chan merge is sequential -- and that is K x (D + 1) scalars.
//   k_vns_returns: ret_t = ret_{t-1} gamma + rew_t per walker (and the reset at episode ends), all K values kept for the sums
//   k_vns_partial: per step and block the shifted sums of the observation columns and of ret_t (the layout of k_vn_reduce_mb)
//   k_vns_merge:   per column the K merges in step order; leaves the running (mean, var) AFTER every step for k_vns_apply
//   k_vns_apply:   the K normalisations
__global__ __launch_bounds__(256) void k_vns_returns(const float* __restrict__ rew, const uint8_t* __restrict__ done, double* ret, double* rets, int K, int B, double gamma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double r = ret[i];
    // sixteen steps per trip: their loads are independent of the recurrence and go out together (one memory round trip per sixteen steps
    // instead of one per step -- the kernel is a chain of K dependent steps per walker and nothing else hides the latency)
    int t = 0;
    for (; t + 16 <= K; t += 16) {
        float w[16]; uint8_t d[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { w[u] = rew[(size_t)(t + u) * B + i]; d[u] = done[(size_t)(t + u) * B + i]; }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            r = r * gamma + (double)w[u];
            rets[(size_t)(t + u) * B + i] = r;
            if (d[u]) r = 0;
        }
    }
    for (; t < K; t++) {
        r = r * gamma + (double)rew[(size_t)t * B + i];
        rets[(size_t)t * B + i] = r;
        if (done[(size_t)t * B + i]) r = 0;
    }
    ret[i] = r;
}
__global__ __launch_bounds__(256) void k_vns_partial(const float* __restrict__ x, const double* __restrict__ rets, const double* __restrict__ mean, const double* __restrict__ ret_mean,
                                                  int B, int D, int flags, double* work) {
    __shared__ double sh[2][256];
    __shared__ double sh2[4];
    const int t = threadIdx.x, W = D + 1, step = blockIdx.y;
    double* wk = work + ((size_t)step * VN_BLOCKS + blockIdx.x) * W * 2;
    if (flags & 1) {
        const float* xs = x + (size_t)step * B * D;
        const int rpb = blockDim.x / D, nthr = rpb * D;
        double s = 0, ss = 0;
        const int col = t % D, rsub = t / D;
        if (t < nthr) {
            const double K0 = mean[col];
            const int stride = VN_BLOCKS * rpb;
            int row = blockIdx.x * rpb + rsub;
            for (; row + 3 * stride < B; row += 4 * stride) {
                const float a0 = xs[(size_t)row * D + col], a1 = xs[(size_t)(row + stride) * D + col], a2 = xs[(size_t)(row + 2 * stride) * D + col], a3 = xs[(size_t)(row + 3 * stride) * D + col];
                const double d0 = (double)a0 - K0, d1 = (double)a1 - K0, d2 = (double)a2 - K0, d3 = (double)a3 - K0;
                s += d0; ss += d0 * d0; s += d1; ss += d1 * d1; s += d2; ss += d2 * d2; s += d3; ss += d3 * d3;
            }
            for (; row < B; row += stride) { const double d = (double)xs[(size_t)row * D + col] - K0; s += d; ss += d * d; }
        }
        sh[0][t] = s; sh[1][t] = ss;
        __syncthreads();
        if (t < D) {
            for (int r = 1; r < rpb; r++) { s += sh[0][r * D + t]; ss += sh[1][r * D + t]; }
            wk[t * 2] = s; wk[t * 2 + 1] = ss;
        }
    }
    if (flags & 4) {
        const int chunk = (B + VN_BLOCKS - 1) / VN_BLOCKS, lo = blockIdx.x * chunk, hi = lo + chunk < B ? lo + chunk : B;
        const double K0 = *ret_mean;
        const double* rs = rets + (size_t)step * B;
        double s = 0, ss = 0;
        for (int i = lo + t; i < hi; i += blockDim.x) { const double d = rs[i] - K0; s += d; ss += d * d; }
        s = block_sum(s, sh2); ss = block_sum(ss, sh2);
        if (t == 0) { wk[D * 2] = s; wk[D * 2 + 1] = ss; }
    }
}
// Three kernels: k_vns_sum adds every step's 32 block partials up, in block order (K x (D + 1) independent sums, spread over the chip, left in
// `stats`); k_vns_merge -- one small workgroup -- runs the K Chan merges of the D + 1 columns in step order, the only sequential part (~10 float64
// operations per step, the sums of eight steps requested ahead); k_vns_rstd turns every step's variances into 1 / sqrt(var + eps) for
// k_vns_apply: the float64 square root and division happen once per step and column, not once per element.
__global__ __launch_bounds__(256) void k_vns_sum(const double* __restrict__ work, int K, int W, double* stats) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= K * W) return;
    const int t = e / W, d = e % W;
    const double* wk = work + (size_t)t * VN_BLOCKS * W * 2;
    double S = 0, SS = 0;
    for (int b = 0; b < VN_BLOCKS; b++) { S += wk[((size_t)b * W + d) * 2]; SS += wk[((size_t)b * W + d) * 2 + 1]; }
    stats[(size_t)e * 2] = S; stats[(size_t)e * 2 + 1] = SS;
}
// var -> 1 / sqrt(var + eps) for every step and column (after the merges)
__global__ __launch_bounds__(256) void k_vns_rstd(double* stats, int KW, double eps) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < KW) stats[(size_t)e * 2 + 1] = 1.0 / sqrt(stats[(size_t)e * 2 + 1] + eps);
}
__global__ __launch_bounds__(128) void k_vns_merge(double* mean, double* var, double* count, double* ret_mean, double* ret_var, double* ret_count,
                                                int K, int B, int D, int flags, double* stats) {
    const int W = D + 1;
    const int d = threadIdx.x;
    const bool active = d < W, is_obs = d < D, upd = is_obs ? (flags & 1) != 0 : (flags & 4) != 0;
    double* mp = is_obs ? mean + d : ret_mean;
    double* vp = is_obs ? var + d : ret_var;
    double m = 0, v = 0, cnt = 0;
    if (active) { m = *mp; v = *vp; cnt = is_obs ? *count : *ret_count; }
    __syncthreads();                           // every column has read the counts before one of them writes them back
    if (!active) return;
    const double K0 = m;                       // the shift of every step's sums
    for (int t0 = 0; t0 < K; t0 += 8) {
        double Sb[8], SSb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int t = t0 + u < K ? t0 + u : K - 1; Sb[u] = stats[((size_t)t * W + d) * 2]; SSb[u] = stats[((size_t)t * W + d) * 2 + 1]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = t0 + u;
            if (t < K) {
                if (upd) {
                    const double S = Sb[u], SS = SSb[u];
                    const double bm = K0 + S / B, bv = SS / B - (S / B) * (S / B);
                    const double tot = cnt + B, delta = bm - m;
                    const double M2 = v * cnt + bv * B + delta * delta * cnt * B / tot;
                    m = m + delta * B / tot; v = M2 / tot; cnt = tot;
                }
                stats[((size_t)t * W + d) * 2] = m; stats[((size_t)t * W + d) * 2 + 1] = v;
            }
        }
    }
    *mp = m; *vp = v;
    if (upd && (d == 0 || d == D)) { if (is_obs) *count = cnt; else *ret_count = cnt; }
}
__global__ __launch_bounds__(256) void k_vns_apply(const float* __restrict__ x, const float* __restrict__ rew, const double* __restrict__ stats, int K, int B, int D,
                                                double clip_obs, double clip_rew, int flags, float* const* obs_out, float* const* rew_out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, per = (size_t)B * D;
    const int W = D + 1;
    if (idx < (size_t)K * per) {
        const int t = (int)(idx / per);
        const size_t e = idx - (size_t)t * per;
        const int k = (int)(e % D);
        const double* st = stats + ((size_t)t * W + k) * 2;          // (mean, 1 / sqrt(var + eps)) after step t
        float o = x[idx];
        if (flags & 2) { double y = ((double)o - st[0]) * st[1]; y = y < -clip_obs ? -clip_obs : (y > clip_obs ? clip_obs : y); o = (float)y; }
        obs_out[t][e] = o;
    }
    if (idx < (size_t)K * B) {
        const int t = (int)(idx / B);
        const size_t e = idx - (size_t)t * B;
        float o = rew[idx];
        if (flags & 8) { double y = (double)o * stats[((size_t)t * W + D) * 2 + 1]; y = y < -clip_rew ? -clip_rew : (y > clip_rew ? clip_rew : y); o = (float)y; }
        rew_out[t][e] = o;
    }
}
static int vn_reduce_launch(const float* obs, const float* rew, double* obs_mean, double* obs_var, double* obs_count, double* ret, double* ret_mean, double* ret_var,
                            double* ret_count, int32_t B, int32_t D, double gamma, int32_t flags, void* workspace, void* stream) {
    if ((flags & 5) && (flags & 32)) {
        if (D > 63) return fail(DL_E_INVAL, "dl_vecnormalize_step: the blocked reduction order (flag 32) takes at most 63 observation columns");
        hipLaunchKernelGGL(k_vn_reduce_blk, dim3(1), dim3(1024), 0, (hipStream_t)stream, obs, rew, obs_mean, obs_var, (const double*)obs_count, ret, ret_mean, ret_var,
                           (const double*)ret_count, B, D, gamma, flags);
    } else if ((flags & 5) && (flags & 16) && workspace) {
        double* work = (double*)workspace;
        unsigned* arrive = (unsigned*)(work + (size_t)2 * VN_BLOCKS * (D + 1));
        hipLaunchKernelGGL(k_vn_reduce_mb, dim3(VN_BLOCKS), dim3(256), 0, (hipStream_t)stream, obs, rew, obs_mean, obs_var, (const double*)obs_count, ret, ret_mean, ret_var,
                           (const double*)ret_count, B, D, gamma, flags, work, arrive);
    } else if (flags & 5)
        hipLaunchKernelGGL(k_vn_reduce, dim3(1), dim3(VN_THREADS), 0, (hipStream_t)stream, obs, rew, obs_mean, obs_var, (const double*)obs_count, ret, ret_mean, ret_var,
                           (const double*)ret_count, B, D, gamma, flags);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_vecnormalize_step(const float* obs, const float* rew, const uint8_t* done, double* obs_mean, double* obs_var, double* obs_count,
                         double* ret, double* ret_mean, double* ret_var, double* ret_count, int32_t B, int32_t D, double gamma, double eps,
                         double clip_obs, double clip_rew, int32_t flags, float* obs_out, float* rew_out, void* workspace, void* stream) {
    if (!obs || !rew || !done || !obs_mean || !obs_var || !obs_count || !ret || !ret_mean || !ret_var || !ret_count || !obs_out || !rew_out || B <= 0 || D <= 0 || D > 128)
        return fail(DL_E_INVAL, "dl_vecnormalize_step: bad arguments");
    if (!(flags & 64)) {         // 64: the moments were advanced by dl_vn_local_sums + the caller's all-reduce + dl_vn_merge_sums
        const int rc = vn_reduce_launch(obs, rew, obs_mean, obs_var, obs_count, ret, ret_mean, ret_var, ret_count, B, D, gamma, flags, workspace, stream);
        if (rc) return rc;
    }
    const size_t ne = (size_t)B * D;
    hipLaunchKernelGGL(k_vn_apply, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, rew, done, (const double*)obs_mean, (const double*)obs_var, obs_count,
                       ret, (const double*)ret_var, ret_count, B, D, eps, clip_obs, clip_rew, flags, obs_out, rew_out);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_vn_local_sums(const float* obs, const float* rew, const double* obs_mean, double* ret, const double* ret_mean, int32_t B, int32_t D, double gamma, int32_t flags,
                     double* sums, void* stream) {
    if (!obs || !rew || !obs_mean || !ret || !ret_mean || !sums || B <= 0 || D <= 0 || D > 63) return fail(DL_E_INVAL, "dl_vn_local_sums: bad arguments");
    hipLaunchKernelGGL(k_vn_sums_blk, dim3(1), dim3(1024), 0, (hipStream_t)stream, obs, rew, obs_mean, ret, ret_mean, B, D, gamma, flags, sums);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_vn_merge_sums(const double* sums, int64_t B_global, double* obs_mean, double* obs_var, double* obs_count, double* ret_mean, double* ret_var, double* ret_count,
                     int32_t D, int32_t flags, void* stream) {
    if (!sums || !obs_mean || !obs_var || !obs_count || !ret_mean || !ret_var || !ret_count || B_global <= 0 || D <= 0 || D > 127) return fail(DL_E_INVAL, "dl_vn_merge_sums: bad arguments");
    hipLaunchKernelGGL(k_vn_merge_sums, dim3(1), dim3(128), 0, (hipStream_t)stream, sums, obs_mean, obs_var, obs_count, ret_mean, ret_var, ret_count, (long long)B_global, D, flags);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_vecnormalize_steps(const dl_vecnorm_state* vn, int32_t K, const float* obs, const float* rew, const uint8_t* done, int32_t B, int32_t D,
                          float* const* obs_out, float* const* rew_out, void* workspace, void* stream) {
    if (!vn || !obs || !rew || !done || !obs_out || !rew_out || !workspace || K <= 0 || B <= 0 || D <= 0 || D > 127)
        return fail(DL_E_INVAL, "dl_vecnormalize_steps: bad arguments");
    if (!vn->obs_mean || !vn->obs_var || !vn->obs_count || !vn->ret || !vn->ret_mean || !vn->ret_var || !vn->ret_count) return fail(DL_E_INVAL, "dl_vecnormalize_steps: NULL state array");
    const int W = D + 1, flags = vn->flags;
    double* rets = (double*)workspace;                               // [K, B]
    double* work = rets + (size_t)K * B;                             // [K, VN_BLOCKS, W, 2]
    double* stats = work + (size_t)K * VN_BLOCKS * W * 2;            // [K, W, 2]
    hipStream_t s = (hipStream_t)stream;
    if (flags & 4) hipLaunchKernelGGL(k_vns_returns, dim3((B + 255) / 256), dim3(256), 0, s, rew, done, vn->ret, rets, K, B, vn->gamma);
    if (flags & 5) hipLaunchKernelGGL(k_vns_partial, dim3(VN_BLOCKS, K), dim3(256), 0, s, obs, (const double*)rets, (const double*)vn->obs_mean, (const double*)vn->ret_mean, B, D, flags, work);
    hipLaunchKernelGGL(k_vns_sum, dim3((K * W + 255) / 256), dim3(256), 0, s, (const double*)work, K, W, stats);
    hipLaunchKernelGGL(k_vns_merge, dim3(1), dim3(128), 0, s, vn->obs_mean, vn->obs_var, vn->obs_count, vn->ret_mean, vn->ret_var, vn->ret_count, K, B, D, flags, stats);
    hipLaunchKernelGGL(k_vns_rstd, dim3((K * W + 255) / 256), dim3(256), 0, s, stats, K * W, vn->eps);
    const size_t ne = (size_t)K * B * D;
    hipLaunchKernelGGL(k_vns_apply, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, obs, rew, (const double*)stats, K, B, D, vn->clip_obs, vn->clip_rew, flags, obs_out, rew_out);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
static int policy_launch(const dl_policy_params* p, const float* obs, int32_t n, const float* eps, uint64_t seed, uint64_t counter, int32_t index_base,
                         int32_t deterministic, float* actions, float* values, float* log_probs, const PolVnFuse& vf, void* stream, const PolPacked pk = PolPacked{nullptr, nullptr, nullptr}) {
    if (!p || !(obs || vf.raw_obs) || !actions || !values || !log_probs || n <= 0) return fail(DL_E_INVAL, "dl_policy_forward: bad arguments");
    if (!p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->wa || !p->ba || !p->wv || !p->bv || !p->log_std) return fail(DL_E_INVAL, "dl_policy_forward: NULL parameter array");
    if (p->hidden <= 0 || p->hidden % 64 || p->hidden > 64 * POL_MAXT || p->obs_dim <= 0 || p->obs_dim > 48 || p->act_dim <= 0 || p->act_dim > 15)
        return fail(DL_E_INVAL, "dl_policy_forward: hidden must be a multiple of 64 and <= 512, obs_dim <= 48, act_dim <= 15");
    // hidden = 512 / 256 / 128 run as EIGHT waves x 4 / 2 / 1 tiles (the form the persistent rollout kernels call: the heads' eight partial sums, hence the bits, are the same
    // in every form); hidden = 64 as four waves x 1 tile
    const int nw = p->hidden >= 128 ? 8 : 4;                    // waves per workgroup
    const int H = p->hidden;
    const size_t lds = pol_lds_bytes(nw);                       // 23 KB (8 waves): below the default limit, no attribute needed on any device
    const dim3 grid((n + POL_ROWS - 1) / POL_ROWS), block(64 * nw);
#define DL_POL_LAUNCH(NTW, NW) \
    hipLaunchKernelGGL((k_policy_forward<NTW, NW>), grid, block, lds, (hipStream_t)stream, *p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, log_probs, vf, PolPacked{nullptr, nullptr, nullptr});
#define DL_POL_LAUNCH8(NTW) \
            if (n <= POL_ROWS * 256 && pk.w2p) {   /* ... and with the hidden layer's weights packed by the caller of a whole rollout (same arithmetic: bit-identical) */ \
                hipLaunchKernelGGL((k_policy_forward<NTW, 8, true, true>), grid, block, pol_lds_bytes_whole(8, H), (hipStream_t)stream, *p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, log_probs, vf, pk); \
            } else if (n <= POL_ROWS * 256) {   /* at most one workgroup per CU on an MI355X: the barrier-free form with the whole h1 block in LDS (51 KB for 512) */ \
                hipLaunchKernelGGL((k_policy_forward<NTW, 8, true>), grid, block, pol_lds_bytes_whole(8, H), (hipStream_t)stream, *p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, log_probs, vf, PolPacked{nullptr, nullptr, nullptr}); \
            } else if (pk.w2p) {                /* more rows than one workgroup per CU: the lean (23 KB) form, packed weights */ \
                hipLaunchKernelGGL((k_policy_forward<NTW, 8, false, true>), grid, block, lds, (hipStream_t)stream, *p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, log_probs, vf, pk); \
            } else DL_POL_LAUNCH(NTW, 8)
    switch (p->hidden / 64) {
        case 1: DL_POL_LAUNCH(1, 4) break;
        case 2: DL_POL_LAUNCH8(1) break;
        case 4: DL_POL_LAUNCH8(2) break;
        case 8: DL_POL_LAUNCH8(4) break;
        default: return fail(DL_E_INVAL, "dl_policy_forward: hidden must be 64, 128, 256 or 512");
    }
#undef DL_POL_LAUNCH8
#undef DL_POL_LAUNCH
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_policy_forward(const dl_policy_params* p, const float* obs, int32_t n, const float* eps, uint64_t seed, uint64_t counter, int32_t index_base,
                      int32_t deterministic, float* actions, float* values, float* log_probs, void* stream) {
    if (!obs) return fail(DL_E_INVAL, "dl_policy_forward: bad arguments");
    PolVnFuse none{};
    return policy_launch(p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, log_probs, none, stream);
}
int dl_policy_pack(const dl_policy_params* p, float* packed, void* stream) {
    if (!p || !packed || !p->w1 || !p->w2 || !p->wa || !p->wv) return fail(DL_E_INVAL, "dl_policy_pack: bad arguments");
    const int H = p->hidden;
    if ((H != 512 && H != 256 && H != 128) || p->obs_dim <= 0 || p->obs_dim > 48 || p->act_dim <= 0 || p->act_dim > 15) return fail(DL_E_INVAL, "dl_policy_pack: the packed layout exists for hidden = 512, 256, 128 (obs_dim <= 48, act_dim <= 15)");
    const int chunks = H * (H / 4) + 12 * H + (H / 4) * 16;
    hipLaunchKernelGGL(k_pack_policy, dim3((chunks + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, packed);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_policy_forward_packed(const dl_policy_params* p, const float* packed, const float* obs, int32_t n, const float* eps, uint64_t seed, uint64_t counter, int32_t index_base,
                             int32_t deterministic, float* actions, float* values, float* log_probs, void* stream) {
    if (!obs) return fail(DL_E_INVAL, "dl_policy_forward_packed: bad arguments");
    PolVnFuse none{};
    PolPacked pk{nullptr, nullptr, nullptr};
    if (packed) {
        if (!p || (p->hidden != 512 && p->hidden != 256 && p->hidden != 128)) return fail(DL_E_INVAL, "dl_policy_forward_packed: the packed layout exists for hidden = 512, 256, 128");
        pk.w2p = packed; pk.w1p = packed + (size_t)p->hidden * p->hidden; pk.whp = pk.w1p + (size_t)48 * p->hidden;
    }
    return policy_launch(p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, log_probs, none, stream, pk);
}
int dl_policy_forward_pair(const dl_policy_params* p, const float* packed, const float* obs, int32_t n, const float* eps, uint64_t seed, uint64_t counter, int32_t index_base,
                           int32_t deterministic, float* actions, float* values, float* log_probs, void* stream) {
    if (!p || !packed || !obs || !actions || !values || !log_probs || n <= 0) return fail(DL_E_INVAL, "dl_policy_forward_pair: bad arguments");
    const int H = p->hidden;
    if ((H != 512 && H != 256 && H != 128) || p->obs_dim <= 0 || p->obs_dim > 48 || p->act_dim <= 0 || p->act_dim > 15) return fail(DL_E_INVAL, "dl_policy_forward_pair: hidden = 512, 256 or 128, obs_dim <= 48, act_dim <= 15");
    PolPacked pk{packed, packed + (size_t)H * H, packed + (size_t)H * H + (size_t)48 * H};
#define DL_PAIR_LAUNCH(HH) hipLaunchKernelGGL(k_policy_forward_pair<HH>, dim3((n + 3) / 4), dim3(128), 0, (hipStream_t)stream, *p, obs, n, eps, seed, counter, index_base, deterministic, actions, values, log_probs, pk)
    if (H == 512) DL_PAIR_LAUNCH(512); else if (H == 256) DL_PAIR_LAUNCH(256); else DL_PAIR_LAUNCH(128);
#undef DL_PAIR_LAUNCH
    HIPCHK(hipGetLastError());
    return DL_OK;
}
static int rollout_policy_launches(dl_handle h, const dl_policy_params* pol, uint64_t seed, uint64_t counter0, int32_t index_base, const dl_vecnorm_state* vn, int32_t T,
                                   float* observations, float* actions, float* values, float* log_probs, float* rewards, uint8_t* episode_starts,
                                   float* next_obs, uint8_t* next_done, float* raw_obs, float* raw_rew, int deterministic, void* stream);
int dl_rollout_policy(dl_handle h, const dl_policy_params* pol, uint64_t seed, uint64_t counter0, int32_t index_base, const dl_vecnorm_state* vn, int32_t T,
                      float* observations, float* actions, float* values, float* log_probs, float* rewards, uint8_t* episode_starts,
                      float* next_obs, uint8_t* next_done, float* raw_obs, float* raw_rew, void* stream) {
    return rollout_policy_launches(h, pol, seed, counter0, index_base, vn, T, observations, actions, values, log_probs, rewards, episode_starts, next_obs, next_done, raw_obs, raw_rew, 0, stream);
}
static int rollout_policy_launches(dl_handle h, const dl_policy_params* pol, uint64_t seed, uint64_t counter0, int32_t index_base, const dl_vecnorm_state* vn, int32_t T,
                                   float* observations, float* actions, float* values, float* log_probs, float* rewards, uint8_t* episode_starts,
                                   float* next_obs, uint8_t* next_done, float* raw_obs, float* raw_rew, int deterministic, void* stream) {
    NEED(h);
    NOFAULT(h);
    if (!pol || !vn || T <= 0 || !observations || !actions || !values || !log_probs || !rewards || !episode_starts || !next_obs || !next_done || !raw_obs || !raw_rew)
        return fail(DL_E_INVAL, "dl_rollout_policy: bad arguments");
    const size_t n = (size_t)h->n, od = (size_t)h->obs_dim, ad = (size_t)h->act_dim;
    if (pol->obs_dim != h->obs_dim || pol->act_dim != h->act_dim) return fail(DL_E_INVAL, "dl_rollout_policy: the policy's observation / action sizes are not the environment's");
    // Three launches per control step: env step, moment reduction, and the policy forward of the NEXT step with the normalisation
    // of this step's outputs folded into its input stage (PolVnFuse).  Step 0 reads observations[0] as given; the outputs of the
    // last step are normalised by the stand-alone k_vn_apply.
    const uint8_t* prev_done = nullptr;
    PolPacked pk{nullptr, nullptr, nullptr};          // the policy's weights, packed once for the T forward passes of this rollout
    { const int rc = h->pack_policy(*pol, &pk, (hipStream_t)stream); if (rc) return rc; }
    for (int t = 0; t < T; t++) {
        const bool last = t + 1 == T;
        PolVnFuse vf{};
        if (t > 0) {
            vf.raw_obs = raw_obs; vf.raw_rew = raw_rew; vf.done = prev_done;
            vf.mean = vn->obs_mean; vf.var = vn->obs_var; vf.count = vn->obs_count; vf.ret = vn->ret; vf.ret_var = vn->ret_var; vf.ret_count = vn->ret_count;
            vf.obs_out = observations + t * n * od; vf.rew_out = rewards + (t - 1) * n;
            vf.eps = vn->eps; vf.clip_obs = vn->clip_obs; vf.clip_rew = vn->clip_rew; vf.flags = vn->flags;
        }
        int rc = policy_launch(pol, observations + t * n * od, (int32_t)n, nullptr, seed, counter0 + (uint64_t)t, index_base, deterministic,
                               actions + t * n * ad, values + t * n, log_probs + t * n, vf, stream, pk);
        if (rc) return rc;
        uint8_t* done = last ? next_done : episode_starts + (t + 1) * n;
        if ((rc = h->step(actions + t * n * ad, raw_obs, raw_rew, done, nullptr, nullptr, (hipStream_t)stream))) return rc;
        prev_done = done;
        if (last)
            rc = dl_vecnormalize_step(raw_obs, raw_rew, done, vn->obs_mean, vn->obs_var, vn->obs_count, vn->ret, vn->ret_mean, vn->ret_var, vn->ret_count, (int32_t)n, (int32_t)od,
                                      vn->gamma, vn->eps, vn->clip_obs, vn->clip_rew, vn->flags, next_obs, rewards + t * n, vn->workspace, stream);
        else
            rc = vn_reduce_launch(raw_obs, raw_rew, vn->obs_mean, vn->obs_var, vn->obs_count, vn->ret, vn->ret_mean, vn->ret_var, vn->ret_count, (int32_t)n, (int32_t)od, vn->gamma, vn->flags, vn->workspace, stream);
        if (rc) return rc;
    }
    return DL_OK;
}
int dl_rollout_persistent_ok(dl_handle h, const dl_policy_params* pol) {
    if (!h || !pol) return 0;
    return h->persistent_ok(pol->hidden, nullptr);
}
int dl_collect_rollouts(dl_handle h, const dl_policy_params* pol, uint64_t seed, uint64_t counter0, int32_t index_base, const dl_vecnorm_state* vn, int32_t T,
                        float* observations, float* actions, float* values, float* log_probs, float* rewards, uint8_t* episode_starts,
                        float* next_obs, uint8_t* next_done, float* raw_obs, float* raw_rew, int32_t mode, void* stream) {
    NEED(h);
    NOFAULT(h);
    if (mode & ~(DL_ROLLOUT_PERSISTENT | DL_ROLLOUT_MOMENTS_PER_ROLLOUT | DL_ROLLOUT_WORKGROUP_TILES | DL_ROLLOUT_DETERMINISTIC)) return fail(DL_E_INVAL, "dl_collect_rollouts: unknown mode bits");
    if ((mode & DL_ROLLOUT_WORKGROUP_TILES) && !(mode & DL_ROLLOUT_MOMENTS_PER_ROLLOUT)) return fail(DL_E_INVAL, "dl_collect_rollouts: DL_ROLLOUT_WORKGROUP_TILES selects the kernel of the per-rollout relaxation (the exact form always runs workgroup tiles)");
    if (!(mode & DL_ROLLOUT_PERSISTENT)) {
        if (mode & DL_ROLLOUT_MOMENTS_PER_ROLLOUT) return fail(DL_E_INVAL, "dl_collect_rollouts: per-rollout moments exist in the persistent form only");
        return rollout_policy_launches(h, pol, seed, counter0, index_base, vn, T, observations, actions, values, log_probs, rewards, episode_starts, next_obs, next_done, raw_obs, raw_rew,
                                       (mode & DL_ROLLOUT_DETERMINISTIC) != 0, stream);
    }
    if (!pol || !vn || T <= 0 || !observations || !actions || !values || !log_probs || !rewards || !episode_starts || !next_obs || !next_done || !raw_obs || !raw_rew)
        return fail(DL_E_INVAL, "dl_collect_rollouts: bad arguments");
    if (pol->obs_dim != h->obs_dim || pol->act_dim != h->act_dim) return fail(DL_E_INVAL, "dl_collect_rollouts: the policy's observation / action sizes are not the environment's");
    if (!pol->w1 || !pol->b1 || !pol->w2 || !pol->b2 || !pol->wa || !pol->ba || !pol->wv || !pol->bv || !pol->log_std) return fail(DL_E_INVAL, "dl_collect_rollouts: NULL parameter array");
    if (!vn->obs_mean || !vn->obs_var || !vn->obs_count || !vn->ret || !vn->ret_mean || !vn->ret_var || !vn->ret_count) return fail(DL_E_INVAL, "dl_collect_rollouts: NULL state array");
    return h->collect_persistent(*pol, seed, counter0, index_base, *vn, T, observations, actions, values, log_probs, rewards, episode_starts, next_obs, next_done, raw_obs, raw_rew,
                                 (mode & DL_ROLLOUT_MOMENTS_PER_ROLLOUT) ? ((mode & DL_ROLLOUT_WORKGROUP_TILES) ? 2 : 1) : 0, (mode & DL_ROLLOUT_DETERMINISTIC) != 0, (hipStream_t)stream);
}
int dl_gae(const float* rew, const float* val, const uint8_t* ep_start, const float* last_val, const uint8_t* last_done, float gamma, float lam, int32_t T, int32_t N, float* adv, float* ret, void* stream) {
    if (!rew || !val || !ep_start || !last_val || !last_done || !adv || !ret || T <= 0 || N <= 0) return fail(DL_E_INVAL, "dl_gae: bad arguments");
    hipLaunchKernelGGL(k_gae_fused, dim3((N + GAE_W - 1) / GAE_W), dim3(GAE_CH * GAE_W), 0, (hipStream_t)stream, rew, val, ep_start, last_val, last_done, gamma, lam, T, N, adv, ret);
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_adv_stats(const float* adv, int64_t n, double* out3, void* workspace, void* stream) {
    if (!adv || !out3 || !workspace || n <= 0) return fail(DL_E_INVAL, "dl_adv_stats: bad arguments");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 128) blocks = 128;         // few blocks: the last-arriver reduction pays one contended device-scope atomic per block
    if (blocks < 1) blocks = 1;
    double* work = (double*)workspace;
    hipLaunchKernelGGL(k_adv_stats, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, adv, (long long)n, out3, work, (unsigned*)(work + 2 * ADV_MAXBLOCKS));
    HIPCHK(hipGetLastError());
    return DL_OK;
}
int dl_adv_normalize(float* adv, int64_t n, const double* sums3, void* stream) {
    if (!adv || !sums3 || n <= 0) return fail(DL_E_INVAL, "dl_adv_normalize: bad arguments");
    hipLaunchKernelGGL(k_adv_normalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, adv, (long long)n, sums3);
    HIPCHK(hipGetLastError());
    return DL_OK;
}

}  // extern "C"

#ifdef DL_EXP_POL_PROF          // diagnostics build only (tools/diag_policy.py): the section stamps of dl_policy.hpp
extern "C" int dl_debug_pol_prof(long long* out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(dl::g_pol_prof), sizeof(long long) * 16) == hipSuccess ? 0 : -1; }
#endif
