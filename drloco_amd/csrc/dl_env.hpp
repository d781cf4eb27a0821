// dl_env.hpp -- per-walker environment logic around the dynamics core: action mapping, mocap
// cursor, observation, imitation reward, termination, Monitor statistics, reset / RSI.
// Restates (citations relative to /root/reference):
//   MimicEnv.step                      drloco/mujoco/mimic_env.py:60-126
//   _rescale_actions / mirror_action   :170-192 / :483-489
//   _get_obs / mirror_obs              :403-437 / :440-480
//   get_imitation_reward               :592-649
//   reset_model                        :526-572
//   StraightWalkingTrajectories.next   drloco/ref_trajecs/straight_walk_trajecs.py:141-159,322-348
//   get_random_init_state              :460-474
//   Monitor.step                       drloco/mujoco/monitor_wrapper.py:88-133
#pragma once

#include "dl_core.hpp"
#include "drloco_hip.h"

namespace dl {

// monitor words (double[MON_WORDS][N])
enum {
    MON_EP_LEN = 0, MON_NSTEPS, MON_RET, MON_LAST, MON_POS, MON_VEL, MON_COM, MON_TOR,
    MON_S_EP_LEN, MON_S_EP_RET, MON_S_MEAN_REW, MON_S_POS, MON_S_VEL, MON_S_COM, MON_S_TOR,
    MON_MOVED, MON_HAS, MON_POSREW, MON_VELREW, MON_COMREW,
    // what the host needs to keep Monitor's per-episode lists (monitor_wrapper.py:93,107,123,131-132): reference-cursor position at the
    // first step and at the end of the episode, the step's mean absolute torque, "shorter than 0.75 x the smoothed length" flag
    MON_INIT_POS, MON_ET_POS, MON_TOR_LAST, MON_DIFFICULT,
    // the FIRST episode a walker finished since the handle was created or all its walkers were reset (dl_reset without a mask), as
    // TrainingMonitor.eval_walking measures an episode (callback.py:300-317: duration incl. the terminal step, walked distance and reward sum
    // WITHOUT it): length, walked distance after the last non-terminal step, reward sum -- what a batched evaluation reads after ONE rollout call
    MON_FIRST_LEN, MON_FIRST_MOVED, MON_FIRST_RET, MON_WALKED_LAST,
    // ... counted by the record's OWN step / reward counters, which a reset of all walkers zeroes: MON_EP_LEN / MON_RET keep the reference Monitor's behaviour of carrying
    // over a reset (monitor_wrapper.py has no reset hook), so on a handle that was stepped before its reset they still hold the episode in flight
    MON_FIRST_CUR_LEN, MON_FIRST_CUR_RET,
    // control steps of this walker that took the reference's exception path (MujocoException -> reward 0, done, double reset: mimic_env.py:86-91) since dl_create
    MON_DIVERGED, MON_WORDS
};

constexpr int DL_DBG_EVALS = 40;      // evaluations per control step the diagnostics record (4 x frame_skip: 20 / 40)
template <typename T> struct DevState {
    T *qpos, *qvel, *warm;   // [NV][N]
    T* comz_off;             // [N]
    int32_t* cur;            // [DL_CUR_WORDS][N]
    double* walked;          // [N]
    double* mon;             // [MON_WORDS][N]
    int32_t* need_reset;     // [N]: 0 none, 1 auto reset after done, 2 double reset after a diverged step
    int32_t* inj_rsi;        // [2][N]: injected RSI draw for all later resets (step < 0: none); test hook
    T* work;                 // [4*NV][N] staging of the RK4 bookkeeping
    int32_t n;
    T* rnd;                  // [5][N] or NULL: per-walker mass scale, floor friction, push force on the torso (x, y, z)
    // push schedule (dl_set_push_schedule) or NULL: the push force of rnd acts on walker w during control step k iff
    // (k + push_phase[w]) % push_period < push_dur, k = push_step0 + step inside the launch (counted by the handle)
    const int32_t* push_phase;
    int32_t push_period, push_dur, push_step0;
    float* dbgf;             // [3*16 + DL_DBG_EVALS][N] or NULL: stage input (q, v, solver start) of the last evaluation with >= dbg_cap iterations; Newton iterations per evaluation of the last control step
    int dbg_cap;             // default: the iteration cap of the model (env DL_DEBUG_CAP_ITERS overrides; diagnostics)
    int32_t* dbg;            // [4][N] or NULL: solver diagnostics of the 16-lane step kernel (sum iters, max iters, sum rows, diverged)
    // fault word of the handle (host-pinned, written by the device with system scope; NULL = none): a wave of a split workgroup that leaves a
    // bounded poll by TIMEOUT ors its reason in (DL_FAULT_*); the host raises DL_E_FAULT at its next call (dl_fault_check)
    int32_t* fault;
    int32_t spin_dyn, spin_srv;   // polls before a dynamics wave / a constraint wave of a split workgroup gives up (dl_debug_set_spin_limit)
    // quirk Q4 (adjust_COM_Z_pos mutates the data set in place, base_ref_trajecs.py:126-127): [n_steps][N], the COM-z offset step s of walker w's copy of the data set
    // carries = the lowest-foot-site height of the last reset that landed on it (dl_get_ref_offsets); written at resets, read when a cursor rolls into another step
    T* zacc;
    int32_t strict_solver;   // dl_config.strict_solver (g_forward, one-wave form)
};

// quirk Q4 is reproduced unless dl_config.intended_semantics says otherwise
template <typename T> DL_HD bool q4_on(const DevCfg<T>& c) { return !(c.intended & DL_INTENDED_COMZ_PER_EPISODE); }
// zacc is written by one lane and read, possibly much later in the same launch, by others: both sides go to the L2 (agent scope), never through a stale L1 line
template <typename T> DL_HD T zacc_load(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}
template <typename T> DL_HD void zacc_store(T* p, T x) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = x;
#endif
}

DL_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// counter-based RSI stream keyed by (seed, global walker index, episode): invariant to sharding
template <typename T> DL_HD void rsi_draw(const DevCfg<T>& c, uint32_t genv, uint32_t episode, int& step, int& pos) {
    const uint64_t r = splitmix64(c.seed ^ splitmix64(((uint64_t)genv << 32) | episode));
    const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
    step = (int)(((uint64_t)lo * (uint64_t)c.n_steps) >> 32);
    const int len = c.step_off[step + 1] - c.step_off[step];
    pos = (int)(((uint64_t)hi * (uint64_t)len) >> 32);
}

// reference value `row` of mocap sample `col` (device table is sample-major, see DevCfg)
template <typename T> DL_HD T ref_at(const DevCfg<T>& c, int row, int col) { return c.table[(size_t)col * c.n_rows + row]; }

template <typename T, typename TP>
DL_HD void ref_lookup(const DevCfg<T>& c, const int32_t (&cur)[DL_CUR_WORDS], T comz_off, T (&qr)[TP::NV], T (&vr)[TP::NV]) {
    const int base = c.step_off[cur[DL_CUR_READ_STEP]] + cur[DL_CUR_POS];
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value;
        qr[j] = ref_at(c, j, base);
        vr[j] = ref_at(c, TP::NV + j, base);
    });
    if (cur[DL_CUR_HAS_DIST]) qr[0] += ref_at(c, 0, c.step_off[cur[DL_CUR_RSI_STEP] + 1] - 1);   // quirk Q1
    // quirk Q4: comz_off = the offset the step being read carries from the last reset onto it (the data set itself was re-anchored: whoever reads the step sees it);
    // DL_INTENDED_COMZ_PER_EPISODE: the reset's own offset, on the reset step only
    if (q4_on(c) || !cur[DL_CUR_HAS_DIST]) qr[2] -= comz_off;
}

template <typename T, typename TP> DL_HD void cursor_next(const DevCfg<T>& c, int32_t (&cur)[DL_CUR_WORDS]) {
    if constexpr (TP::ENV_KIND == 1) {
        // BaseReferenceTrajectories.next (drloco/ref_trajecs/base_ref_trajecs.py:95-103): wrap to the start
        cur[DL_CUR_POS] += c.stride;
        if (cur[DL_CUR_POS] >= c.step_off[1] - c.step_off[0] - 1) cur[DL_CUR_POS] = 0;
        return;
    }
    cur[DL_CUR_POS] += c.stride;
    const int rs = cur[DL_CUR_READ_STEP];
    const int dif = cur[DL_CUR_POS] - (c.step_off[rs + 1] - c.step_off[rs]) + 1;
    if (dif > 0) {
        if (cur[DL_CUR_I_STEP] >= c.n_steps - 1) cur[DL_CUR_I_STEP] = c.step_is_left[cur[DL_CUR_I_STEP]] ? 0 : 1;
        else { cur[DL_CUR_I_STEP] += 1; cur[DL_CUR_COUNT] += 1; }
        cur[DL_CUR_HAS_DIST] = 1;
        cur[DL_CUR_READ_STEP] = cur[DL_CUR_I_STEP];
        cur[DL_CUR_POS] = dif;
    }
}

template <typename T, typename TP>
DL_HD void get_obs(const DevCfg<T>& c, const int32_t (&cur)[DL_CUR_WORDS], const T (&q)[TP::NV], const T (&v)[TP::NV], float (&o)[TP::OBS]) {
    if constexpr (TP::ENV_KIND == 1) {
        // MimicWalker165cm65kg: 4 x (phase angle, phase radius) from joint phase plots
        // (mimic_env.py:330-401), 2 desired velocities = mean reference pelvis x / z velocity over the
        // next 0.5 s (loco3d_trajecs.py:51-97, through float64 prefix sums), qpos[1:], qvel
        static_for<4>([&](auto ki) {
            constexpr int kk = ki.value, j = TP::phase_joint(kk);
            o[2 * kk] = (float)(dl_atan2(v[j], -q[j]) * T(0.31830988618379067154));
            o[2 * kk + 1] = (float)(dl_sqrt(q[j] * q[j] + v[j] * v[j]) / T(5));
        });
        const int L = c.step_off[1] - c.step_off[0], pos = cur[DL_CUR_POS];
        const int end = pos + 250 < L - 1 ? pos + 250 : L - 1;
        const double cnt = (double)(end - pos);
        o[8] = (float)((c.pref[end] - c.pref[pos]) / cnt);
        o[9] = (float)((c.pref[(size_t)c.total_len + 1 + end] - c.pref[(size_t)c.total_len + 1 + pos]) / cnt);
        static_for<TP::NV - 1>([&](auto ji) { o[10 + ji.value] = (float)q[ji.value + 1]; });
        static_for<TP::NV>([&](auto ji) { o[9 + TP::NV + ji.value] = (float)v[ji.value]; });
        return;
    }
    const int rs = cur[DL_CUR_READ_STEP];
    T raw[TP::OBS];
    raw[0] = T(cur[DL_CUR_POS]) / T(c.step_off[rs + 1] - c.step_off[rs]);
    const int iv = cur[DL_CUR_I_STEP] - cur[DL_CUR_COUNT] + 1;
    raw[1] = c.step_vel[iv > 0 ? iv : 0];
    static_for<TP::NV - 1>([&](auto ji) { raw[2 + ji.value] = q[ji.value + 1]; });
    static_for<TP::NV>([&](auto ji) { raw[1 + TP::NV + ji.value] = v[ji.value]; });
    const bool mirr = c.mirror_policy && c.step_is_left[cur[DL_CUR_I_STEP]];
    static_for<TP::OBS>([&](auto ki) {
        constexpr int kk = ki.value;
        const T plain = raw[kk];
        const T mir = TP::obs_neg(kk) ? -raw[TP::obs_perm(kk)] : raw[TP::obs_perm(kk)];
        o[kk] = (float)(mirr ? mir : plain);
    });
}

template <typename T, typename TP>
DL_HD T imitation_reward(const DevCfg<T>& c, const int32_t (&cur)[DL_CUR_WORDS], T comz_off, const T (&q)[TP::NV], const T (&v)[TP::NV], T (&terms)[3]) {
    T qr[TP::NV], vr[TP::NV];
    ref_lookup<T, TP>(c, cur, comz_off, qr, vr);
    T sp = T(0), sv = T(0), sc = T(0);
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value;
        const T dq = q[j] - qr[j], dv = v[j] - vr[j];
        if constexpr (j < 3) sc += dq * dq; else { sp += dq * dq; sv += dv * dv; }
    });
    terms[0] = dl_exp(T(-3) * sp); terms[1] = dl_exp(T(-0.05) * sv); terms[2] = dl_exp(T(-16) * sc);
    return (c.rew_w[0] * terms[0] + c.rew_w[1] * terms[1] + c.rew_w[2] * terms[2]) * c.rew_scale;
}

DL_HD void mon_smooth(double* mon, int n, int i, int word, int bit, double x, double alpha) {
    const unsigned has = (unsigned)mon[(size_t)MON_HAS * n + i];
    double& s = mon[(size_t)word * n + i];
    if (!(has & (1u << bit))) { s = x; mon[(size_t)MON_HAS * n + i] = (double)(has | (1u << bit)); }
    else s = alpha * x + (1 - alpha) * s;
}

// cur_pos: refs._pos after this step's refs.next() (before any reset), as Monitor.step reads it
DL_HD void monitor_step(double* mon, int n, int i, double rew, bool done, const double (&terms)[3], double tor, double walked, int cur_pos, bool exc = false) {
    auto W = [&](int w) -> double& { return mon[(size_t)w * n + i]; };
    if (exc) W(MON_DIVERGED) += 1;
    W(MON_FIRST_CUR_LEN) += 1; W(MON_FIRST_CUR_RET) += rew;
    if (W(MON_EP_LEN) == 0) W(MON_INIT_POS) = (double)cur_pos;      // monitor_wrapper.py:91-93
    W(MON_TOR_LAST) = tor;
    W(MON_EP_LEN) += 1; W(MON_NSTEPS) += 1; W(MON_RET) += rew; W(MON_LAST) = rew;
    W(MON_POS) += terms[0]; W(MON_VEL) += terms[1]; W(MON_COM) += terms[2]; W(MON_TOR) += tor;
    if (done) {
        const double len = W(MON_EP_LEN);
        mon_smooth(mon, n, i, MON_S_MEAN_REW, 0, (W(MON_RET) - W(MON_LAST)) / (len - 1), 0.9);
        mon_smooth(mon, n, i, MON_S_POS, 1, W(MON_POS) / W(MON_NSTEPS), 0.9);
        mon_smooth(mon, n, i, MON_S_VEL, 2, W(MON_VEL) / W(MON_NSTEPS), 0.9);
        mon_smooth(mon, n, i, MON_S_COM, 3, W(MON_COM) / W(MON_NSTEPS), 0.9);
        mon_smooth(mon, n, i, MON_S_EP_RET, 4, W(MON_RET), 0.25);
        mon_smooth(mon, n, i, MON_S_EP_LEN, 5, len, 0.75);
        W(MON_ET_POS) = (double)cur_pos;                                                  // :105-107
        W(MON_DIFFICULT) = (len < W(MON_S_EP_LEN) * 0.75) ? 1.0 : 0.0;                    // :122-123 (after the smoothing update)
        mon_smooth(mon, n, i, MON_S_TOR, 6, W(MON_TOR) / len, 0.75);
        W(MON_MOVED) = walked;
        if (W(MON_FIRST_LEN) == 0) { W(MON_FIRST_LEN) = W(MON_FIRST_CUR_LEN); W(MON_FIRST_MOVED) = W(MON_WALKED_LAST); W(MON_FIRST_RET) = W(MON_FIRST_CUR_RET) - rew; }
        W(MON_FIRST_CUR_LEN) = 0; W(MON_FIRST_CUR_RET) = 0;
        W(MON_EP_LEN) = 0; W(MON_RET) = 0; W(MON_TOR) = 0;
    }
    W(MON_WALKED_LAST) = done ? 0.0 : walked;          // the walked distance after this step if the episode goes on (the next step may be terminal)
}

// one control step of walker i.  Writes obs only for walkers that continue; finished walkers get
// term_obs and need_reset = 1 (their obs row is written by env_reset_lane).
template <typename T, typename TP>
DL_HD void env_step_lane(const DL_CONST DevModel<T, TP>& m, const DevCfg<T>& c, const LaneMem<T>& mem, const DevState<T>& st, int i,
                         const float* actions, float* obs, float* rew, uint8_t* done, float* term_obs, float* rew_terms,
                         const T* inj_q, const T* inj_v, const int32_t* inj_flags) {
    const int n = st.n;
    T q[TP::NV], v[TP::NV], warm[TP::NV], ctrl[TP::NU];
    int32_t cur[DL_CUR_WORDS];
    static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; q[j] = st.qpos[(size_t)j * n + i]; v[j] = st.qvel[(size_t)j * n + i]; warm[j] = st.warm[(size_t)j * n + i]; });
    static_for<DL_CUR_WORDS>([&](auto ki) { cur[ki.value] = st.cur[(size_t)ki.value * n + i]; });
    // _rescale_actions, then mirror_action with the cursor BEFORE refs.next()
    T raw[TP::NU];
    static_for<TP::NU>([&](auto ai) {
        constexpr int a = ai.value;
        const T x = dl_clamp((T)actions[(size_t)i * TP::NU + a], T(-1), T(1));
        raw[a] = x > T(0) ? x * m.ctrl_hi[a] : dl_abs(x) * m.ctrl_lo[a];
    });
    const bool mirr_a = TP::ENV_KIND == 0 && c.mirror_policy && c.step_is_left[cur[DL_CUR_I_STEP]];
    static_for<TP::NU>([&](auto ai) {
        constexpr int a = ai.value;
        const T mir = TP::act_neg(a) ? -raw[TP::act_perm(a)] : raw[TP::act_perm(a)];
        ctrl[a] = mirr_a ? mir : raw[a];
    });
    bool exc = false;
    const int flag = inj_flags ? inj_flags[i] : 0;   // test hook: 1 = inject end state, 2 = inject exception
    if (flag == 2) exc = true;
    else if (flag == 1) {
        static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; q[j] = inj_q[(size_t)j * n + i]; v[j] = inj_v[(size_t)j * n + i]; });
    } else {
        const GlobalMem<T> gw{st.work + i, n};
#pragma unroll 1
        for (int kf = 0; kf < m.frame_skip && !exc; kf++) exc = mj_step_rk4<T, TP>(m, mem, gw, q, v, ctrl, warm);
    }
    // everything that is only needed after the physics is loaded here (nothing but q, v, warm, ctrl
    // and the cursor stays live across the forward evaluations)
    T comz = st.comz_off[i];
    double walked = st.walked[i];
    T tor = T(0);
    static_for<TP::NU>([&](auto ai) { constexpr int a = ai.value; tor += dl_abs(dl_clamp(ctrl[a], m.force_lo[a], m.force_hi[a])); });
    const double tor_mean = (double)tor / TP::NU;
    double terms[3] = {st.mon[(size_t)MON_POSREW * n + i], st.mon[(size_t)MON_VELREW * n + i], st.mon[(size_t)MON_COMREW * n + i]};
    float r;
    bool dn;
    if (exc) {
        // mimic_env.py:86-91: obs = self.reset(); return obs, 0, True, {}  -- then the vec env resets again
        r = 0.0f; dn = true; walked = 0;
        terms[0] = terms[1] = terms[2] = 1.0;    // the in-step reset() re-evaluated the reward terms (:562)
        st.need_reset[i] = 2;
    } else {
        cursor_next<T, TP>(c, cur);
        if (q4_on(c)) comz = zacc_load(st.zacc + (size_t)cur[DL_CUR_READ_STEP] * n + i);          // the offset of the step the cursor reads NOW
        float o[TP::OBS];
        get_obs<T, TP>(c, cur, q, v, o);
        cur[DL_CUR_EP_DUR] += 1;
        const T vx = dl_clamp(v[0], T(-5.5), T(5.5)), vy = dl_clamp(v[1], T(-5.5), T(5.5));
        walked += (double)dl_sqrt(vx * vx + vy * vy) * (double)c.inv_ctrl_freq;
        const bool timeout = cur[DL_CUR_EP_DUR] >= c.ep_dur_max;
        dn = (q[2] < c.com_z_min) || timeout;
        if (dn) r = timeout ? 0.0f : -0.0f;      // _get_ET_reward: ep_rews is always empty -> +-0
        else {
            T tt[3];
            r = (float)(imitation_reward<T, TP>(c, cur, comz, q, v, tt) + c.alive_bonus);
            terms[0] = (double)tt[0]; terms[1] = (double)tt[1]; terms[2] = (double)tt[2];
        }
        float* dst = dn ? term_obs : obs;
        if (dst) static_for<TP::OBS>([&](auto ki) { dst[(size_t)i * TP::OBS + ki.value] = dl_sat_out(o[ki.value]); });
        if (dn) st.need_reset[i] = 1;
    }
    monitor_step(st.mon, n, i, (double)r, dn, terms, tor_mean, walked, cur[DL_CUR_POS], exc);
    st.mon[(size_t)MON_POSREW * n + i] = terms[0]; st.mon[(size_t)MON_VELREW * n + i] = terms[1]; st.mon[(size_t)MON_COMREW * n + i] = terms[2];
    if (rew_terms) { rew_terms[3 * (size_t)i] = (float)terms[0]; rew_terms[3 * (size_t)i + 1] = (float)terms[1]; rew_terms[3 * (size_t)i + 2] = (float)terms[2]; }
    rew[i] = r == r ? r : 0.0f;
    done[i] = dn ? 1 : 0;
    static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; st.qpos[(size_t)j * n + i] = q[j]; st.qvel[(size_t)j * n + i] = v[j]; st.warm[(size_t)j * n + i] = warm[j]; });
    static_for<DL_CUR_WORDS>([&](auto ki) { st.cur[(size_t)ki.value * n + i] = cur[ki.value]; });
    st.walked[i] = walked;
}

// MujocoEnv.reset + reset_model for walker i, `nrep` times in a row (2 after a diverged step:
// the first reset's observation is the terminal observation).
template <typename T, typename TP>
DL_HD void env_reset_lane(const DL_CONST DevModel<T, TP>& m, const DevCfg<T>& c, const LaneMem<T>& mem, const DevState<T>& st, int i, int nrep,
                          const int32_t* init_step, const int32_t* init_pos, float* obs, float* term_obs, int eval_mode) {
    const int n = st.n;
    T q[TP::NV], v[TP::NV], warm[TP::NV], zero_u[TP::NU], zero_w[TP::NV];
    int32_t cur[DL_CUR_WORDS];
    static_for<DL_CUR_WORDS>([&](auto ki) { cur[ki.value] = st.cur[(size_t)ki.value * n + i]; });
    static_for<TP::NU>([&](auto ai) { zero_u[ai.value] = T(0); });
    static_for<TP::NV>([&](auto ji) { zero_w[ji.value] = T(0); });
    T comz = T(0);
    float o[TP::OBS];
#pragma unroll 1
    for (int rep = 0; rep < nrep; rep++) {
        int s, p, read = -1;
        if (init_step) { s = init_step[i]; p = init_pos[i]; }
        else if (eval_mode && TP::ENV_KIND == 1) { s = 0; p = 0; }   // base get_deterministic_init_state(0 %)
        else if (eval_mode) {
            // _get_deterministic_init_state (straight_walk_trajecs.py:237-265) incl. quirk Q3
            s = cur[DL_CUR_EVAL_K];
            p = (int)(0.75 * (double)(c.step_off[s + 1] - c.step_off[s]));
            read = (c.intended & DL_INTENDED_EVAL_OWN_STEP) ? s : 0;
            cur[DL_CUR_EVAL_K] = (s + 1 >= 20) ? 0 : s + 1;
        }
        else if (st.inj_rsi && st.inj_rsi[i] >= 0) { s = st.inj_rsi[i]; p = st.inj_rsi[(size_t)n + i]; }
        else rsi_draw(c, (uint32_t)(c.env_index_base + i), (uint32_t)cur[DL_CUR_EPISODE], s, p);
        cur[DL_CUR_EPISODE] += 1;
        cur[DL_CUR_EP_DUR] = 0;
        if (c.intended & DL_INTENDED_COUNT_PER_EPISODE) cur[DL_CUR_COUNT] = 1;
        cur[DL_CUR_I_STEP] = s; cur[DL_CUR_RSI_STEP] = s; cur[DL_CUR_READ_STEP] = read >= 0 ? read : s; cur[DL_CUR_POS] = p; cur[DL_CUR_HAS_DIST] = 0;
        ref_lookup<T, TP>(c, cur, T(0), q, v);
        // lowest foot-sole corner onto the floor (mimic_env.py:547-559)
        {
            Kin<T, TP> k;
            kinematics<T, TP>(m, q, k);
            T low = T(1e30);
            static_for<TP::NS>([&](auto si) {
                constexpr int sidx = si.value, b = TP::site_body(sidx);
                const V3<T> sp = body_point<T, TP>(k, b, m.site_pos[sidx]);
                low = dl_min(low, k.rootz + sp.z);
            });
            q[2] -= low;
            comz = low;
            // quirk Q4 (adjust_COM_Z_pos): the step's row of this walker's data set is re-anchored in place.  The reference subtracts `low` measured at the ALREADY shifted
            // row; the foot height follows the root's z one to one, so the row ends at (pristine - low measured at the pristine row): the offset is replaced, not summed
            if (q4_on(c)) zacc_store(st.zacc + (size_t)cur[DL_CUR_READ_STEP] * n + i, low);
        }
        // set_state -> mj_forward: qacc of the initial state seeds the warmstart
        (void)forward_io<T, TP>(m, mem, q, v, zero_u, zero_w, warm);
        cursor_next<T, TP>(c, cur);
        get_obs<T, TP>(c, cur, q, v, o);
        if (nrep == 2 && rep == 0 && term_obs) static_for<TP::OBS>([&](auto ki) { term_obs[(size_t)i * TP::OBS + ki.value] = o[ki.value]; });
    }
    if (obs) static_for<TP::OBS>([&](auto ki) { obs[(size_t)i * TP::OBS + ki.value] = o[ki.value]; });
    static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; st.qpos[(size_t)j * n + i] = q[j]; st.qvel[(size_t)j * n + i] = v[j]; st.warm[(size_t)j * n + i] = warm[j]; });
    static_for<DL_CUR_WORDS>([&](auto ki) { st.cur[(size_t)ki.value * n + i] = cur[ki.value]; });
    st.comz_off[i] = comz;
    st.walked[i] = 0;
    // reset_model's sanity check evaluates the imitation reward at the init state (:562): all terms are 1
    st.mon[(size_t)MON_POSREW * n + i] = 1.0; st.mon[(size_t)MON_VELREW * n + i] = 1.0; st.mon[(size_t)MON_COMREW * n + i] = 1.0;
    st.need_reset[i] = 0;
}

}  // namespace dl
