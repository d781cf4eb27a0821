// dl_group_env.hpp -- the bodies of the 16-lanes-per-walker kernels (one wave = four walkers): a forward evaluation
// and one or more control steps of the environment (action map, frame_skip x RK4 mj_step through g_forward, cursor /
// observation / reward / termination / Monitor, and the vec-env auto reset of finished walkers in the same launch).
// Written as device functions of (lane, walker block) so that dl_kernels.hip wraps them into __global__ kernels and
// tests/host_emu runs the very same source on the host (a wave as 64 fibers).
// Restates, per walker (citations relative to /root/reference): MimicEnv.step drloco/mujoco/mimic_env.py:60-126,
// _rescale_actions / mirror_action :170-192 / :483-489, _get_obs / mirror_obs :403-437 / :440-480 (165 cm walker:
// joint phase features :330-401, desired velocities drloco/ref_trajecs/loco3d_trajecs.py:51-97), get_imitation_reward
// :592-649, reset_model :526-572, refs.next drloco/ref_trajecs/straight_walk_trajecs.py:141-159,322-348 /
// base_ref_trajecs.py:95-103, Monitor.step drloco/mujoco/monitor_wrapper.py:88-133.
#pragma once

#include "dl_group.hpp"

#ifndef DL_EXP_EXTRAP
#define DL_EXP_EXTRAP 0          // experiment switch: quadratic (three-point) extrapolation of the solver's start point in RK4 stages 1 and 3, weight DL_EXP_EXTRAP_W
#endif
#ifndef DL_EXTRAP_BETA
#define DL_EXTRAP_BETA 1.0          // weight of the linear extrapolation of the solver's start point in RK4 stages 1 and 3 (lane dofs; 1 = the next stage lies half a step further)
#endif
#ifndef DL_EXP_EXTRAP_W
#define DL_EXP_EXTRAP_W 1.0
#endif

namespace dl {

// Workgroups are dealt round-robin to the 8 XCDs (workgroup i runs on XCD i % 8), each with its own L2.  The SoA
// state rows put 16 consecutive walkers into one 64-byte line, i.e. 4 consecutive walker groups share their lines:
// map workgroup i to walker group (i % 8) * (G / 8) + i / 8 so that neighbouring groups run on the SAME XCD and a
// line is fetched into one L2 only.  (G not a multiple of 8: identity.)
__device__ __forceinline__ int g_block_of_workgroup(int wg, int nwg) {
    constexpr int XCDS = 8;
    if (nwg % XCDS) return wg;
    return (wg % XCDS) * (nwg / XCDS) + wg / XCDS;
}

// value of dof d (compile time) of the walker, identical in the 16 lanes of its row: a replicated root translation or the
// row broadcast of the owning lane
template <typename TP, int d, typename T> __device__ __forceinline__ T g_dof_value(T lane_val, const GX<T, GD<TP>::NX>& xs) {
    if constexpr (d < GD<TP>::NX) return xs.x[d]; else return rbcast<d - GD<TP>::NX>(lane_val);
}

template <typename T, typename TP> __device__ __forceinline__ void g_load_walk(const DL_CONST GModel<T, TP>* m, const DevState<T>& st, int w, GWalk<T>& wk) {
    wk = GWalk<T>{T(1), m->floor_friction, mk<T>(0, 0, 0), false};
    if (st.rnd) {
        const size_t ws = (size_t)w, n = (size_t)st.n;
        wk.mscale = st.rnd[ws]; wk.floor_mu = st.rnd[n + ws];
        wk.push = mk<T>(st.rnd[2 * n + ws], st.rnd[3 * n + ws], st.rnd[4 * n + ws]);
        wk.pushed = wk.push.x != T(0) || wk.push.y != T(0) || wk.push.z != T(0);
    }
}

// forward dynamics of the walkers' current state (dl_forward): `wblock` = walker block of this wave, `tim` (TIMED) = [8][nwg]
template <typename T, typename TP, bool TIMED = false>
__device__ __forceinline__ void g_wave_forward(int lane, int wblock, int wg, int nwg, DL_LDS T* smem, const GModel<T, TP>* __restrict__ gm, const DevState<T>& st, const T* ctrl,
                                               T* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter, long long* tim) {
    using D = GD<TP>;
    constexpr int NL = D::NL, NX = D::NX;
    const int grp = lane >> 4, j = lane & 15, n = st.n;
    const int w = wblock * GW + grp;
    const bool valid = w < n;
    const int wi = valid ? w : n - 1;          // out-of-range rows redo the last walker (keeps the wave uniform)
    const DL_CONST GModel<T, TP>* m = (const DL_CONST GModel<T, TP>*)gm;
    const GLane<T, D::NPASS> ln = m->lanes[j];     // this lane's record of the model block (built on the host by g_load_lane)
    GConst<T, TP> cst;
    g_load_const<T, TP>(*m, cst);
    GWalk<T> wk;
    g_load_walk<T, TP>(m, st, wi, wk);
    GCtx<T, TP> g{smem + (size_t)grp * GLds<TP>::TOTAL, m, j, &ln, &cst, &wk, smem + (size_t)grp * GLds<TP>::TOTAL + GLds<TP>::MM, nullptr, nullptr, nullptr, 0,
                  smem + (size_t)grp * GLds<TP>::TOTAL + GLds<TP>::BFR, smem + (size_t)grp * GLds<TP>::TOTAL + GLds<TP>::MISC};
    g.strict = st.strict_solver;
    T q = T(0), v = T(0), wm = T(0), force = T(0);
    if (j < NL) {
        const size_t o = (size_t)(j + NX) * n + wi;
        q = st.qpos[o]; v = st.qvel[o]; wm = st.warm[o];
        const int a = ln.act;
        if (a >= 0) {
            const T u = dl_clamp(ctrl ? ctrl[(size_t)a * n + wi] : T(0), ln.ctrl_lo, ln.ctrl_hi);
            force = ln.gear * dl_clamp(u, ln.force_lo, ln.force_hi);
        }
    }
    GX<T, NX> qx, vx, wx, ax;
    static_for<NX>([&](auto ti) { constexpr int t = ti.value; const size_t o = (size_t)t * n + wi; qx.x[t] = st.qpos[o]; vx.x[t] = st.qvel[o]; wx.x[t] = st.warm[o]; });
    int nc, ne, ni;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    GLaneTopo<T> lt;
    g_lane_topo<T, TP>(j, lt);
    const T a = g_forward<T, TP, TIMED>(g, lt, grp, q, v, force, wm, qx, vx, wx, ax, nc, ne, ni, tacc);
    if constexpr (TIMED) { if (lane == 0) for (int k = 0; k < 8; k++) tim[(size_t)k * nwg + wg] = tacc[k]; }
    if (valid && j < NL) qacc[(size_t)(j + NX) * n + w] = a;
    if (valid && j == 0) {
        static_for<NX>([&](auto ti) { qacc[(size_t)ti.value * n + w] = ax.x[ti.value]; });
        if (ncon) ncon[w] = nc; if (nefc) nefc[w] = ne; if (niter) niter[w] = ni;
    }
}

// The constraint wave of a split workgroup: serves the dynamics wave that owns the same four walkers (same LDS regions).  Per request
// (q and the solver's start point B v + a in the mailbox, body frames and root height in the walker's LDS region) it runs the first half of
// g_forward's constraint stage -- collision, contact records, limit and contact rows, all in the walker's LDS region -- and reports
// (ncon, nlim, the lanes' limit rows); the contact Jacobians are the dynamics wave's.
// Waits are bounded polls (never a hung GPU); a wait that runs out sets the handle's fault word and ends this wave -- the dynamics wave's
// next request then runs out too, its walkers take the exception path and the host raises DL_E_FAULT (dl_fault_check).
template <typename T, typename TP>
__device__ __forceinline__ void g_constraint_server(int lane, int wblock, DL_LDS T* smem, const GModel<T, TP>* __restrict__ gm, const DevState<T>& st,
                                                    const float* __restrict__ actions_all = nullptr, int nsteps = 0) {
    using D = GD<TP>;
    using Ld = GLds<TP>;
    using Sp = GSplit<TP>;
    constexpr int NX = D::NX;
    static_assert(!DL_JAC_ON_PARTNER || NX == 0, "the partner-side contact Jacobians (experiment) exist for the lane-only walker");
    const int grp = lane >> 4, j = lane & 15, n = st.n;
    const int w0 = wblock * GW + grp;
    const int w = w0 < n ? w0 : n - 1;
    const DL_CONST GModel<T, TP>* m = (const DL_CONST GModel<T, TP>*)gm;
    const GLane<T, D::NPASS> ln = m->lanes[j];
    GConst<T, TP> cst;
    g_load_const<T, TP, false>(*m, cst);          // (fetch-at-use policy: this wave's registers hold the detection of the next evaluation across its waits)
    GWalk<T> wk;
    g_load_walk<T, TP>(m, st, w, wk);
    DL_LDS T* wb = smem + (size_t)grp * Sp::TOTAL;
    GCtx<T, TP> g{wb, m, j, &ln, &cst, &wk, wb + Sp::MMX, wb + Sp::MB, smem + Sp::MB, st.fault, st.spin_srv, wb + Sp::BFRX, wb + Sp::RZX};
    GLaneTopo<T> lt;
    g_lane_topo<T, TP>(j, lt);
    volatile DL_LDS int* flags = (volatile DL_LDS int*)g.mbox0;
    int seq = 0;
    // multi-step launches: the action row of the NEXT control step is a cold line of a tape that streams through (67 MB per rollout at 4096
    // walkers) -- the dynamics wave would wait out an HBM round trip at every step start with nothing to switch to.  This wave is idle most of
    // the time: after the first evaluation of a step it touches the next step's row of its four walkers (the value is thrown away; the line
    // then sits in this CU's L1 / the XCD's L2).  The step index is counted from the requests (4 per mj_step); steps without evaluations
    // (exception path) make it lag -- it is a prefetch, it only has to stay inside the tape.
    const int evals_per_step = 4 * m->frame_skip;
    float pf_sink = 0.0f;
#ifdef DL_EXP_SPLIT_PROF
    long long srv_busy = 0;
#endif
    // what this wave holds in advance, for the configuration its partner announced with the last request: body frames, joint axes and mass matrix in LDS
    // (GSplit::BFRX / AXX, the mirror block), what the dynamics wave needs to know of the constraints in the mailbox (the lanes' limit rows, the counts) and
    // the configuration half of the constraints in registers (det)
    GDet<T, TP> det;
    GKin<T> kin{};
    GX<T, NX> rq_qx{}, rq_qnx{};          // replicated root translations: the request's configuration (command 2) and NEXT configuration (uniform over a walker's row)
    // One code site per job, wave-uniform (scalar) control: a pass of the loop is  [wait for a request]  ->  [commit + post the rows]  ->  [geometry of a
    // configuration + post that it is there].  A command-2 request (nothing usable in advance) takes two passes: geometry of ITS configuration first
    // (posted as MB_MOK), then the commit and the look-ahead.
    bool owe_commit = false;          // a command-2 request whose geometry is done and whose rows are still to be committed
    T rq_q = T(0), rq_x0 = T(0), rq_qn = T(0);          // the request's words: this evaluation's configuration (command 2), the solver's start point, the NEXT configuration
    int cmd = 0;
#ifdef DL_EXP_SPLIT_PROF
    long long tsrv0 = 0;
#endif
    // The first evaluation of a launch: its configuration is the walkers' state in memory (unless a test injects another one), so the geometry starts NOW,
    // next to the dynamics wave's prologue (state, model and action loads), instead of with the first request: posted as the look-ahead of "request 0".
    bool boot = true;
    for (;;) {
        if (boot) {
            cmd = 3;
#ifdef DL_EXP_SPLIT_PROF
            tsrv0 = DL_CLOCK();
#endif
        } else if (!owe_commit) {
            // (sequence number, command) in one 8-byte read: the partner stores the command first, LDS operations of a wave complete in order
            static_assert(Sp::MB_CMD == Sp::MB_CMDSEQ + 1 && Sp::MB_CMDSEQ % 2 == 0, "the command word pair is one aligned 8-byte word");
            int cur = seq, it = 0;
            for (;;) {
                const long long wd = *(volatile DL_LDS long long*)(flags + Sp::MB_CMDSEQ);
                cur = DL_UNIFORM((int)(wd & 0xffffffffll)); cmd = DL_UNIFORM((int)(wd >> 32));
                if (cur != seq || it >= g.spin_limit) break;
                DL_SLEEP(); it++;
            }
            if (cur == seq) {         // timeout: the partner never asked and never released -- say so (the partner's next request then times out as well)
                if (lane == 0 && g.fault) DL_FAULT_OR(g.fault, DL_FAULT_SRV_TIMEOUT);
                break;
            }
            if (cmd == 0) break;          // released
#ifdef DL_EXP_SPLIT_PROF
            tsrv0 = DL_CLOCK();
#endif
            DL_WG_ACQUIRE();
            seq = cur;
            g_sync<T>();
            // The request is consumed HERE, before anything is posted: the dynamics wave overwrites these mailbox words with its next request, and it can start
            // that as soon as this request's rows are posted (a solve of walkers in the air takes a few thousand cycles -- less than a cache miss of this wave).
            rq_x0 = g.mbox[Sp::MB_X0 + j]; rq_qn = g.mbox[Sp::MB_QN + j];
            if (cmd == 2) rq_q = g.mbox[Sp::MB_Q + j];
            g_pin(rq_x0); g_pin(rq_qn); g_pin(rq_q);
            if constexpr (NX > 0) {
                const Q4<T> a = ld4(g.mbox + Sp::MB_QNX), b = ld4(g.mbox + Sp::MB_QX);
                const T an[3] = {a.a, a.b, a.c}, bn[3] = {b.a, b.b, b.c};
                static_for<NX>([&](auto ti) { constexpr int t = ti.value; rq_qnx.x[t] = an[t]; rq_qx.x[t] = bn[t]; g_pin(rq_qnx.x[t]); g_pin(rq_qx.x[t]); });
            }
        }
        T qg;                 // the configuration whose geometry this pass computes
        GX<T, NX> qxg{};
        int post;             // ... and the flag that says it is there
        if (cmd == 3) {
            qg = j < D::NL ? st.qpos[(size_t)(j + NX) * n + w] : T(0); post = Sp::MB_PRE;
            static_for<NX>([&](auto ti) { constexpr int t = ti.value; qxg.x[t] = st.qpos[(size_t)t * n + w]; });
            boot = false;
        } else if (cmd == 1 || owe_commit) {
            // ---- commit: contact records and rows of this evaluation from the detection in registers (the limit rows take the solver's start point):
            // stores only, then the flag
            g_commit_constraints<T, TP>(g, det, rq_x0);
#if DL_JAC_ON_PARTNER
            { const T x0x[1] = {T(0)}; g_contact_jacobians<T, TP>(g, lt, kin, det.ncon, x0, x0x); }
#endif
            DL_WG_RELEASE();
            if (lane == 0) flags[Sp::MB_DONESEQ] = seq;
            DL_WAKE();
#if defined(DL_EXP_SPLIT_PROF) && DL_EXP_SPLIT_PROF == 3          // (3: this wave's time from seeing a request to posting its rows, instead of its whole busy time)
            srv_busy += DL_CLOCK() - tsrv0;
#endif
            if (DL_PREFETCH_ACTIONS && actions_all && (seq - 1) % evals_per_step == 0) {
                const int next = (seq - 1) / evals_per_step + 1;
                if (next < nsteps && j < TP::NU) pf_sink += actions_all[((size_t)next * n + w) * TP::NU + j];
            }
            // in advance, while the dynamics wave solves: the NEXT evaluation's configuration.  (After command 2 the dynamics wave may still be reading what
            // the previous pass computed: it says when it has it, MB_MFREE.  After command 1 it had taken everything before it sent the request.)
            if (owe_commit) {
                int freed = 0;
                for (int k2 = 0; !(freed = DL_UNIFORM((int)(flags[Sp::MB_MFREE] == seq))) && k2 < g.spin_limit; k2++) DL_SLEEP();
                if (!freed) {
                    if (lane == 0 && g.fault) DL_FAULT_OR(g.fault, DL_FAULT_SRV_TIMEOUT);
                    break;
                }
                DL_WG_ACQUIRE();
                owe_commit = false;
            }
#ifdef DL_EXP_R4_LATE_READ          // TEST SWITCH (tests/host_emu only): the round-4 defect re-introduced -- the announced configuration is read AFTER the rows' flag was posted,
            rq_qn = g.mbox[Sp::MB_QN + j];          // when the word may already carry the dynamics wave's NEXT request (a mailbox word belongs to the dynamics wave again once the rows are posted)
#endif
            qg = rq_qn; qxg = rq_qnx; post = Sp::MB_PRE;
        } else {
            // command 2: this evaluation's configuration is not the one announced (first request of a launch, reset, injected state): its geometry now
            // (the dynamics wave waits for it: MB_MOK), its rows in the next pass
            qg = rq_q; qxg = rq_qx; post = Sp::MB_MOK;
            owe_commit = true;
        }
        // ---- geometry: kinematics -> body frames + root height (g_fk publishes into this wave's own region) and the lanes' joint axes; the mass matrix; the
        // configuration half of the constraints (registers) and what the dynamics wave needs to know of it.  The detection comes last: its results stay in
        // registers until the next request, across nothing but the wait.
        g_fk<T, TP, true, false>(g, lt, qg, qxg, kin);
        st4(wb + Sp::AXX + 4 * j, kin.axis.x, kin.axis.y, kin.axis.z, kin.rootz);
        g_mass_rows<T, TP>(g, lt, kin);
        g_detect_constraints<T, TP, false>(g, lt, grp, qg, det);
        g.mbox[Sp::MB_LIM + j] = (T)det.my_lim; g.mbox[Sp::MB_SGN + j] = det.lim_sign;
        if (j == 0) { g.mbox[Sp::MB_NCON] = (T)det.ncon; g.mbox[Sp::MB_NLIM] = (T)det.nlim; }
        g_sync<T>();
        DL_WG_RELEASE();
        if (lane == 0) flags[post] = seq;
        DL_WAKE();
#if defined(DL_EXP_SPLIT_PROF) && DL_EXP_SPLIT_PROF != 3          // busy cycles of this wave (request seen -> look-ahead finished), summed over the launch: dbg slot 0 (tools/diag_split.py)
        if (!owe_commit) srv_busy += DL_CLOCK() - tsrv0;
#endif
    }
#ifdef DL_EXP_SPLIT_PROF
    if (j == 0 && w0 < n && st.dbg) st.dbg[w] = (int)(srv_busy >> 4);
#endif
    if (DL_PREFETCH_ACTIONS && actions_all && pf_sink == 12345.678f) g.mbox[Sp::MB_SIZE - 1] = pf_sink;      // (keeps the prefetch loads alive)
}

// `nsteps` control steps (time-major arrays: step s uses actions[s], writes obs[s], rew[s], done[s]).  More than one
// step per launch is for callers whose actions do not depend on the observations (dl_rollout_fixed): the launch then
// lasts as long as the wave with the largest SUM over the steps, not the sum of the per-step maxima.
// TIMED: tim = [10][nwg]: 0-6 g_forward's sections, 7 whole kernel, 8 before the physics, 9 after it.
// SPLIT: the dynamics wave of a split workgroup (its partner runs g_constraint_server on the same walkers); smem = the LDS of the wave pair.
template <typename T, typename TP, bool TIMED = false, bool SPLIT = false>
__device__ __forceinline__ void g_wave_env_step(int lane, int wblock, int wg, int nwg, DL_LDS T* smem, const GModel<T, TP>* __restrict__ gm, const DevCfg<T>& c, const DevState<T>& st,
                                                const float* __restrict__ actions_all, float* obs_all, float* rew_all, uint8_t* done_all, float* term_obs_all, float* rew_terms_all,
                                                const T* inj_q, const T* inj_v, const int32_t* inj_flags, float* ctrl_out, int eval_mode, int nsteps, long long* tim) {
    using D = GD<TP>;
    using Ld = GLds<TP>;
    constexpr int NL = D::NL, NX = D::NX, NV = D::NV, NU = TP::NU, OBS = TP::OBS;
    static_assert(NX == 0 || NX == 3, "the replicated dofs are the three root translations");
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_begin = 0;
    if constexpr (TIMED) t_begin = DL_CLOCK();
#ifdef DL_EXP_SPLIT_PROF
    const long long t_begin0 = DL_CLOCK();
#endif
    const int grp = lane >> 4, j = lane & 15, n = st.n;
    const int w0 = wblock * GW + grp;
    const bool valid = w0 < n;
    const int w1 = valid ? w0 : n - 1;
    const DL_CONST GModel<T, TP>* m = (const DL_CONST GModel<T, TP>*)gm;
    const GLane<T, D::NPASS> ln = m->lanes[j];     // this lane's record of the model block (built on the host by g_load_lane)
    GConst<T, TP> cst;
    g_load_const<T, TP>(*m, cst);
    GWalk<T> wk;
    g_load_walk<T, TP>(m, st, w1, wk);
    constexpr int WSTRIDE = SPLIT ? GSplit<TP>::TOTAL : Ld::TOTAL;         // LDS words per walker
    GCtx<T, TP> g{smem + (size_t)grp * WSTRIDE, m, j, &ln, &cst, &wk,
                  smem + (size_t)grp * WSTRIDE + (SPLIT ? GSplit<TP>::MMX : Ld::MM), SPLIT ? smem + (size_t)grp * WSTRIDE + GSplit<TP>::MB : nullptr, SPLIT ? smem + GSplit<TP>::MB : nullptr,
                  st.fault, st.spin_dyn, smem + (size_t)grp * WSTRIDE + Ld::BFR, smem + (size_t)grp * WSTRIDE + Ld::MISC};
    if constexpr (!SPLIT) g.strict = st.strict_solver;
    int split_seq[4] = {0, 0, 0, 0};         // [0] command counter of the wave pair, [1], [2] cycle counters of the profiling build, [3] the hand-over has failed
    T q_ann = __builtin_bit_cast(T, (std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>)(~0ull));      // split workgroup: the configuration announced to the partner wave with the last request (none yet: a NaN pattern no state carries)
    DL_LDS T* wb = g.wb;
    const bool isdof = j < NL;
    const int jd = j + NX;                         // dof of this lane
    GLaneTopo<T> lt;
    g_lane_topo<T, TP>(j, lt);
    T q = T(0), v = T(0), warm = T(0);
    if (isdof) { const size_t o = (size_t)jd * n + w1; q = st.qpos[o]; v = st.qvel[o]; warm = st.warm[o]; }
    if constexpr (SPLIT) q_ann = q;          // the partner wave starts with the geometry of the state in memory (g_constraint_server): "announced" by the launch itself
    GX<T, NX> qx, vx, warmx;
    static_for<NX>([&](auto ti) { constexpr int t = ti.value; const size_t o = (size_t)t * n + w1; qx.x[t] = st.qpos[o]; vx.x[t] = st.qvel[o]; warmx.x[t] = st.warm[o]; });
    GX<T, NX> qx_ann = qx;                   // (the replicated dofs of the announced configuration)
    int32_t cur[DL_CUR_WORDS];
#pragma unroll
    for (int k = 0; k < DL_CUR_WORDS; k++) cur[k] = st.cur[(size_t)k * n + w1];
    // per-walker words of the environment logic live in registers across the control steps of this launch
    // the walked distance (a float64 accumulator, touched once per control step) sits in two free words of the walker's LDS region, not in registers across the physics
    DL_LDS T* const walked_w = smem + (size_t)grp * WSTRIDE + Ld::MISC + 2;
    auto walked_put = [&](double x) { if (j == 0) { if constexpr (sizeof(T) == 8) walked_w[0] = x; else { const uint64_t b = __builtin_bit_cast(uint64_t, x); uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32); DL_VPIN(lo); DL_VPIN(hi); walked_w[0] = __builtin_bit_cast(T, lo); walked_w[1] = __builtin_bit_cast(T, hi); } } };      // (pinned: a constant 0.0 is otherwise a register pair hoisted out of the step loop)
    auto walked_get = [&]() -> double { if constexpr (sizeof(T) == 8) return walked_w[0]; else { const T w0 = walked_w[0], w1 = walked_w[1]; return __builtin_bit_cast(double, (uint64_t)__builtin_bit_cast(uint32_t, w0) | ((uint64_t)__builtin_bit_cast(uint32_t, w1) << 32)); } };
    walked_put(st.walked[w1]);
    // COM-z offset of the reference step the cursor reads: quirk Q4 (default) keeps one per step and walker in memory (st.zacc: the data set the reference mutates in place),
    // a register holds the current step's; DL_INTENDED_COMZ_PER_EPISODE: the last reset's, valid on the reset step only
    const bool q4 = q4_on(c);
    T comz = q4 ? zacc_load(st.zacc + (size_t)cur[DL_CUR_READ_STEP] * n + w1) : st.comz_off[w1];
    // (the reward terms of the last step are values of type T held as doubles in the Monitor words: kept as T here, widened where they are used)
    T terms[3] = {(T)st.mon[(size_t)MON_POSREW * n + w1], (T)st.mon[(size_t)MON_VELREW * n + w1], (T)st.mon[(size_t)MON_COMREW * n + w1]};
    long long t_phys_end = 0;
    // push schedule on the device (BASELINE config 5 without a host round trip per control step)
    const V3<T> push_force = wk.push;
    const int push_phase = (st.push_phase && st.rnd) ? st.push_phase[w1] : -1;
#pragma unroll 1
    for (int step = 0; step < nsteps; step++) {
    // the walker index is opaque per control step: address arithmetic on it (some sixty per-walker words of state, Monitor and output
    // arrays) is then done where it is used instead of being hoisted out of the loop as sixty 64-bit pointers (100 registers)
    int w = w1;
    DL_VPIN(w);
    if (push_phase >= 0) {
        const bool on = ((st.push_step0 + step + push_phase) % st.push_period) < st.push_dur;
        wk.push = on ? push_force : mk<T>(0, 0, 0);
        wk.pushed = on && (push_force.x != T(0) || push_force.y != T(0) || push_force.z != T(0));
    }
    const float* aa = actions_all;
    DL_SPIN(aa);
    const float* __restrict__ actions = aa + (size_t)step * n * NU;
    // (the output rows of this step are addressed AFTER the physics, from a step count made opaque there: formed here they are five 64-bit values that live,
    //  spilled to scratch, through the twenty forward evaluations)
    // ---- _rescale_actions + mirror_action (cursor BEFORE refs.next()), per actuated dof
    bool mirr_a = false;
    if constexpr (TP::ENV_KIND == 0) mirr_a = c.mirror_policy && c.step_is_left[cur[DL_CUR_I_STEP]];
    T ctrl = T(0), force = T(0);
    const int a = isdof ? ln.act : -1;
    if (a >= 0) {
        int a_o = a;
        DL_VPIN(a_o);             // (opaque: the addresses formed from it -- the permutation table's entry, the test hook's slot -- are per-lane pointers that would otherwise be hoisted out of the step loop and spilled)
        const int src = mirr_a ? TP::act_perm(a_o) : a;
        const int jsrc = TP::act_dof(src) - NX;
        const T x = dl_clamp((T)actions[(size_t)w * NU + src], T(-1), T(1));
        const T raw = x > T(0) ? x * m->ctrl_hi[jsrc] : dl_abs(x) * m->ctrl_lo[jsrc];
        ctrl = (mirr_a && TP::act_neg(a)) ? -raw : raw;
        const T u = dl_clamp(ctrl, ln.ctrl_lo, ln.ctrl_hi);
        force = ln.gear * dl_clamp(u, ln.force_lo, ln.force_hi);
        if (ctrl_out && valid) ctrl_out[(size_t)step * n * NU + (size_t)w * NU + a_o] = (float)ctrl;      // test hook: sim.data.ctrl as the reference sets it
    }
    const T tor_sum = gsum(a >= 0 ? dl_abs(dl_clamp(ctrl, ln.force_lo, ln.force_hi)) : T(0));
    if constexpr (SPLIT) {
        if constexpr (GSplit<TP>::LANE_FILE) {          // the lane file of this control step (g_smooth_dynamics reads it once per evaluation)
            st4(wb + GSplit<TP>::LSP + 4 * j, force, ln.damping, T(0), T(0));
            if (j == 0) st4(wb + Ld::MISC + 4, cst.xs_damping[0], cst.xs_damping[NX > 1 ? 1 : 0], cst.xs_damping[NX > 2 ? 2 : 0], T(0));
        }
    }
    // ---- what the END of the step will look up does not depend on the physics: refs.next() is a function of the cursor alone.  The cursor
    // is advanced on a copy here and the reference sample, the step length and the desired velocity behind it are requested BEFORE the
    // physics (chains of dependent table reads: a lone wave per SIMD would wait out three or four memory round trips after the physics);
    // a diverged step (exception path) discards them.  Lane-only walkers.  EXPERIMENT (DL_PREFETCH_REFS, off by default: measured no gain).
    constexpr bool PRE = (NX == 0) && (TP::ENV_KIND == 0) && DL_PREFETCH_REFS;
    int32_t cur_n[DL_CUR_WORDS];
    T pre_qr = T(0), pre_vr = T(0), pre_dist = T(0), pre_desvel = T(0);
    int pre_len = 1, pre_left = 0;
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < DL_CUR_WORDS; k++) cur_n[k] = cur[k];
        cursor_next<T, TP>(c, cur_n);
        const int rs = cur_n[DL_CUR_READ_STEP], off0 = c.step_off[rs];
        const int base_n = off0 + cur_n[DL_CUR_POS];
        pre_len = c.step_off[rs + 1] - off0;
        if (isdof) { pre_qr = ref_at(c, jd, base_n); pre_vr = ref_at(c, NV + jd, base_n); }
        if (cur_n[DL_CUR_HAS_DIST]) pre_dist = ref_at(c, 0, c.step_off[cur_n[DL_CUR_RSI_STEP] + 1] - 1);
        const int iv = cur_n[DL_CUR_I_STEP] - cur_n[DL_CUR_COUNT] + 1;
        pre_desvel = c.step_vel[iv > 0 ? iv : 0];
        pre_left = c.step_is_left[cur_n[DL_CUR_I_STEP]];
    }
    if constexpr (TIMED) tacc[8] = DL_CLOCK() - t_begin;
    // ---- physics
    bool exc = false;
    const int flag = inj_flags ? inj_flags[w] : 0;
    if (flag == 2) exc = true;
    else if (flag == 1) {
        int jd_o = jd;
        DL_VPIN(jd_o);            // (opaque: `inj_q + jd * n` is otherwise a per-lane pointer hoisted out of the step loop and kept, spilled, through the launch -- for a test hook)
        if (isdof) { q = inj_q[(size_t)jd_o * n + w]; v = inj_v[(size_t)jd_o * n + w]; }
        static_for<NX>([&](auto ti) { constexpr int t = ti.value; qx.x[t] = inj_q[(size_t)t * n + w]; vx.x[t] = inj_v[(size_t)t * n + w]; });
    }
    const bool simulate = flag == 0;
    int dbg_it = 0, dbg_max = 0, dbg_rows = 0;
    if (__any(simulate)) {
        const T h = m->timestep;
        const int fs = m->frame_skip;
        T acc_s2_prev = T(0);
#if DL_EXP_EXTRAP
        T acc_s0_prev = T(0), acc_s2_pp = T(0);          // experiment: quadratic extrapolation of the solver's start (lane dofs only)
#endif
        GX<T, NX> accx_s2_prev;
        static_for<NX>([&](auto ti) { accx_s2_prev.x[ti.value] = T(0); });
#pragma unroll 1
        for (int kf = 0; kf < fs; kf++) {
            // mj_checkPos / mj_checkVel
            {
                bool bad = isdof && (dl_bad(q) || dl_bad(v));
                static_for<NX>([&](auto ti) { bad = bad || dl_bad(qx.x[ti.value]) || dl_bad(vx.x[ti.value]); });
                if (simulate && !exc && gany(bad)) exc = true;
            }
            const T q0 = q, v0 = v;
            T dq = T(0), dv = T(0), qs = q, vs = v, q_end = q;
            T acc_s0 = warm;
            GX<T, NX> qx0 = qx, vx0 = vx, dqx, dvx, qsx = qx, vsx = vx, accx_s0 = warmx, qx_end = qx;
            static_for<NX>([&](auto ti) { dqx.x[ti.value] = T(0); dvx.x[ti.value] = T(0); });
#pragma unroll 1
            for (int stage = 0; stage < 4; stage++) {
                int nc, ne, ni;
                // starting point of the Newton iteration (the minimiser does not depend on it): the last solution (MuJoCo's
                // qacc_warmstart), linearly extrapolated where the next stage lies half a time step further: stage 1 from
                // (stage 2 of the previous mj_step, stage 0), stage 3 from (stage 0, stage 2)
                T start = warm;
                GX<T, NX> startx = warmx, accx;
#if DL_EXP_EXTRAP
                if (NX == 0 && stage == 1 && kf > 0) { const T lin = warm + (warm - acc_s2_prev), quad = T(3) * warm - T(3) * acc_s2_prev + acc_s0_prev; start = lin + T(DL_EXP_EXTRAP_W) * (quad - lin); }
                else if (NX == 0 && stage == 3 && kf > 0) { const T lin = warm + (warm - acc_s0), quad = T(3) * warm - T(3) * acc_s0 + acc_s2_pp; start = lin + T(DL_EXP_EXTRAP_W) * (quad - lin); }
                else
#endif
                if (stage == 1 && kf > 0) { start = warm + T(DL_EXTRAP_BETA) * (warm - acc_s2_prev); static_for<NX>([&](auto ti) { constexpr int t = ti.value; startx.x[t] = warmx.x[t] + (warmx.x[t] - accx_s2_prev.x[t]); }); }
                else if (stage == 3) { start = warm + T(DL_EXTRAP_BETA) * (warm - acc_s0); static_for<NX>([&](auto ti) { constexpr int t = ti.value; startx.x[t] = warmx.x[t] + (warmx.x[t] - accx_s0.x[t]); }); }
                // RK4: the configuration of the NEXT evaluation depends on this stage's velocity only -- known before this stage's solve.  A split
                // workgroup hands it to the partner wave, which computes that configuration's mass matrix while this wave solves (g_forward<SPLIT>).
                const T wgt = (stage == 0 || stage == 3) ? T(1) / T(6) : T(1) / T(3);
                const T al = stage == 2 ? T(1) : T(0.5);
                const T dq_new = dq + wgt * vs;
                const T q_ahead = stage == 3 ? q0 + h * dq_new : q0 + h * al * vs;          // stage 3: the state after this mj_step = stage 0 of the next
                GX<T, NX> dqx_new, qx_ahead;          // the same for the replicated dofs: formed ONCE, used for the announcement and as the next stage's configuration (the same bits)
                static_for<NX>([&](auto ti) {
                    constexpr int t = ti.value;
                    dqx_new.x[t] = dqx.x[t] + wgt * vsx.x[t];
                    qx_ahead.x[t] = stage == 3 ? qx0.x[t] + h * dqx_new.x[t] : qx0.x[t] + h * al * vsx.x[t];
                });
                const T acc = g_forward<T, TP, TIMED, SPLIT>(g, lt, grp, qs, vs, force, start, qsx, vsx, startx, accx, nc, ne, ni, tacc, split_seq, q_ahead, &q_ann, &qx_ahead, &qx_ann);
                if constexpr (SPLIT) { if (split_seq[3] && simulate) exc = true; }      // the hand-over with the constraint wave failed: MujocoException path
#if DL_EXP_EXTRAP
                if (stage == 2) acc_s2_pp = acc_s2_prev;          // (before the overwrite below: the previous mj_step's stage 2)
#endif
                if (stage == 0) { acc_s0 = acc; accx_s0 = accx; }
                if (stage == 2) { acc_s2_prev = acc; accx_s2_prev = accx; }
                dbg_it += ni; dbg_max = ni > dbg_max ? ni : dbg_max; dbg_rows += ne;
                if (st.dbgf && valid) {
                    if (ni >= st.dbg_cap) { st.dbgf[(size_t)j * n + w] = (float)qs; st.dbgf[(size_t)(16 + j) * n + w] = (float)vs; st.dbgf[(size_t)(32 + j) * n + w] = (float)start; }
                    // Newton iterations and constraint rows of this walker in evaluation (kf, stage) of the launch's last control step: iterations + 128 rows
                    if (j == 0 && 4 * kf + stage < DL_DBG_EVALS) st.dbgf[(size_t)(48 + 4 * kf + stage) * n + w] = (float)(ni + 128 * ne);
                }
                if (simulate && !exc) { warm = acc; warmx = accx; }
                if (stage == 0 && simulate && !exc) {     // mj_checkAcc
                    bool bad = isdof && dl_bad(acc);
                    static_for<NX>([&](auto ti) { bad = bad || dl_bad(accx.x[ti.value]); });
                    if (gany(bad)) exc = true;
                }
                dq = dq_new; dv += wgt * acc;
                q_end = q_ahead;
                qs = q_ahead; vs = v0 + h * al * acc;
                static_for<NX>([&](auto ti) {
                    constexpr int t = ti.value;
                    dqx.x[t] = dqx_new.x[t]; dvx.x[t] += wgt * accx.x[t];
                    qsx.x[t] = qx_ahead.x[t]; vsx.x[t] = vx0.x[t] + h * al * accx.x[t];
                });
                qx_end = qx_ahead;
            }
#if DL_EXP_EXTRAP
            acc_s0_prev = acc_s0;
#endif
            if (simulate && !exc) {
                q = q_end; v = v0 + h * dv;          // q_end = q0 + h * dq, formed before the last stage's solve
                static_for<NX>([&](auto ti) { constexpr int t = ti.value; qx.x[t] = qx_end.x[t]; vx.x[t] = vx0.x[t] + h * dvx.x[t]; });          // qx_end = qx0 + h * dqx, formed before the last stage's solve
            }
        }
    }
    if constexpr (TIMED) t_phys_end = DL_CLOCK();
    // ---- environment logic
    const double tor_mean = (double)tor_sum / NU;
    float r;
    bool dn;
    // stage q, v in LDS for the observation writer (all NV dofs in dof order)
    auto stage_qv = [&]() {
        if (isdof) { wb[Ld::Q + jd] = q; wb[Ld::V + jd] = v; }
        if (j == 0) static_for<NX>([&](auto ti) { constexpr int t = ti.value; wb[Ld::Q + t] = qx.x[t]; wb[Ld::V + t] = vx.x[t]; });
    };
    // observation from q, v staged in LDS: OBS outputs over 16 lanes
    auto write_obs = [&](float* dst_base, bool use_pre) {
        if (!(valid && dst_base)) return;
        if constexpr (TP::ENV_KIND == 0) {
            // mimic_env.py:403-437 + mirror_obs :440-480
            T phase_var, desvel;
            bool mirr_o;
            if (PRE && use_pre) {           // looked up before the physics (same cursor, same table entries)
                phase_var = T(cur[DL_CUR_POS]) / T(pre_len); desvel = pre_desvel; mirr_o = c.mirror_policy && pre_left;
            } else {
                const int rs = cur[DL_CUR_READ_STEP];
                phase_var = T(cur[DL_CUR_POS]) / T(c.step_off[rs + 1] - c.step_off[rs]);
                const int iv = cur[DL_CUR_I_STEP] - cur[DL_CUR_COUNT] + 1;
                desvel = c.step_vel[iv > 0 ? iv : 0];
                mirr_o = c.mirror_policy && c.step_is_left[cur[DL_CUR_I_STEP]];
            }
            auto raw_obs = [&](int k) -> T { return k == 0 ? phase_var : (k == 1 ? desvel : (k < 1 + NV ? wb[Ld::Q + (k - 1)] : wb[Ld::V + (k - 1 - NV)])); };
            for (int k = j; k < OBS; k += GL) {
                const T plain = raw_obs(k);
                const T mir = TP::obs_neg(k) ? -raw_obs(TP::obs_perm(k)) : raw_obs(TP::obs_perm(k));
                dst_base[(size_t)w * OBS + k] = dl_sat_out((float)(mirr_o ? mir : plain));
            }
        } else {
            // MimicWalker165cm65kg: 4 x (phase angle, phase radius) from joint phase plots (mimic_env.py:330-401), 2 desired
            // velocities = mean reference pelvis x / z velocity over the next 0.5 s (loco3d_trajecs.py:51-97, through float64
            // prefix sums), qpos[1:], qvel
            const int L = c.step_off[1] - c.step_off[0], pos = cur[DL_CUR_POS];
            const int end = pos + 250 < L - 1 ? pos + 250 : L - 1;
            const double cnt = (double)(end - pos);
            for (int k = j; k < OBS; k += GL) {
                float o;
                if (k < 8) {
                    const int d = TP::phase_joint(k >> 1);
                    const T qj = wb[Ld::Q + d], vj = wb[Ld::V + d];
                    o = (k & 1) ? (float)(dl_sqrt(qj * qj + vj * vj) / T(5)) : (float)(dl_atan2(vj, -qj) * T(0.31830988618379067154));
                }
                else if (k == 8) o = (float)((c.pref[end] - c.pref[pos]) / cnt);
                else if (k == 9) o = (float)((c.pref[(size_t)c.total_len + 1 + end] - c.pref[(size_t)c.total_len + 1 + pos]) / cnt);
                else if (k < 9 + NV) o = (float)wb[Ld::Q + (k - 9)];
                else o = (float)wb[Ld::V + (k - 9 - NV)];
                dst_base[(size_t)w * OBS + k] = dl_sat_out(o);
            }
        }
    };
    int step_o = step;
    DL_SPIN(step_o);
    float* obs = obs_all + (size_t)step_o * n * OBS;
    float* rew = rew_all + (size_t)step_o * n;
    uint8_t* done = done_all + (size_t)step_o * n;
    float* term_obs = term_obs_all ? term_obs_all + (size_t)step_o * n * OBS : nullptr;
    float* rew_terms = rew_terms_all ? rew_terms_all + (size_t)step_o * n * 3 : nullptr;
    if (exc) {
        r = 0.0f; dn = true; walked_put(0.0);
        terms[0] = terms[1] = terms[2] = T(1);
    } else {
        const int rs_old = cur[DL_CUR_READ_STEP];
        if constexpr (PRE) {
#pragma unroll
            for (int k = 0; k < DL_CUR_WORDS; k++) cur[k] = cur_n[k];
        } else cursor_next<T, TP>(c, cur);
        if (q4 && cur[DL_CUR_READ_STEP] != rs_old) comz = zacc_load(st.zacc + (size_t)cur[DL_CUR_READ_STEP] * n + w);          // rolled into another step: the offset ITS row carries (quirk Q4)
        g_sync<T>();
        stage_qv();
        g_sync<T>();
        cur[DL_CUR_EP_DUR] += 1;
        const T vx0 = dl_clamp(g_dof_value<TP, 0>(v, vx), T(-5.5), T(5.5)), vy0 = dl_clamp(g_dof_value<TP, 1>(v, vx), T(-5.5), T(5.5));
        T icf = c.inv_ctrl_freq;
        DL_VPIN(icf);             // (the widened constant is otherwise hoisted out of the step loop)
        walked_put(walked_get() + (double)dl_sqrt(vx0 * vx0 + vy0 * vy0) * (double)icf);
        const bool timeout = cur[DL_CUR_EP_DUR] >= c.ep_dur_max;
        const T qz = g_dof_value<TP, 2>(q, qx);
        dn = (qz < c.com_z_min) || timeout;
        if (dn) r = timeout ? 0.0f : -0.0f;
        else {
            // imitation reward: every dof lane contributes its squared differences (dofs 0..2 = COM term)
            const int base = c.step_off[cur[DL_CUR_READ_STEP]] + cur[DL_CUR_POS];
            auto ref_q = [&](int d) -> T {
                T qr = ref_at(c, d, base);
                if (cur[DL_CUR_HAS_DIST] && d == 0) qr += ref_at(c, 0, c.step_off[cur[DL_CUR_RSI_STEP] + 1] - 1);
                if (d == 2 && (q4 || !cur[DL_CUR_HAS_DIST])) qr -= comz;
                return qr;
            };
            T dp = T(0), dvv = T(0), dc = T(0);
            if (isdof) {
                T d1, d2;
                if constexpr (PRE) {
                    T qr = pre_qr;
                    if (cur[DL_CUR_HAS_DIST] && jd == 0) qr += pre_dist;
                    if (jd == 2 && (q4 || !cur[DL_CUR_HAS_DIST])) qr -= comz;
                    d1 = q - qr; d2 = v - pre_vr;
                } else { d1 = q - ref_q(jd); d2 = v - ref_at(c, NV + jd, base); }
                if (jd < 3) dc = d1 * d1; else { dp = d1 * d1; dvv = d2 * d2; }
            }
            T s3[3] = {dp, dvv, dc};
            gsum_n<3>(s3);
            static_for<NX>([&](auto ti) { constexpr int t = ti.value; const T d1 = qx.x[t] - ref_q(t); s3[2] += d1 * d1; });
            const T tp = dl_exp(T(-3) * s3[0]), tv = dl_exp(T(-0.05) * s3[1]), tc = dl_exp(T(-16) * s3[2]);
            terms[0] = tp; terms[1] = tv; terms[2] = tc;
            r = (float)((c.rew_w[0] * tp + c.rew_w[1] * tv + c.rew_w[2] * tc) * c.rew_scale + c.alive_bonus);
        }
        write_obs(dn ? term_obs : obs, true);
    }
    if (valid && j == 0 && st.dbg) {
#ifndef DL_EXP_SPLIT_PROF
        st.dbg[w] += dbg_it;
#endif
        st.dbg[(size_t)n + w] = dbg_max; st.dbg[(size_t)2 * n + w] += dbg_rows; st.dbg[(size_t)3 * n + w] += exc ? 1 : 0;
#ifdef DL_EXP_SPLIT_PROF
        if constexpr (SPLIT) { st.dbg[(size_t)n + w] = split_seq[1]; st.dbg[(size_t)2 * n + w] = split_seq[2]; st.dbg[(size_t)3 * n + w] = (int)((DL_CLOCK() - t_begin0) >> 4); }
#endif
    }
    if (valid && j == 0) {
        const double terms_d[3] = {(double)terms[0], (double)terms[1], (double)terms[2]};
        monitor_step(st.mon, n, w, (double)r, dn, terms_d, tor_mean, walked_get(), cur[DL_CUR_POS], exc);
        if (rew_terms) { rew_terms[3 * (size_t)w] = (float)terms[0]; rew_terms[3 * (size_t)w + 1] = (float)terms[1]; rew_terms[3 * (size_t)w + 2] = (float)terms[2]; }
        rew[w] = r == r ? r : 0.0f;          // (a NaN can only come out of a state beyond float32's range: see dl_sat_out)
        done[w] = dn ? 1 : 0;
    }
    // ---- vec-env auto reset inside the same launch (SubprocVecEnv worker: obs = env.reset() after done; a
    // diverged step resets twice, the first reset's observation being the terminal observation):
    // MujocoEnv.reset -> reset_model (mimic_env.py:526-572).  The warm start of the new episode is zero: the
    // solver's minimiser does not depend on it.
    const int nrep = exc ? 2 : (dn ? 1 : 0);
    if (nrep > 0) {
#pragma unroll 1
        for (int rep = 0; rep < nrep; rep++) {
            int s0, p0, read = -1;
            if (eval_mode && TP::ENV_KIND == 1) { s0 = 0; p0 = 0; }       // base get_deterministic_init_state(0 %)
            else if (eval_mode) {
                s0 = cur[DL_CUR_EVAL_K];
                p0 = (int)(0.75 * (double)(c.step_off[s0 + 1] - c.step_off[s0]));
                read = (c.intended & DL_INTENDED_EVAL_OWN_STEP) ? s0 : 0;
                cur[DL_CUR_EVAL_K] = (s0 + 1 >= 20) ? 0 : s0 + 1;
            }
            else if (st.inj_rsi && st.inj_rsi[w] >= 0) { s0 = st.inj_rsi[w]; p0 = st.inj_rsi[(size_t)n + w]; }
            else rsi_draw(c, (uint32_t)(c.env_index_base + w), (uint32_t)cur[DL_CUR_EPISODE], s0, p0);
            cur[DL_CUR_EPISODE] += 1;
            cur[DL_CUR_EP_DUR] = 0;
            if (c.intended & DL_INTENDED_COUNT_PER_EPISODE) cur[DL_CUR_COUNT] = 1;
            cur[DL_CUR_I_STEP] = s0; cur[DL_CUR_RSI_STEP] = s0; cur[DL_CUR_READ_STEP] = read >= 0 ? read : s0; cur[DL_CUR_POS] = p0; cur[DL_CUR_HAS_DIST] = 0;
            const int base = c.step_off[cur[DL_CUR_READ_STEP]] + p0;
            if (isdof) { q = ref_at(c, jd, base); v = ref_at(c, NV + jd, base); }
            static_for<NX>([&](auto ti) { constexpr int t = ti.value; qx.x[t] = ref_at(c, t, base); vx.x[t] = ref_at(c, NV + t, base); });
            { GKin<T> kin; g_fk<T, TP>(g, lt, q, qx, kin); }
            comz = g_lowest_site<T, TP>(g);
            if constexpr (NX > 0) qx.x[2] -= comz; else { if (j == 2) q -= comz; }
            // quirk Q4 (adjust_COM_Z_pos): the row of the step just bound is re-anchored in this walker's data set (see env_reset_lane: replaced, not summed)
            if (q4 && valid && j == 0) zacc_store(st.zacc + (size_t)cur[DL_CUR_READ_STEP] * n + w, comz);
            g_sync<T>();
            stage_qv();
            warm = T(0);
            static_for<NX>([&](auto ti) { warmx.x[ti.value] = T(0); });
            const int rs_reset = cur[DL_CUR_READ_STEP];
            cursor_next<T, TP>(c, cur);
            if (q4 && cur[DL_CUR_READ_STEP] != rs_reset) comz = zacc_load(st.zacc + (size_t)cur[DL_CUR_READ_STEP] * n + w);          // (a reset onto the last samples of a step: the first refs.next() rolls over)
            g_sync<T>();
            write_obs((nrep == 2 && rep == 0) ? term_obs : obs, false);
            g_sync<T>();
        }
        walked_put(0.0);
        terms[0] = terms[1] = terms[2] = T(1);       // reset_model's sanity check evaluates the reward terms at the init state (:562)
    }
    }   // control steps of this launch
    if constexpr (SPLIT) {      // release the constraint wave
        if (lane == 0) { ((volatile DL_LDS int*)g.mbox0)[GSplit<TP>::MB_CMD] = 0; ((volatile DL_LDS int*)g.mbox0)[GSplit<TP>::MB_CMDSEQ] = split_seq[0] + 1; }
        DL_WAKE();
    }
    // (the walker index is opaque here as well: the addresses of the final stores are formed now -- kept from the loads at the top they were twenty 64-bit
    //  values per lane that lived, spilled to scratch, through the whole launch)
    int we = w1;
    DL_VPIN(we);
    if (valid && j == 0) {
        st.comz_off[we] = comz;
        st.mon[(size_t)MON_POSREW * n + we] = (double)terms[0]; st.mon[(size_t)MON_VELREW * n + we] = (double)terms[1]; st.mon[(size_t)MON_COMREW * n + we] = (double)terms[2];
        st.walked[we] = walked_get();
#pragma unroll
        for (int k = 0; k < DL_CUR_WORDS; k++) st.cur[(size_t)k * n + we] = cur[k];
        static_for<NX>([&](auto ti) { constexpr int t = ti.value; const size_t o = (size_t)t * n + we; st.qpos[o] = qx.x[t]; st.qvel[o] = vx.x[t]; st.warm[o] = warmx.x[t]; });
    }
    if (valid && isdof) { const size_t o = (size_t)jd * n + we; st.qpos[o] = q; st.qvel[o] = v; st.warm[o] = warm; }
    if constexpr (TIMED) {
        const long long t_end = DL_CLOCK();
        tacc[7] = t_end - t_begin; tacc[9] = t_end - t_phys_end;
        if (lane == 0) for (int k = 0; k < 10; k++) tim[(size_t)k * nwg + wg] = tacc[k];
    }
}

}  // namespace dl
