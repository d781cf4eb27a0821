// emu.cpp -- TEST INFRASTRUCTURE: compiles the kernels' per-lane source (drloco_amd/csrc/dl_core.hpp,
// dl_env.hpp) for the host and drives it one lane at a time, so that the device code's logic can be
// checked against the independent oracle on a machine without a GPU.  Never linked into the product.
// The 16-lanes-per-walker kernels (dl_group.hpp, dl_group_env.hpp) run here too: DL_GROUP_EMU swaps the wave-level
// builtins for dl_group_emu.hpp's fibers (one wave = four walkers = 64 cooperative lanes).
#define DL_GROUP_EMU 1
#include <cstdlib>
#include <vector>

#include "../../drloco_amd/csrc/dl_host.hpp"

using namespace dl;

template <typename T, typename TP> struct Emu {
    DevModel<T, TP> m;
    DevCfg<T> c;
    DevState<T> st;
    std::vector<T> qpos, qvel, warm, comz, zacc, table, step_vel, lane, work;
    std::vector<int32_t> cur, need, inj, step_off, is_left, inj_flags;
    std::vector<double> walked, mon;
    std::vector<T> inj_q, inj_v;
    int n;
    int eval_mode = 0;
    std::vector<double> pref;
    GModel<T, TP> gm;                 // table-driven model of the 16-lane kernels
    bool gm_ok = false;
    std::vector<T> rnd, smem;
    std::vector<int32_t> push_phase;
    int push_step = 0;
    int32_t fault_word = 0;
};

template <typename T, typename TP> static Emu<T, TP>* emu_create(const dl_model_desc* d, const dl_refs_desc* r, const dl_config* cfg, int n) {
    std::string why;
    if (!check_topology<TP>(*d, why)) { fprintf(stderr, "%s\n", why.c_str()); return nullptr; }
    auto* e = new Emu<T, TP>();
    e->n = n;
    fill_dev_model<T, TP>(*d, e->m);
    fill_dev_cfg<T>(*cfg, *r, e->c);
    size_t tn = (size_t)r->n_rows * r->total_len;
    e->table.resize(tn);
    {
        std::vector<double> tr;
        transpose_refs(*r, tr);
        for (size_t k = 0; k < tn; k++) e->table[k] = (T)tr[k];
    }
    e->step_off.assign(r->step_off, r->step_off + r->n_steps + 1);
    e->is_left.assign(r->step_is_left, r->step_is_left + r->n_steps);
    e->step_vel.resize(r->n_steps);
    for (int k = 0; k < r->n_steps; k++) e->step_vel[k] = (T)r->step_vel[k];
    if (TP::ENV_KIND == 1) { loco3d_prefix_sums(*r, TP::NV, e->pref); e->c.pref = e->pref.data(); }
    e->c.table = e->table.data(); e->c.step_off = e->step_off.data(); e->c.step_is_left = e->is_left.data(); e->c.step_vel = e->step_vel.data();
    e->qpos.assign((size_t)TP::NV * n, 0); e->qvel.assign((size_t)TP::NV * n, 0); e->warm.assign((size_t)TP::NV * n, 0);
    for (int j = 0; j < TP::NV; j++) for (int i = 0; i < n; i++) e->qpos[(size_t)j * n + i] = (T)d->jnt_qpos0[j];
    e->comz.assign(n, 0); e->cur.assign((size_t)DL_CUR_WORDS * n, 0); e->need.assign(n, 0); e->inj.assign((size_t)2 * n, -1);
    for (int i = 0; i < n; i++) e->cur[(size_t)DL_CUR_COUNT * n + i] = 1;
    e->walked.assign(n, 0); e->mon.assign((size_t)MON_WORDS * n, 0);
    e->inj_flags.assign(n, 0); e->inj_q.assign((size_t)TP::NV * n, 0); e->inj_v.assign((size_t)TP::NV * n, 0);
    e->lane.assign(MemLayout<TP>::TOTAL, 0);
    e->work.assign((size_t)4 * TP::NV * n, 0);
    e->st = DevState<T>{e->qpos.data(), e->qvel.data(), e->warm.data(), e->comz.data(), e->cur.data(), e->walked.data(), e->mon.data(), e->need.data(), e->inj.data(), e->work.data(), n};
    e->zacc.assign((size_t)r->n_steps * n, 0); e->st.zacc = e->zacc.data();          // quirk Q4's record
    e->st.strict_solver = cfg->strict_solver ? 1 : 0;
    e->st.rnd = nullptr; e->st.dbgf = nullptr; e->st.dbg = nullptr; e->st.dbg_cap = 1 << 30;
    e->st.push_phase = nullptr; e->st.push_period = 1; e->st.push_dur = 0; e->st.push_step0 = 0;
    e->gm_ok = fill_group_model<T, TP>(*d, e->gm, why);
    e->smem.assign((size_t)GW * GLds<TP>::TOTAL, 0);
    return e;
}

template <typename T, typename TP> static void emu_reset(Emu<T, TP>* e, const uint8_t* mask, const int32_t* is, const int32_t* ip, float* obs) {
    LaneMem<T> mem{e->lane.data(), 1};
    for (int i = 0; i < e->n; i++) {
        if (mask && !mask[i]) continue;
        env_reset_lane<T, TP>(e->m, e->c, mem, e->st, i, 1, is, ip, obs, nullptr, e->eval_mode);
    }
}
template <typename T, typename TP> static void emu_step(Emu<T, TP>* e, const float* act, float* obs, float* rew, uint8_t* done, float* term, float* terms) {
    LaneMem<T> mem{e->lane.data(), 1};
    for (int i = 0; i < e->n; i++)
        env_step_lane<T, TP>(e->m, e->c, mem, e->st, i, act, obs, rew, done, term, terms, e->inj_q.data(), e->inj_v.data(), e->inj_flags.data());
    for (int i = 0; i < e->n; i++) {
        e->inj_flags[i] = 0;
        if (e->need[i]) env_reset_lane<T, TP>(e->m, e->c, mem, e->st, i, e->need[i], nullptr, nullptr, obs, term, e->eval_mode);
    }
}
template <typename T, typename TP> static void emu_forward(Emu<T, TP>* e, const T* ctrl, T* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter) {
    LaneMem<T> mem{e->lane.data(), 1};
    int n = e->n;
    for (int i = 0; i < n; i++) {
        T q[TP::NV], v[TP::NV], w[TP::NV], u[TP::NU], a[TP::NV];
        for (int j = 0; j < TP::NV; j++) { q[j] = e->qpos[(size_t)j * n + i]; v[j] = e->qvel[(size_t)j * n + i]; w[j] = e->warm[(size_t)j * n + i]; }
        for (int k = 0; k < TP::NU; k++) u[k] = ctrl ? ctrl[(size_t)k * n + i] : T(0);
        const int info = forward_io<T, TP>(e->m, mem, q, v, u, w, a);
        for (int j = 0; j < TP::NV; j++) qacc[(size_t)j * n + i] = a[j];
        if (ncon) ncon[i] = info & 255;
        if (nefc) nefc[i] = (info >> 8) & 255;
        if (niter) niter[i] = info >> 16;
    }
}

// ---- the 16-lane kernels: every wave (four walkers) is run as 64 fibers, waves one after the other
template <typename T, typename TP> static int emu_gforward(Emu<T, TP>* e, const T* ctrl, T* qacc, int32_t* ncon, int32_t* nefc, int32_t* niter) {
    if (!e->gm_ok) return -1;
    const int nwg = (e->n + GW - 1) / GW;
    for (int wg = 0; wg < nwg; wg++)
        dlemu::run_wave(64, [&](int lane) { g_wave_forward<T, TP, false>(lane, g_block_of_workgroup(wg, nwg), wg, nwg, e->smem.data(), &e->gm, e->st, ctrl, qacc, ncon, nefc, niter, nullptr); });
    return 0;
}
template <typename T, typename TP> static int emu_gstep(Emu<T, TP>* e, int nsteps, const float* act, float* obs, float* rew, uint8_t* done, float* term, float* terms, float* ctrl_out) {
    if (!e->gm_ok) return -1;
    const int nwg = (e->n + GW - 1) / GW;
    bool armed = false;
    for (int i = 0; i < e->n; i++) armed = armed || e->inj_flags[i] != 0;
    e->st.push_step0 = e->push_step; e->push_step += nsteps;
    for (int wg = 0; wg < nwg; wg++)
        dlemu::run_wave(64, [&](int lane) {
            g_wave_env_step<T, TP, false>(lane, g_block_of_workgroup(wg, nwg), wg, nwg, e->smem.data(), &e->gm, e->c, e->st, act, obs, rew, done, term, terms,
                                          e->inj_q.data(), e->inj_v.data(), armed ? e->inj_flags.data() : nullptr, ctrl_out, e->eval_mode, nsteps, nullptr);
        });
    for (int i = 0; i < e->n; i++) e->inj_flags[i] = 0;
    return 0;
}
// ---- the split workgroup's wave PAIR on the host: the dynamics wave (g_wave_env_step<SPLIT>) and its partner (g_constraint_server) as two emulated waves that share the
// pair's LDS block, their rounds interleaved by dlemu::run_pair's schedule (`policy`, `seed`).  One pair (four walkers) after the other.
template <typename T, typename TP> static int emu_gstep_split(Emu<T, TP>* e, int nsteps, const float* act, float* obs, float* rew, uint8_t* done, int policy, uint64_t seed) {
    if (!e->gm_ok) return -1;
    using Sp = GSplit<TP>;
    const int npair = (e->n + GW - 1) / GW;
    std::vector<T> smem((size_t)GW * Sp::TOTAL, 0);
    e->fault_word = 0;
    e->st.fault = &e->fault_word; e->st.spin_dyn = Sp::SPIN_LIMIT; e->st.spin_srv = Sp::SPIN_LIMIT;
    e->st.push_step0 = e->push_step; e->push_step += nsteps;
    for (int pr = 0; pr < npair; pr++) {
        std::fill(smem.begin(), smem.end(), T(0));
        volatile int* f = (volatile int*)(smem.data() + Sp::MB);
        f[Sp::MB_CMDSEQ] = 0; f[Sp::MB_DONESEQ] = 0; f[Sp::MB_CMD] = 1; f[Sp::MB_MOK] = 0; f[Sp::MB_MFREE] = 0; f[Sp::MB_PRE] = -1;
        dlemu::run_pair(64,
            [&](int lane) { g_wave_env_step<T, TP, false, true>(lane, pr, pr, npair, smem.data(), &e->gm, e->c, e->st, act, obs, rew, done, (float*)nullptr, (float*)nullptr,
                                                                 e->inj_q.data(), e->inj_v.data(), nullptr, (float*)nullptr, e->eval_mode, nsteps, nullptr); },
            [&](int lane) { g_constraint_server<T, TP>(lane, pr, smem.data(), &e->gm, e->st, nsteps > 1 ? act : nullptr, nsteps); },
            policy, seed + 7919ull * (uint64_t)pr);
    }
    const int fw = e->fault_word;
    e->st.fault = nullptr;
    return fw;          // 0, or the fault bits a wave raised (DL_FAULT_*)
}

template <typename T, typename TP> static void emu_set_rnd(Emu<T, TP>* e, const float* mass_scale, const float* floor_mu, const float* push) {
    const int n = e->n;
    if (e->rnd.empty()) { e->rnd.assign((size_t)5 * n, 0); for (int i = 0; i < n; i++) { e->rnd[i] = 1; e->rnd[(size_t)n + i] = e->gm.floor_friction; } e->st.rnd = e->rnd.data(); }
    for (int i = 0; i < n; i++) {
        if (mass_scale) e->rnd[i] = (T)mass_scale[i];
        if (floor_mu) e->rnd[(size_t)n + i] = (T)floor_mu[i];
        if (push) for (int k = 0; k < 3; k++) e->rnd[(size_t)(2 + k) * n + i] = (T)push[3 * i + k];
    }
}

template <typename T, typename TP> static void emu_set_push_schedule(Emu<T, TP>* e, const float* force, const int32_t* phase, int period, int duration) {
    emu_set_rnd<T, TP>(e, nullptr, nullptr, force);
    e->push_phase.assign(phase, phase + e->n);
    e->st.push_phase = e->push_phase.data(); e->st.push_period = period; e->st.push_dur = duration;
    e->push_step = 0;
}

#define EMU_API(SUF, T, TP) \
    extern "C" int dle_gstep_split_##SUF(void* h, int k, const float* a, float* o, float* r, uint8_t* d, int policy, uint64_t seed) { return emu_gstep_split<T, TP>((Emu<T, TP>*)h, k, a, o, r, d, policy, seed); } \
    extern "C" void dle_set_push_schedule_##SUF(void* h, const float* f, const int32_t* ph, int per, int dur) { emu_set_push_schedule<T, TP>((Emu<T, TP>*)h, f, ph, per, dur); } \
    extern "C" int dle_gforward_##SUF(void* h, const T* u, T* qa, int32_t* nc, int32_t* ne, int32_t* ni) { return emu_gforward<T, TP>((Emu<T, TP>*)h, u, qa, nc, ne, ni); } \
    extern "C" int dle_gstep_##SUF(void* h, int k, const float* a, float* o, float* r, uint8_t* d, float* t, float* tt, float* cu) { return emu_gstep<T, TP>((Emu<T, TP>*)h, k, a, o, r, d, t, tt, cu); } \
    extern "C" void dle_set_rnd_##SUF(void* h, const float* ms, const float* mu, const float* push) { emu_set_rnd<T, TP>((Emu<T, TP>*)h, ms, mu, push); } \
    extern "C" int dle_glds_bytes_##SUF(void) { return (int)(GW * GLds<TP>::TOTAL * sizeof(T)); }                                                                                                   \
    extern "C" void* dle_create_##SUF(const dl_model_desc* d, const dl_refs_desc* r, const dl_config* c, int n) { return emu_create<T, TP>(d, r, c, n); } \
    extern "C" void dle_destroy_##SUF(void* h) { delete (Emu<T, TP>*)h; }                                                     \
    extern "C" void dle_reset_##SUF(void* h, const uint8_t* m, const int32_t* is, const int32_t* ip, float* obs) { emu_reset<T, TP>((Emu<T, TP>*)h, m, is, ip, obs); } \
    extern "C" void dle_step_##SUF(void* h, const float* a, float* o, float* r, uint8_t* d, float* t, float* tt) { emu_step<T, TP>((Emu<T, TP>*)h, a, o, r, d, t, tt); } \
    extern "C" void dle_forward_##SUF(void* h, const T* u, T* qa, int32_t* nc, int32_t* ne, int32_t* ni) { emu_forward<T, TP>((Emu<T, TP>*)h, u, qa, nc, ne, ni); } \
    extern "C" void dle_get_state_##SUF(void* h, T* q, T* v, T* w, int32_t* cur, double* walked) {                        \
        auto* e = (Emu<T, TP>*)h; size_t m = (size_t)TP::NV * e->n;                                                           \
        if (q) memcpy(q, e->qpos.data(), m * sizeof(T)); if (v) memcpy(v, e->qvel.data(), m * sizeof(T));                 \
        if (w) memcpy(w, e->warm.data(), m * sizeof(T)); if (cur) memcpy(cur, e->cur.data(), e->cur.size() * 4);          \
        if (walked) memcpy(walked, e->walked.data(), e->n * 8); }                                                         \
    extern "C" void dle_set_state_##SUF(void* h, const T* q, const T* v, const T* w, const int32_t* cur, const double* walked) { \
        auto* e = (Emu<T, TP>*)h; size_t m = (size_t)TP::NV * e->n;                                                           \
        if (q) memcpy(e->qpos.data(), q, m * sizeof(T)); if (v) memcpy(e->qvel.data(), v, m * sizeof(T));                 \
        if (w) memcpy(e->warm.data(), w, m * sizeof(T)); if (cur) memcpy(e->cur.data(), cur, e->cur.size() * 4);          \
        if (walked) memcpy(e->walked.data(), walked, e->n * 8); }                                                         \
    extern "C" void dle_inject_##SUF(void* h, int i, int flag, const T* q, const T* v) {                                  \
        auto* e = (Emu<T, TP>*)h; e->inj_flags[i] = flag;                                                                     \
        if (q) for (int j = 0; j < TP::NV; j++) { e->inj_q[(size_t)j * e->n + i] = q[j]; e->inj_v[(size_t)j * e->n + i] = v[j]; } } \
    extern "C" void dle_inject_rsi_##SUF(void* h, int i, int s, int p) { auto* e = (Emu<T, TP>*)h; e->inj[i] = s; e->inj[(size_t)e->n + i] = p; } \
    extern "C" void dle_get_zacc_##SUF(void* h, T* out) { auto* e = (Emu<T, TP>*)h; memcpy(out, e->zacc.data(), e->zacc.size() * sizeof(T)); }  \
    extern "C" void dle_set_eval_##SUF(void* h, int on) { ((Emu<T, TP>*)h)->eval_mode = on; }                                \
    extern "C" void dle_mon_##SUF(void* h, int word, double* out) { auto* e = (Emu<T, TP>*)h; memcpy(out, e->mon.data() + (size_t)word * e->n, e->n * 8); }

EMU_API(f32, float, TopoStraight)
#ifndef DL_EMU_ONLY_F32          // (the defect-injection build of tests/test_split_protocol_emu.py needs one instantiation only)
EMU_API(f64, double, TopoStraight)
EMU_API(f64_165, double, TopoWalker165)
EMU_API(f32_165, float, TopoWalker165)
#endif
