// dl_host.hpp -- host-side helpers: descriptor validation and conversion into the kernels'
// parameter blocks.  Shared by the C-ABI (dl_api.hip) and the host-emulation test build.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dl_env.hpp"

namespace dl {

// the compiled kernels are specialised for one kinematic tree; refuse anything else loudly
template <typename TP> inline bool check_topology(const dl_model_desc& d, std::string& why) {
    char buf[256];
    auto fail = [&](const char* what, int idx) { snprintf(buf, sizeof buf, "unsupported model topology: %s[%d]", what, idx); why = buf; return false; };
    if (d.nbody != TP::NB || d.nv != TP::NV || d.nu != TP::NU || d.ngeom != TP::NG || d.nsite != TP::NS) { why = "unsupported model topology: sizes"; return false; }
    for (int b = 0; b < TP::NB; b++) if (d.body_parent[b] != TP::body_parent(b)) return fail("body_parent", b);
    for (int j = 0; j < TP::NV; j++) {
        if (d.jnt_body[j] != TP::dof_body(j)) return fail("jnt_body", j);
        if (d.jnt_type[j] != TP::dof_type(j)) return fail("jnt_type", j);
        if (d.jnt_limited[j] != TP::dof_limited(j)) return fail("jnt_limited", j);
        for (int k = 0; k < 3; k++) {
            const double want = (k == TP::dof_axis(j)) ? (double)TP::dof_sign(j) : 0.0;
            if (std::fabs(d.jnt_axis[j][k] - want) > 1e-12) return fail("jnt_axis", j);
            if (d.jnt_type[j] == DL_JNT_HINGE && d.jnt_pos[j][k] != 0.0) return fail("jnt_pos (hinge anchors must be the body origin)", j);
        }
        if (d.jnt_type[j] == DL_JNT_SLIDE && d.jnt_body[j] != 1) return fail("slide joints must belong to the root body", j);
    }
    for (int g = 0; g < TP::NG; g++) {
        if (d.geom_body[g] != TP::geom_body(g)) return fail("geom_body", g);
        if (d.geom_type[g] != TP::geom_type(g)) return fail("geom_type", g);
    }
    for (int s = 0; s < TP::NS; s++) if (d.site_body[s] != TP::site_body(s)) return fail("site_body", s);
    for (int a = 0; a < TP::NU; a++) if (d.act_dof[a] != TP::act_dof(a)) return fail("act_dof", a);
    if (d.body_pos[1][0] != 0 || d.body_pos[1][1] != 0) { why = "root body must sit above the origin"; return false; }
    if (d.solimp[4] != 1.0 && d.solimp[4] != 2.0) { why = "unsupported solimp power (kernels implement 1 and 2)"; return false; }
    if (!(d.solimp[2] > 1e-15) || d.solimp[0] == d.solimp[1]) { why = "unsupported solimp (flat impedance)"; return false; }
    return true;
}

template <typename T, typename TP> inline void fill_dev_model(const dl_model_desc& d, DevModel<T, TP>& m) {
    std::memset(&m, 0, sizeof m);
    for (int b = 0; b < TP::NB; b++) {
        for (int k = 0; k < 3; k++) { m.body_pos[b][k] = (T)d.body_pos[b][k]; m.body_ipos[b][k] = (T)d.body_ipos[b][k]; m.body_inertia[b][k] = (T)d.body_inertia[b][k]; }
        m.body_mass[b] = (T)d.body_mass[b];
        m.body_invw[b] = (T)d.body_invweight0[b][0];
    }
    for (int j = 0; j < TP::NV; j++) {
        m.qpos0[j] = (T)d.jnt_qpos0[j]; m.range[j][0] = (T)d.jnt_range[j][0]; m.range[j][1] = (T)d.jnt_range[j][1];
        m.damping[j] = (T)d.jnt_damping[j]; m.armature[j] = (T)d.jnt_armature[j]; m.dof_invw[j] = (T)d.dof_invweight0[j];
    }
    for (int g = 0; g < TP::NG; g++) {
        for (int k = 0; k < 3; k++) { m.geom_pos[g][k] = (T)d.geom_pos[g][k]; m.geom_size[g][k] = (T)d.geom_size[g][k]; }
        for (int k = 0; k < 9; k++) m.geom_mat[g][k] = (T)d.geom_mat[g][k];
        m.geom_mu[g] = (T)(d.geom_friction[g] > d.floor_friction ? d.geom_friction[g] : d.floor_friction);
    }
    for (int s = 0; s < TP::NS; s++) for (int k = 0; k < 3; k++) m.site_pos[s][k] = (T)d.site_pos[s][k];
    for (int a = 0; a < TP::NU; a++) {
        m.ctrl_lo[a] = (T)d.act_ctrlrange[a][0]; m.ctrl_hi[a] = (T)d.act_ctrlrange[a][1];
        m.force_lo[a] = (T)d.act_forcerange[a][0]; m.force_hi[a] = (T)d.act_forcerange[a][1];
        m.gear[a] = (T)d.act_gear[a];
    }
    m.timestep = (T)d.timestep;
    m.gravity_z = (T)d.gravity[2];
    double tc = d.solref[0], dr = d.solref[1], dmax = d.solimp[1];
    if (tc < 2 * d.timestep) tc = 2 * d.timestep;   // refsafe
    m.solK = (T)(1.0 / std::fmax(1e-15, dmax * dmax * tc * tc * dr * dr));
    m.solB = (T)(2.0 / std::fmax(1e-15, dmax * tc));
    for (int k = 0; k < 5; k++) m.solimp[k] = (T)d.solimp[k];
    m.meaninertia = (T)d.meaninertia;
    m.tolerance = (T)d.tolerance;
    m.ls_tolerance = (T)d.ls_tolerance;
    // float32 cannot resolve MuJoCo's 1e-8 tolerances: stop the solver / line search at its own resolution
    if (sizeof(T) == 4) { m.tolerance = (T)std::fmax(d.tolerance, 1e-6); m.ls_reltol = (T)1e-5; m.tol_rel = (T)1e-6; }
    else { m.ls_reltol = (T)0; m.tol_rel = (T)0; }
    m.iterations = d.iterations; m.ls_iterations = d.ls_iterations; m.frame_skip = d.frame_skip;
}

template <typename T> inline void fill_dev_cfg(const dl_config& c, const dl_refs_desc& r, DevCfg<T>& o) {
    std::memset(&o, 0, sizeof o);
    for (int k = 0; k < 3; k++) o.rew_w[k] = (T)c.rew_weights[k];
    o.rew_scale = (T)c.rew_scale; o.alive_bonus = (T)c.alive_bonus; o.com_z_min = (T)c.com_z_min;
    o.inv_ctrl_freq = (T)(1.0 / c.ctrl_freq);
    o.ep_dur_max = c.ep_dur_max; o.mirror_policy = c.mirror_policy; o.env_index_base = c.env_index_base; o.seed = c.seed;
    o.intended = c.intended_semantics;
    o.n_steps = r.n_steps; o.total_len = r.total_len; o.stride = r.stride; o.n_rows = r.n_rows;
}

// dl_refs_desc.table is row-major [n_rows][total_len]; the kernels read it sample-major [total_len][n_rows]
inline void transpose_refs(const dl_refs_desc& r, std::vector<double>& out) {
    out.resize((size_t)r.n_rows * r.total_len);
    for (int row = 0; row < r.n_rows; row++)
        for (int col = 0; col < r.total_len; col++) out[(size_t)col * r.n_rows + row] = r.table[(size_t)row * r.total_len + col];
}

// float64 prefix sums of the two desired-velocity rows (rows nv, nv+1 of the table): [2][total_len+1]
inline void loco3d_prefix_sums(const dl_refs_desc& r, int nv, std::vector<double>& out) {
    const size_t L = (size_t)r.total_len;
    out.assign(2 * (L + 1), 0.0);
    for (int k = 0; k < 2; k++) {
        const double* row = r.table + (size_t)(nv + k) * L;
        double acc = 0;
        for (size_t i = 0; i < L; i++) { out[k * (L + 1) + i] = acc; acc += row[i]; }
        out[k * (L + 1) + L] = acc;
    }
}


// ---- table-driven model for the 16-lanes-per-walker kernels (dl_group.hpp)
}  // namespace dl
#if defined(__HIPCC__) || defined(DL_GROUP_EMU)
#include "dl_group_env.hpp"
namespace dl {
template <typename T, typename TP> inline bool fill_group_model(const dl_model_desc& d, GModel<T, TP>& g, std::string& why) {
    using D = GD<TP>;
    constexpr int NX = D::NX;
    std::memset(&g, 0, sizeof g);
    if (d.nv != TP::NV || d.nbody > D::MAXB || d.ngeom > D::MAXB || d.nsite > GL) { why = "model does not fit the 16-lane kernels"; return false; }
    g.nv = d.nv; g.nb = d.nbody; g.nu = d.nu; g.ngeom = d.ngeom; g.nsite = d.nsite; g.frame_skip = d.frame_skip;
    g.iterations = d.iterations; g.ls_iterations = d.ls_iterations;
    g.timestep = (T)d.timestep; g.gravity_z = (T)d.gravity[2];
    double tc = d.solref[0], dr = d.solref[1], dmax = d.solimp[1];
    if (tc < 2 * d.timestep) tc = 2 * d.timestep;
    g.solK = (T)(1.0 / std::fmax(1e-15, dmax * dmax * tc * tc * dr * dr));
    g.solB = (T)(2.0 / std::fmax(1e-15, dmax * tc));
    for (int k = 0; k < 5; k++) g.solimp[k] = (T)d.solimp[k];
    g.solimp_inv[0] = (T)(1.0 / d.solimp[2]); g.solimp_inv[1] = (T)(1.0 / d.solimp[3]); g.solimp_inv[2] = (T)(1.0 / (1.0 - d.solimp[3]));
    g.meaninertia = (T)d.meaninertia; g.tolerance = (T)d.tolerance; g.ls_tolerance = (T)d.ls_tolerance;
    if (sizeof(T) == 4) { g.tolerance = (T)std::fmax(d.tolerance, 1e-6); g.ls_reltol = (T)1e-5; g.tol_rel = (T)1e-6; } else { g.ls_reltol = (T)0; g.tol_rel = (T)0; }
    g.root_z0 = (T)d.body_pos[1][2];
    // bodies
    for (int b = 0; b < d.nbody; b++) {
        for (int k = 0; k < 3; k++) { g.body_pos[b][k] = (T)d.body_pos[b][k]; g.body_ipos[b][k] = (T)d.body_ipos[b][k]; g.body_inertia[b][k] = (T)d.body_inertia[b][k]; }
        g.body_mass[b] = (T)d.body_mass[b]; g.body_invw[b] = (T)d.body_invweight0[b][0];
    }
    // the leading NX dofs (root translations) are carried replicated, the others own lane dof - NX
    for (int t = 0; t < NX; t++) { g.xs_qpos0[t] = (T)d.jnt_qpos0[t]; g.xs_damping[t] = (T)d.jnt_damping[t]; g.xs_armature[t] = (T)d.jnt_armature[t]; }
    for (int l = 0; l < D::NL; l++) {
        const int j = l + NX;
        g.dof_body[l] = d.jnt_body[j]; g.dof_type[l] = d.jnt_type[j]; g.dof_limited[l] = d.jnt_limited[j];
        int ax = -1; double sg = 0;
        for (int k = 0; k < 3; k++) if (std::fabs(d.jnt_axis[j][k]) > 0.5) { ax = k; sg = d.jnt_axis[j][k] > 0 ? 1 : -1; }
        g.dof_axis[l] = ax; g.dof_sign[l] = (T)sg;
        g.qpos0[l] = (T)d.jnt_qpos0[j]; g.range_lo[l] = (T)d.jnt_range[j][0]; g.range_hi[l] = (T)d.jnt_range[j][1];
        g.damping[l] = (T)d.jnt_damping[j]; g.armature[l] = (T)d.jnt_armature[j]; g.dof_invw[l] = (T)d.dof_invweight0[j];
        g.dof_act[l] = -1;
    }
    g.root_last_dof = -1;
    for (int j = 0; j < d.nv; j++) if (d.jnt_body[j] == 1) g.root_last_dof = j;
    for (int a = 0; a < d.nu; a++) {
        const int l = d.act_dof[a] - NX;
        if (l < 0) { why = "a replicated root translation is actuated"; return false; }
        g.dof_act[l] = a;
        g.ctrl_lo[l] = (T)d.act_ctrlrange[a][0]; g.ctrl_hi[l] = (T)d.act_ctrlrange[a][1];
        g.force_lo[l] = (T)d.act_forcerange[a][0]; g.force_hi[l] = (T)d.act_forcerange[a][1]; g.gear[l] = (T)d.act_gear[a];
    }
    // geoms and collision candidates in contact order
    int nc = 0;
    for (int ge = 0; ge < d.ngeom; ge++) {
        g.geom_body[ge] = d.geom_body[ge]; g.geom_type[ge] = d.geom_type[ge];
        for (int k = 0; k < 3; k++) { g.geom_pos[ge][k] = (T)d.geom_pos[ge][k]; g.geom_size[ge][k] = (T)d.geom_size[ge][k]; }
        for (int k = 0; k < 9; k++) g.geom_mat[ge][k] = (T)d.geom_mat[ge][k];
        g.geom_friction[ge] = (T)d.geom_friction[ge];
        const int cnt = d.geom_type[ge] == DL_GEOM_CAPSULE ? 2 : 8;
        for (int k = 0; k < cnt; k++) { if (nc >= GL * D::NPASS) { why = "too many collision candidates"; return false; } g.cand_geom[nc] = ge; g.cand_sub[nc] = k; nc++; }
    }
    g.ncand = nc;
    g.floor_friction = (T)d.floor_friction;
    for (int s = 0; s < d.nsite; s++) { g.site_body[s] = d.site_body[s]; for (int k = 0; k < 3; k++) g.site_pos[s][k] = (T)d.site_pos[s][k]; }
    for (int j = 0; j < GL; j++) g_load_lane<T, TP>(g, j, g.lanes[j]);      // per-lane records, read by the kernels with wide loads
    return true;
}
}  // namespace dl
#endif
