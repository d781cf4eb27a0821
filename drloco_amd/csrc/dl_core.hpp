// dl_core.hpp -- per-walker physics + environment logic of the MI355X DRLoco hot path.
//
// One walker per lane.  Everything a lane needs for one forward-dynamics evaluation lives in
// registers: the kinematic tree is a compile-time constant (`Topo`), all loops over bodies/dofs
// are unrolled through static_for<> so that per-lane arrays are indexed with constants only.
// The dynamically sized data (constraint rows, contacts) lives in LDS, laid out [slot][lane].
//
// Formulation (deliberately different from the CPU oracle's dense J^T I J one):
//   * spatial quantities in world orientation about the root body's origin (keeps fp32
//     magnitudes small and makes parent<->child transforms the identity);
//   * mass matrix by composite rigid bodies, bias by recursive Newton-Euler;
//   * M and H = M + J^T D J share the tree sparsity pattern; both are factorised with the
//     fill-in free L^T D L recursion (leaves first);
//   * constraint Jacobian rows are never stored: J x and J^T f are evaluated through body
//     twists/wrenches, rows are regenerated from the contact point when H is assembled;
//   * Newton solver with exact line search as in MuJoCo ([3P], call site
//     /root/reference/drloco/mujoco/mimic_env.py:83).
//
// The file is `__host__ __device__` so that tests can run the very same source on the CPU
// (tests/host_emu) next to the independent oracle; the product only ever runs it on the GPU.
#pragma once

#include <stdint.h>
#include <math.h>
#include <utility>

#if defined(__HIPCC__)
#define DL_HD __host__ __device__ __forceinline__
#else
#define DL_HD inline __attribute__((always_inline))
#endif

namespace dl {

// ------------------------------------------------------------------------------------------
// compile-time loop
template <int I> struct IC { static constexpr int value = I; constexpr operator int() const { return I; } };
template <typename F, int... Is> DL_HD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, typename F> DL_HD void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ------------------------------------------------------------------------------------------
// topology of walker3d_flat_feet.xml (/root/reference/drloco/mujoco/xml/walker3d_flat_feet.xml:15-80)
struct TopoStraight {
    static constexpr int NB = 8, NV = 14, NU = 8, NG = 7, NS = 8, NLIM = 8;
    static constexpr int MAXCON = 18;              // 5 capsules x 2 + 2 boxes x 4
    static constexpr int MAXROW = NLIM + 4 * MAXCON; // 80
    static constexpr int OBS = 29;
    static constexpr int body_parent_[NB] = {0, 0, 1, 2, 3, 1, 5, 6};
    static constexpr int dof_body_[NV] = {1, 1, 1, 1, 1, 1, 2, 2, 3, 4, 5, 5, 6, 7};
    static constexpr int dof_type_[NV] = {0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};   // 0 slide, 1 hinge
    static constexpr int dof_axis_[NV] = {0, 1, 2, 0, 1, 2, 1, 0, 1, 1, 1, 0, 1, 1};   // body-local coordinate axis
    static constexpr int dof_sign_[NV] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    static constexpr int dof_parent_[NV] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 5, 10, 11, 12};
    static constexpr int dof_limited_[NV] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1};
    static constexpr int geom_body_[NG] = {1, 2, 3, 4, 5, 6, 7};
    static constexpr int geom_type_[NG] = {0, 0, 0, 1, 0, 0, 1};                         // 0 capsule, 1 box
    static constexpr int site_body_[NS] = {4, 4, 4, 4, 7, 7, 7, 7};
    static constexpr int act_dof_[NU] = {6, 7, 8, 9, 10, 11, 12, 13};
    static constexpr int body_parent(int b) { return body_parent_[b]; }
    static constexpr int dof_body(int j) { return dof_body_[j]; }
    static constexpr int dof_type(int j) { return dof_type_[j]; }
    static constexpr int dof_axis(int j) { return dof_axis_[j]; }
    static constexpr int dof_sign(int j) { return dof_sign_[j]; }
    static constexpr int dof_parent(int j) { return dof_parent_[j]; }
    static constexpr int dof_limited(int j) { return dof_limited_[j]; }
    static constexpr int geom_body(int g) { return geom_body_[g]; }
    static constexpr int geom_type(int g) { return geom_type_[g]; }
    static constexpr int site_body(int s) { return site_body_[s]; }
    static constexpr int act_dof(int a) { return act_dof_[a]; }
    // dof j is dof i itself or one of its ancestors in the dof tree
    static constexpr bool dof_anc(int i, int j) {
        while (i >= 0) { if (i == j) return true; i = dof_parent_[i]; }
        return false;
    }
    // dof j moves body b
    static constexpr bool body_anc(int b, int j) {
        while (b > 0) { if (dof_body_[j] == b) return true; b = body_parent_[b]; }
        return false;
    }
    // last dof of body b (or of its nearest ancestor that has dofs)
    static constexpr int body_last_dof(int b) {
        while (b > 0) {
            int last = -1;
            for (int j = 0; j < NV; j++) if (dof_body_[j] == b) last = j;
            if (last >= 0) return last;
            b = body_parent_[b];
        }
        return -1;
    }
    // bitmask over dofs that move body b (runtime use)
    static constexpr uint32_t body_mask(int b) {
        uint32_t m = 0;
        for (int j = 0; j < NV; j++) if (body_anc(b, j)) m |= 1u << j;
        return m;
    }
    // observation mirroring (/root/reference/drloco/mujoco/mimic_env.py:452-463)
    static constexpr int obs_perm_[OBS] = {0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 7, 8, 9, 10, 15, 16, 17, 18, 19, 20, 25, 26, 27, 28, 21, 22, 23, 24};
    static constexpr int obs_neg_[OBS] = {0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0};
    static constexpr int obs_perm(int k) { return obs_perm_[k]; }
    static constexpr int obs_neg(int k) { return obs_neg_[k]; }
    // action mirroring (:483-489)
    static constexpr int act_perm_[NU] = {4, 5, 6, 7, 0, 1, 2, 3};
    static constexpr int act_neg_[NU] = {0, 1, 0, 0, 0, 1, 0, 0};
    static constexpr int act_perm(int k) { return act_perm_[k]; }
    static constexpr int act_neg(int k) { return act_neg_[k]; }
};

// ------------------------------------------------------------------------------------------
// numeric model parameters (uniform across lanes: kernel argument -> SGPRs / scalar loads)
template <typename T, typename TP> struct DevModel {
    T body_pos[TP::NB][3], body_mass[TP::NB], body_ipos[TP::NB][3], body_inertia[TP::NB][3];
    T qpos0[TP::NV], range[TP::NV][2], damping[TP::NV], armature[TP::NV], dof_invw[TP::NV];
    T geom_pos[TP::NG][3], geom_mat[TP::NG][9], geom_size[TP::NG][3], geom_mu[TP::NG], body_invw[TP::NB];
    T site_pos[TP::NS][3];
    T ctrl_lo[TP::NU], ctrl_hi[TP::NU], force_lo[TP::NU], force_hi[TP::NU], gear[TP::NU];
    T timestep, gravity_z, solK, solB, solimp[5], meaninertia, tolerance, ls_tolerance, ls_reltol;
    int32_t iterations, ls_iterations, frame_skip;
};

// environment constants (drloco/config/hypers.py, config.py) + reference table view
template <typename T> struct DevCfg {
    T rew_w[3], rew_scale, alive_bonus, com_z_min, inv_ctrl_freq;
    int32_t ep_dur_max, mirror_policy, env_index_base;
    uint64_t seed;
    int32_t n_steps, total_len, stride;
    const T* table;            // [2*NV][total_len]
    const int32_t* step_off;   // [n_steps+1]
    const int32_t* step_is_left;
    const T* step_vel;
};

// ------------------------------------------------------------------------------------------
// math helpers
template <typename T> struct V3 { T x, y, z; };
template <typename T> DL_HD V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> DL_HD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> DL_HD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> DL_HD V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> DL_HD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> DL_HD V3<T> cross(V3<T> a, V3<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <typename T> DL_HD T comp(V3<T> a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

DL_HD float dl_sqrt(float x) { return sqrtf(x); }
DL_HD double dl_sqrt(double x) { return sqrt(x); }
DL_HD float dl_abs(float x) { return fabsf(x); }
DL_HD double dl_abs(double x) { return fabs(x); }
DL_HD float dl_exp(float x) { return expf(x); }
DL_HD double dl_exp(double x) { return exp(x); }
DL_HD float dl_pow(float x, float y) { return powf(x, y); }
DL_HD double dl_pow(double x, double y) { return pow(x, y); }
DL_HD void dl_sincos(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }
DL_HD void dl_sincos(double x, double& s, double& c) { s = sin(x); c = cos(x); }
template <typename T> DL_HD T dl_max(T a, T b) { return a > b ? a : b; }
template <typename T> DL_HD T dl_min(T a, T b) { return a < b ? a : b; }
template <typename T> DL_HD T dl_clamp(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
template <typename T> DL_HD bool dl_bad(T x) { return !(x == x) || x > T(1e10) || x < T(-1e10); }

// symmetric 3x3
template <typename T> struct S3 { T xx, xy, xz, yy, yz, zz; };
template <typename T> DL_HD V3<T> mul(const S3<T>& I, V3<T> w) {
    return {I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z};
}
// spatial inertia about the reference point: mass, first moment h = m c, rotational inertia
template <typename T> struct SI { T m; V3<T> h; S3<T> I; };
template <typename T> DL_HD void si_add(SI<T>& a, const SI<T>& b) {
    a.m += b.m; a.h = a.h + b.h;
    a.I.xx += b.I.xx; a.I.xy += b.I.xy; a.I.xz += b.I.xz; a.I.yy += b.I.yy; a.I.yz += b.I.yz; a.I.zz += b.I.zz;
}
// spatial motion / force vectors: (angular, linear)
template <typename T> struct SV { V3<T> w, v; };
template <typename T> DL_HD SV<T> operator+(SV<T> a, SV<T> b) { return {a.w + b.w, a.v + b.v}; }
// I * motion -> force (n: moment about the reference point, f: linear)
template <typename T> DL_HD SV<T> si_mul(const SI<T>& s, SV<T> m) { return {mul(s.I, m.w) + cross(s.h, m.v), s.m * m.v + cross(m.w, s.h)}; }
template <typename T> DL_HD T sdot(SV<T> motion, SV<T> force) { return dot(motion.w, force.w) + dot(motion.v, force.v); }

// ------------------------------------------------------------------------------------------
// LDS (or host array) view of one lane's dynamic storage: element (arr, slot) at base[(arr*cap+slot)*stride]
template <typename T> struct LaneMem {
    T* base; int stride;
    DL_HD T& operator()(int idx) const { return base[(size_t)idx * stride]; }
};
// slot layout inside LaneMem
template <typename TP> struct MemLayout {
    static constexpr int ROW_D = 0, ROW_JAREF = TP::MAXROW, ROW_JV = 2 * TP::MAXROW;
    static constexpr int CON_PX = 3 * TP::MAXROW;            // contact point relative to the root origin
    static constexpr int CON_PY = CON_PX + TP::MAXCON, CON_PZ = CON_PY + TP::MAXCON;
    static constexpr int CON_TX = CON_PZ + TP::MAXCON, CON_TY = CON_TX + TP::MAXCON;   // first tangent (unit, in the floor plane)
    static constexpr int CON_MU = CON_TY + TP::MAXCON, CON_DIST = CON_MU + TP::MAXCON;
    static constexpr int TOTAL = CON_DIST + TP::MAXCON;       // elements of T per lane
};

// ------------------------------------------------------------------------------------------
// per-evaluation kinematic state kept in registers
template <typename T, typename TP> struct Kin {
    V3<T> RX[TP::NB], RY[TP::NB], RZ[TP::NB];  // body frame columns (world)
    V3<T> pos[TP::NB];                         // body origin relative to the root body origin
    V3<T> axis[TP::NV];                        // dof axis (world)
    T rootz;                                   // absolute height of the root body origin
};

template <typename T, typename TP> DL_HD V3<T> dof_anchor(const Kin<T, TP>& k, int j) { return k.pos[TP::dof_body(j)]; }

// [3P] mj_kinematics for slide/hinge trees whose slides sit on the root body
template <typename T, typename TP>
DL_HD void kinematics(const DevModel<T, TP>& m, const T (&q)[TP::NV], Kin<T, TP>& k) {
    k.RX[0] = mk<T>(1, 0, 0); k.RY[0] = mk<T>(0, 1, 0); k.RZ[0] = mk<T>(0, 0, 1);
    k.pos[0] = mk<T>(0, 0, 0);
    k.rootz = m.body_pos[1][2];
    static_for<TP::NB - 1>([&](auto bi) {
        constexpr int b = bi.value + 1, p = TP::body_parent(b);
        V3<T> X, Y, Z, pos;
        if constexpr (p == 0) { X = mk<T>(1, 0, 0); Y = mk<T>(0, 1, 0); Z = mk<T>(0, 0, 1); pos = mk<T>(0, 0, 0); }
        else {
            X = k.RX[p]; Y = k.RY[p]; Z = k.RZ[p];
            pos = k.pos[p] + m.body_pos[b][0] * X + m.body_pos[b][1] * Y + m.body_pos[b][2] * Z;
        }
        static_for<TP::NV>([&](auto ji) {
            constexpr int j = ji.value;
            if constexpr (TP::dof_body(j) == b) {
                constexpr int ax = TP::dof_axis(j);
                const T sg = T(TP::dof_sign(j));
                V3<T> a = ax == 0 ? X : (ax == 1 ? Y : Z);
                k.axis[j] = sg * a;
                const T dq = q[j] - m.qpos0[j];
                if constexpr (TP::dof_type(j) == 0) {
                    k.rootz += sg * a.z * dq;          // only the height matters: the floor is an infinite plane
                } else {
                    T s, c;
                    dl_sincos(sg * dq, s, c);
                    if constexpr (ax == 0) { V3<T> y2 = c * Y + s * Z, z2 = c * Z - s * Y; Y = y2; Z = z2; }
                    else if constexpr (ax == 1) { V3<T> x2 = c * X - s * Z, z2 = s * X + c * Z; X = x2; Z = z2; }
                    else { V3<T> x2 = c * X + s * Y, y2 = c * Y - s * X; X = x2; Y = y2; }
                }
            }
        });
        k.RX[b] = X; k.RY[b] = Y; k.RZ[b] = Z; k.pos[b] = pos;
    });
}

template <typename T, typename TP> DL_HD V3<T> body_point(const Kin<T, TP>& k, int b, const T* local) {
    return k.pos[b] + local[0] * k.RX[b] + local[1] * k.RY[b] + local[2] * k.RZ[b];
}

// motion subspace of dof j about the reference point
template <typename T, typename TP, int j> DL_HD SV<T> dof_S(const Kin<T, TP>& k) {
    if constexpr (TP::dof_type(j) == 0) return {mk<T>(0, 0, 0), k.axis[j]};
    else return {k.axis[j], cross(k.pos[TP::dof_body(j)], k.axis[j])};
}

// ------------------------------------------------------------------------------------------
// sparse symmetric matrix with the tree pattern, stored as a full lower array whose unused
// entries are never touched (and therefore never materialised)
template <typename T, typename TP> struct TreeMat { T a[TP::NV][TP::NV]; };

// in-place L^T D L (Featherstone): afterwards a[k][k] = D_k, a[k][i] (i ancestor of k) = L_ki
template <typename T, typename TP> DL_HD void ltdl_factor(TreeMat<T, TP>& H) {
    static_for<TP::NV>([&](auto kr) {
        constexpr int k = TP::NV - 1 - kr.value;
        const T inv = T(1) / H.a[k][k];
        static_for<TP::NV>([&](auto ir) {
            constexpr int i = TP::NV - 1 - ir.value;        // descending: nearest ancestor first
            if constexpr (i < k && TP::dof_anc(k, i)) {
                const T a = H.a[k][i] * inv;
                static_for<TP::NV>([&](auto jr) {
                    constexpr int j = jr.value;
                    if constexpr (j <= i && TP::dof_anc(i, j)) H.a[i][j] -= a * H.a[k][j];
                });
                H.a[k][i] = a;
            }
        });
    });
}
// x <- (L^T D L)^-1 x
template <typename T, typename TP> DL_HD void ltdl_solve(const TreeMat<T, TP>& H, T (&x)[TP::NV]) {
    static_for<TP::NV>([&](auto kr) {
        constexpr int k = TP::NV - 1 - kr.value;
        static_for<TP::NV>([&](auto ir) {
            constexpr int i = ir.value;
            if constexpr (i < k && TP::dof_anc(k, i)) x[i] -= H.a[k][i] * x[k];
        });
    });
    static_for<TP::NV>([&](auto kr) { constexpr int k = kr.value; x[k] = x[k] / H.a[k][k]; });
    static_for<TP::NV>([&](auto kr) {
        constexpr int k = kr.value;
        static_for<TP::NV>([&](auto ir) {
            constexpr int i = ir.value;
            if constexpr (i < k && TP::dof_anc(k, i)) x[k] -= H.a[k][i] * x[i];
        });
    });
}
// r = M x for the sparse symmetric M (lower stored)
template <typename T, typename TP> DL_HD void treemat_mul(const TreeMat<T, TP>& M, const T (&x)[TP::NV], T (&r)[TP::NV]) {
    static_for<TP::NV>([&](auto ir) { r[ir.value] = M.a[ir.value][ir.value] * x[ir.value]; });
    static_for<TP::NV>([&](auto ir) {
        constexpr int i = ir.value;
        static_for<TP::NV>([&](auto jr) {
            constexpr int j = jr.value;
            if constexpr (j < i && TP::dof_anc(i, j)) { r[i] += M.a[i][j] * x[j]; r[j] += M.a[i][j] * x[i]; }
        });
    });
}

// ------------------------------------------------------------------------------------------
// [3P] mj_crb + mj_rne(bias): mass matrix (lower, tree pattern) and bias forces
template <typename T, typename TP>
DL_HD void inertia_and_bias(const DevModel<T, TP>& m, const Kin<T, TP>& k, const T (&v)[TP::NV], TreeMat<T, TP>& M, T (&bias)[TP::NV]) {
    SI<T> Ib[TP::NB];
    static_for<TP::NB - 1>([&](auto bi) {
        constexpr int b = bi.value + 1;
        const V3<T> c = body_point<T, TP>(k, b, m.body_ipos[b]);
        const T mass = m.body_mass[b];
        const T i0 = m.body_inertia[b][0], i1 = m.body_inertia[b][1], i2 = m.body_inertia[b][2];
        const V3<T> X = k.RX[b], Y = k.RY[b], Z = k.RZ[b];
        const T cc = dot(c, c);
        SI<T> s;
        s.m = mass; s.h = mass * c;
        s.I.xx = i0 * X.x * X.x + i1 * Y.x * Y.x + i2 * Z.x * Z.x + mass * (cc - c.x * c.x);
        s.I.yy = i0 * X.y * X.y + i1 * Y.y * Y.y + i2 * Z.y * Z.y + mass * (cc - c.y * c.y);
        s.I.zz = i0 * X.z * X.z + i1 * Y.z * Y.z + i2 * Z.z * Z.z + mass * (cc - c.z * c.z);
        s.I.xy = i0 * X.x * X.y + i1 * Y.x * Y.y + i2 * Z.x * Z.y - mass * c.x * c.y;
        s.I.xz = i0 * X.x * X.z + i1 * Y.x * Y.z + i2 * Z.x * Z.z - mass * c.x * c.z;
        s.I.yz = i0 * X.y * X.z + i1 * Y.y * Y.z + i2 * Z.y * Z.z - mass * c.y * c.z;
        Ib[b] = s;
    });
    // ---- recursive Newton-Euler with qacc = 0; gravity enters as a base acceleration of -g
    SV<T> vel[TP::NV], acc[TP::NV];
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value, p = TP::dof_parent(j);
        const SV<T> S = dof_S<T, TP, j>(k);
        const SV<T> vJ = {v[j] * S.w, v[j] * S.v};
        if constexpr (p < 0) {
            vel[j] = vJ;
            acc[j] = {mk<T>(0, 0, 0), mk<T>(0, 0, -m.gravity_z)};
        } else {
            vel[j] = vel[p] + vJ;
            acc[j] = {acc[p].w + cross(vel[p].w, vJ.w), acc[p].v + cross(vel[p].w, vJ.v) + cross(vel[p].v, vJ.w)};
        }
    });
    SV<T> F[TP::NB];
    static_for<TP::NB - 1>([&](auto bi) {
        constexpr int b = bi.value + 1, ld = TP::body_last_dof(b);
        const SV<T> Iv = si_mul(Ib[b], vel[ld]);
        const SV<T> Ia = si_mul(Ib[b], acc[ld]);
        F[b] = {Ia.w + cross(vel[ld].w, Iv.w) + cross(vel[ld].v, Iv.v), Ia.v + cross(vel[ld].w, Iv.v)};
    });
    // accumulate wrenches and composite inertias towards the root (same reference point: plain sums)
    static_for<TP::NB - 2>([&](auto bi) {
        constexpr int b = TP::NB - 1 - bi.value, p = TP::body_parent(b);
        if constexpr (p > 0) { F[p] = F[p] + F[b]; si_add(Ib[p], Ib[b]); }
    });
    static_for<TP::NV>([&](auto ii) {
        constexpr int i = ii.value, b = TP::dof_body(i);
        const SV<T> S = dof_S<T, TP, i>(k);
        bias[i] = sdot(S, F[b]);
        const SV<T> f = si_mul(Ib[b], S);
        static_for<TP::NV>([&](auto ji) {
            constexpr int j = ji.value;
            if constexpr (j <= i && TP::dof_anc(i, j)) {
                const SV<T> Sj = dof_S<T, TP, j>(k);
                M.a[i][j] = sdot(Sj, f);
            }
        });
        M.a[i][i] += m.armature[i];
    });
}

// ------------------------------------------------------------------------------------------
// constraint bookkeeping of one evaluation
template <typename TP> struct EfcInfo {
    int nlim, ncon, nefc;
    uint64_t lim_code;   // 5 bits per limit row: dof (4) | upper-side flag (1)
    uint64_t con_body;   // 3 bits per contact: body id
};

// twist of every body under generalised velocity x (for J x) -- returns vel per dof
template <typename T, typename TP> DL_HD void body_twists(const Kin<T, TP>& k, const T (&x)[TP::NV], SV<T> (&vel)[TP::NV]) {
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value, p = TP::dof_parent(j);
        const SV<T> S = dof_S<T, TP, j>(k);
        const SV<T> vJ = {x[j] * S.w, x[j] * S.v};
        if constexpr (p < 0) vel[j] = vJ; else vel[j] = vel[p] + vJ;
    });
}
template <typename T, typename TP> DL_HD SV<T> twist_of_body(const SV<T> (&vel)[TP::NV], int b) {
    SV<T> r = {mk<T>(0, 0, 0), mk<T>(0, 0, 0)};
    static_for<TP::NB - 1>([&](auto bi) {
        constexpr int bb = bi.value + 1;
        if (b == bb) r = vel[TP::body_last_dof(bb)];
    });
    return r;
}

// [3P] solimp sigmoid (getimpedance)
template <typename T> DL_HD T impedance(const T* si, T pos) {
    T x = dl_abs(pos) / si[2];
    if (x >= T(1)) return si[1];
    if (x <= T(0)) return si[0];
    T y;
    if (si[4] == T(1)) y = x;
    else if (si[4] == T(2)) y = (x <= si[3]) ? x * x / si[3] : T(1) - (T(1) - x) * (T(1) - x) / (T(1) - si[3]);
    else if (x <= si[3]) y = dl_pow(x, si[4]) / dl_pow(si[3], si[4] - T(1));
    else y = T(1) - dl_pow(T(1) - x, si[4]) / dl_pow(T(1) - si[3], si[4] - T(1));
    return si[0] + y * (si[1] - si[0]);
}

// [3P] mj_collision (plane vs capsule / box) + mj_makeConstraint + mj_makeImpedance +
// mj_referenceConstraint.  Writes contacts and rows (D, Jaref := -aref) into lane memory.
template <typename T, typename TP>
DL_HD void make_constraints(const DevModel<T, TP>& m, const Kin<T, TP>& k, const T (&q)[TP::NV], const T (&v)[TP::NV],
                            const LaneMem<T>& mem, EfcInfo<TP>& e) {
    using L = MemLayout<TP>;
    e.nlim = 0; e.ncon = 0; e.lim_code = 0; e.con_body = 0;
    // joint limits: rows +-e_j
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value;
        if constexpr (TP::dof_limited(j)) {
            const T dlo = q[j] - m.range[j][0], dhi = m.range[j][1] - q[j];
            const bool lo = dlo < T(0), hi = dhi < T(0);
            if (lo || hi) {
                const T dist = lo ? dlo : dhi;
                const T imp = impedance(m.solimp, dist);
                const T R = dl_max(T(1e-15), (T(1) - imp) * m.dof_invw[j] / imp);
                const T vel = lo ? v[j] : -v[j];
                const int r = e.nlim;
                mem(L::ROW_D + r) = T(1) / R;
                mem(L::ROW_JAREF + r) = m.solB * vel + m.solK * imp * dist;      // = -aref
                e.lim_code |= (uint64_t)(j | (lo ? 0 : 16)) << (5 * r);
                e.nlim = r + 1;
            }
        }
    });
    // contacts
    auto add_contact = [&](int body, T mu, T invw, V3<T> p, T dist, T tx, T ty) {
        const int c = e.ncon;
        if (c >= TP::MAXCON) return;
        mem(L::CON_PX + c) = p.x; mem(L::CON_PY + c) = p.y; mem(L::CON_PZ + c) = p.z;
        mem(L::CON_TX + c) = tx; mem(L::CON_TY + c) = ty; mem(L::CON_MU + c) = mu; mem(L::CON_DIST + c) = dist;
        e.con_body |= (uint64_t)body << (3 * c);
        e.ncon = c + 1;
        (void)invw;
    };
    static_for<TP::NG>([&](auto gi) {
        constexpr int g = gi.value, b = TP::geom_body(g);
        const V3<T> gp = body_point<T, TP>(k, b, m.geom_pos[g]);
        const T* gm = m.geom_mat[g];
        if constexpr (TP::geom_type(g) == 0) {
            // capsule: axis = third column of the geom frame
            const V3<T> ax = gm[2] * k.RX[b] + gm[5] * k.RY[b] + gm[8] * k.RZ[b];
            const T rad = m.geom_size[g][0], half = m.geom_size[g][1];
            // tangent: capsule axis projected into the floor plane (mju_makeFrame)
            T tx = ax.x, ty = ax.y;
            const T n2 = tx * tx + ty * ty;
            if (n2 < T(1e-30)) { tx = T(1); ty = T(0); }
            else { const T inv = T(1) / dl_sqrt(n2); tx *= inv; ty *= inv; }
            for (int s = 0; s < 2; s++) {
                const V3<T> c = gp + (s == 0 ? half : -half) * ax;
                const T dist = k.rootz + c.z - rad;
                if (dist < T(0)) add_contact(b, m.geom_mu[g], m.body_invw[b], mk<T>(c.x, c.y, c.z - (rad + T(0.5) * dist)), dist, tx, ty);
            }
        } else {
            const V3<T> ex = gm[0] * k.RX[b] + gm[3] * k.RY[b] + gm[6] * k.RZ[b];
            const V3<T> ey = gm[1] * k.RX[b] + gm[4] * k.RY[b] + gm[7] * k.RZ[b];
            const V3<T> ez = gm[2] * k.RX[b] + gm[5] * k.RY[b] + gm[8] * k.RZ[b];
            int cnt = 0;
            static_for<8>([&](auto ci) {
                constexpr int i = ci.value;
                const T sx = (i & 1) ? m.geom_size[g][0] : -m.geom_size[g][0];
                const T sy = (i & 2) ? m.geom_size[g][1] : -m.geom_size[g][1];
                const T sz = (i & 4) ? m.geom_size[g][2] : -m.geom_size[g][2];
                const V3<T> corner = sx * ex + sy * ey + sz * ez;
                const T dist = k.rootz + gp.z + corner.z;
                if (cnt < 4 && dist < T(0) && !(corner.z > T(0))) {
                    add_contact(b, m.geom_mu[g], m.body_invw[b], mk<T>(gp.x + corner.x, gp.y + corner.y, gp.z + corner.z - T(0.5) * dist), dist, T(0), T(1));
                    cnt++;
                }
            });
        }
    });
    // contact rows: 4 pyramid edges n +- mu t1, n +- mu t2; t1 = (tx, ty, 0), t2 = n x t1 = (-ty, tx, 0)
    SV<T> vel[TP::NV];
    body_twists<T, TP>(k, v, vel);
    for (int c = 0; c < e.ncon; c++) {
        const int body = (int)((e.con_body >> (3 * c)) & 7);
        const V3<T> p = mk<T>(mem(L::CON_PX + c), mem(L::CON_PY + c), mem(L::CON_PZ + c));
        const T tx = mem(L::CON_TX + c), ty = mem(L::CON_TY + c), mu = mem(L::CON_MU + c), dist = mem(L::CON_DIST + c);
        T invw = T(0);
        static_for<TP::NB - 1>([&](auto bi) { if (body == bi.value + 1) invw = m.body_invw[bi.value + 1]; });
        const SV<T> tw = twist_of_body<T, TP>(vel, body);
        const V3<T> pv = tw.v + cross(tw.w, p);
        const T vn = pv.z, vt1 = tx * pv.x + ty * pv.y, vt2 = -ty * pv.x + tx * pv.y;
        const T imp = impedance(m.solimp, dist);
        const T diag = invw * (T(1) + mu * mu);
        const T R = T(2) * mu * mu * dl_max(T(1e-15), (T(1) - imp) * diag / imp);
        const T D = T(1) / R, kd = m.solK * imp * dist;
        const int r = e.nlim + 4 * c;
        mem(L::ROW_D + r) = D; mem(L::ROW_D + r + 1) = D; mem(L::ROW_D + r + 2) = D; mem(L::ROW_D + r + 3) = D;
        mem(L::ROW_JAREF + r) = m.solB * (vn + mu * vt1) + kd;
        mem(L::ROW_JAREF + r + 1) = m.solB * (vn - mu * vt1) + kd;
        mem(L::ROW_JAREF + r + 2) = m.solB * (vn + mu * vt2) + kd;
        mem(L::ROW_JAREF + r + 3) = m.solB * (vn - mu * vt2) + kd;
    }
    e.nefc = e.nlim + 4 * e.ncon;
}

// out[r] (+)= (J x)[r] for all rows; arr selects the destination array in lane memory
template <typename T, typename TP, bool ACCUM>
DL_HD void mul_J(const Kin<T, TP>& k, const EfcInfo<TP>& e, const LaneMem<T>& mem, const T (&x)[TP::NV], int arr) {
    using L = MemLayout<TP>;
    for (int r = 0; r < e.nlim; r++) {
        const int code = (int)((e.lim_code >> (5 * r)) & 31), j = code & 15;
        T xj = T(0);
        static_for<TP::NV>([&](auto ji) { if (j == ji.value) xj = x[ji.value]; });
        const T val = (code & 16) ? -xj : xj;
        if (ACCUM) mem(arr + r) += val; else mem(arr + r) = val;
    }
    if (e.ncon == 0) return;
    SV<T> vel[TP::NV];
    body_twists<T, TP>(k, x, vel);
    for (int c = 0; c < e.ncon; c++) {
        const int body = (int)((e.con_body >> (3 * c)) & 7);
        const V3<T> p = mk<T>(mem(L::CON_PX + c), mem(L::CON_PY + c), mem(L::CON_PZ + c));
        const T tx = mem(L::CON_TX + c), ty = mem(L::CON_TY + c), mu = mem(L::CON_MU + c);
        const SV<T> tw = twist_of_body<T, TP>(vel, body);
        const V3<T> pv = tw.v + cross(tw.w, p);
        const T vn = pv.z, vt1 = mu * (tx * pv.x + ty * pv.y), vt2 = mu * (-ty * pv.x + tx * pv.y);
        const int r = arr + e.nlim + 4 * c;
        if (ACCUM) { mem(r) += vn + vt1; mem(r + 1) += vn - vt1; mem(r + 2) += vn + vt2; mem(r + 3) += vn - vt2; }
        else { mem(r) = vn + vt1; mem(r + 1) = vn - vt1; mem(r + 2) = vn + vt2; mem(r + 3) = vn - vt2; }
    }
}

// Jacobian of contact c in the contact frame: jn, jt1, jt2 (dense over dofs; zero where the dof
// does not move the body)
template <typename T, typename TP>
DL_HD void contact_jac(const Kin<T, TP>& k, int body, V3<T> p, T tx, T ty, T (&jn)[TP::NV], T (&jt1)[TP::NV], T (&jt2)[TP::NV]) {
    uint32_t mask = 0;
    static_for<TP::NB - 1>([&](auto bi) { if (body == bi.value + 1) mask = TP::body_mask(bi.value + 1); });
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value;
        V3<T> w;
        if constexpr (TP::dof_type(j) == 0) w = k.axis[j];
        else w = cross(k.axis[j], p - k.pos[TP::dof_body(j)]);
        const T on = ((mask >> j) & 1u) ? T(1) : T(0);
        jn[j] = on * w.z; jt1[j] = on * (tx * w.x + ty * w.y); jt2[j] = on * (-ty * w.x + tx * w.y);
    });
}

// ------------------------------------------------------------------------------------------
// line search on the piecewise quadratic along `search` (MuJoCo's bracketing scheme)
template <typename T> struct LsPoint { T alpha, cost, d1, d2; };

template <typename T, typename TP>
DL_HD LsPoint<T> ls_eval(const LaneMem<T>& mem, int nefc, T g0, T g1, T g2, T alpha) {
    using L = MemLayout<TP>;
    T cost = g0 + alpha * g1 + alpha * alpha * g2, d1 = g1 + T(2) * alpha * g2, d2 = T(2) * g2;
    for (int r = 0; r < nefc; r++) {
        const T jv = mem(L::ROW_JV + r);
        const T x = mem(L::ROW_JAREF + r) + alpha * jv;
        if (x < T(0)) {
            const T D = mem(L::ROW_D + r);
            cost += T(0.5) * D * x * x; d1 += D * x * jv; d2 += D * jv * jv;
        }
    }
    return {alpha, cost, d1, d2};
}

template <typename T, typename TP>
DL_HD bool ls_update_bracket(const LaneMem<T>& mem, int nefc, T g0, T g1, T g2, LsPoint<T>& p, const LsPoint<T> (&cand)[3], LsPoint<T>& pnext) {
    bool flag = false;
    for (int i = 0; i < 3; i++) {
        if (p.d1 < T(0) && cand[i].d1 < T(0) && p.d1 < cand[i].d1) { p = cand[i]; flag = true; }
        else if (p.d1 > T(0) && cand[i].d1 > T(0) && p.d1 > cand[i].d1) { p = cand[i]; flag = true; }
    }
    if (flag) pnext = ls_eval<T, TP>(mem, nefc, g0, g1, g2, p.alpha - p.d1 / p.d2);
    return flag;
}

template <typename T, typename TP>
DL_HD T linesearch(const LaneMem<T>& mem, int nefc, T g0, T g1, T g2, T gtol, int maxit) {
    LsPoint<T> p0 = ls_eval<T, TP>(mem, nefc, g0, g1, g2, T(0));
    LsPoint<T> p1 = ls_eval<T, TP>(mem, nefc, g0, g1, g2, -p0.d1 / p0.d2);
    if (p0.cost < p1.cost) p1 = p0;
    if (dl_abs(p1.d1) < gtol) return p1.alpha;
    const T dir = p1.d1 < T(0) ? T(1) : T(-1);
    LsPoint<T> p2 = p1, pmid, p1next, p2next;
    bool p2update = false;
    int it = 0;
    while (p1.d1 * dir <= -gtol && it < maxit) {
        p2 = p1; p2update = true;
        p1 = ls_eval<T, TP>(mem, nefc, g0, g1, g2, p1.alpha - p1.d1 / p1.d2);
        it++;
        if (dl_abs(p1.d1) < gtol) return p1.alpha;
    }
    if (it >= maxit || !p2update) return p1.alpha;
    p2next = p1;
    p1next = ls_eval<T, TP>(mem, nefc, g0, g1, g2, p1.alpha - p1.d1 / p1.d2);
    while (it < maxit) {
        pmid = ls_eval<T, TP>(mem, nefc, g0, g1, g2, T(0.5) * (p1.alpha + p2.alpha));
        it++;
        const LsPoint<T> cand[3] = {p1next, p2next, pmid};
        int best = -1;
        T bestcost = T(0);
        for (int i = 0; i < 3; i++)
            if (dl_abs(cand[i].d1) < gtol && (best < 0 || cand[i].cost < bestcost)) { best = i; bestcost = cand[i].cost; }
        if (best >= 0) return best == 0 ? cand[0].alpha : (best == 1 ? cand[1].alpha : cand[2].alpha);
        const bool b1 = ls_update_bracket<T, TP>(mem, nefc, g0, g1, g2, p1, cand, p1next);
        const bool b2 = ls_update_bracket<T, TP>(mem, nefc, g0, g1, g2, p2, cand, p2next);
        if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : T(0);
    }
    if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
    if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
    return T(0);
}

// ------------------------------------------------------------------------------------------
// solver state update at the current qacc: cost, gradient, Hessian factor, Newton direction.
// ROW_JAREF holds J qacc - aref.
template <typename T, typename TP>
DL_HD void solver_update(const Kin<T, TP>& k, const EfcInfo<TP>& e, const LaneMem<T>& mem, const TreeMat<T, TP>& M,
                         const T (&qacc)[TP::NV], const T (&Ma)[TP::NV], const T (&smooth)[TP::NV], const T (&qacc_smooth)[TP::NV],
                         T& cost, T& gauss, T (&grad)[TP::NV], T (&Mgrad)[TP::NV]) {
    using L = MemLayout<TP>;
    TreeMat<T, TP> H;
    static_for<TP::NV>([&](auto ii) {
        constexpr int i = ii.value;
        static_for<TP::NV>([&](auto ji) {
            constexpr int j = ji.value;
            if constexpr (j <= i && TP::dof_anc(i, j)) H.a[i][j] = M.a[i][j];
        });
    });
    T fcon[TP::NV];
    static_for<TP::NV>([&](auto ii) { fcon[ii.value] = T(0); });
    T c = T(0);
    for (int r = 0; r < e.nlim; r++) {
        const T jar = mem(L::ROW_JAREF + r);
        if (jar < T(0)) {
            const T D = mem(L::ROW_D + r);
            const int code = (int)((e.lim_code >> (5 * r)) & 31), j = code & 15;
            const T f = (code & 16) ? D * jar : -D * jar;      // J^T f with J = -+1
            c += T(0.5) * D * jar * jar;
            static_for<TP::NV>([&](auto ji) {
                if constexpr (TP::dof_limited(ji.value)) if (j == ji.value) { fcon[ji.value] += f; H.a[ji.value][ji.value] += D; }
            });
        }
    }
    for (int cc = 0; cc < e.ncon; cc++) {
        const int r0 = e.nlim + 4 * cc;
        T jar[4], Dr[4];
        bool any = false;
        for (int s = 0; s < 4; s++) { jar[s] = mem(L::ROW_JAREF + r0 + s); Dr[s] = mem(L::ROW_D + r0 + s); any = any || jar[s] < T(0); }
        if (!any) continue;
        const int body = (int)((e.con_body >> (3 * cc)) & 7);
        const V3<T> p = mk<T>(mem(L::CON_PX + cc), mem(L::CON_PY + cc), mem(L::CON_PZ + cc));
        const T tx = mem(L::CON_TX + cc), ty = mem(L::CON_TY + cc), mu = mem(L::CON_MU + cc);
        T jn[TP::NV], jt1[TP::NV], jt2[TP::NV];
        contact_jac<T, TP>(k, body, p, tx, ty, jn, jt1, jt2);
        for (int s = 0; s < 4; s++) {
            if (!(jar[s] < T(0))) continue;
            const T D = Dr[s], f = -D * jar[s];
            const T sg = (s & 1) ? -mu : mu;
            c += T(0.5) * D * jar[s] * jar[s];
            T row[TP::NV];
            static_for<TP::NV>([&](auto ji) {
                constexpr int j = ji.value;
                row[j] = jn[j] + sg * (s < 2 ? jt1[j] : jt2[j]);
                fcon[j] += row[j] * f;
            });
            static_for<TP::NV>([&](auto ii) {
                constexpr int i = ii.value;
                const T di = D * row[i];
                static_for<TP::NV>([&](auto ji) {
                    constexpr int j = ji.value;
                    if constexpr (j <= i && TP::dof_anc(i, j)) H.a[i][j] += di * row[j];
                });
            });
        }
    }
    T g = T(0);
    static_for<TP::NV>([&](auto ii) {
        constexpr int i = ii.value;
        g += T(0.5) * (Ma[i] - smooth[i]) * (qacc[i] - qacc_smooth[i]);
        grad[i] = Ma[i] - smooth[i] - fcon[i];
        Mgrad[i] = grad[i];
    });
    gauss = g;
    cost = c + g;
    ltdl_factor<T, TP>(H);
    ltdl_solve<T, TP>(H, Mgrad);
}

// cost of a candidate acceleration (warmstart choice); uses ROW_JV as scratch for J a
template <typename T, typename TP>
DL_HD T candidate_cost(const Kin<T, TP>& k, const EfcInfo<TP>& e, const LaneMem<T>& mem, const TreeMat<T, TP>& M,
                       const T (&a)[TP::NV], const T (&smooth)[TP::NV], const T (&qacc_smooth)[TP::NV]) {
    using L = MemLayout<TP>;
    T Ma[TP::NV];
    treemat_mul<T, TP>(M, a, Ma);
    mul_J<T, TP, false>(k, e, mem, a, L::ROW_JV);
    T cost = T(0);
    for (int r = 0; r < e.nefc; r++) {
        const T x = mem(L::ROW_JV + r) + mem(L::ROW_JAREF + r);     // J a - aref (JAREF holds -aref here)
        if (x < T(0)) cost += T(0.5) * mem(L::ROW_D + r) * x * x;
    }
    static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; cost += T(0.5) * (Ma[i] - smooth[i]) * (a[i] - qacc_smooth[i]); });
    return cost;
}

// [3P] mj_forward: returns qacc; `warm` is qacc_warmstart (input only)
template <typename T, typename TP>
DL_HD void forward(const DevModel<T, TP>& m, const LaneMem<T>& mem, const T (&q)[TP::NV], const T (&v)[TP::NV],
                   const T (&ctrl)[TP::NU], const T (&warm)[TP::NV], T (&qacc)[TP::NV], EfcInfo<TP>& e, int& niter) {
    using L = MemLayout<TP>;
    Kin<T, TP> k;
    kinematics<T, TP>(m, q, k);
    TreeMat<T, TP> M;
    T smooth[TP::NV], qacc_smooth[TP::NV];
    {
        T bias[TP::NV];
        inertia_and_bias<T, TP>(m, k, v, M, bias);
        static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; smooth[j] = -m.damping[j] * v[j] - bias[j]; });
        static_for<TP::NU>([&](auto ai) {
            constexpr int a = ai.value;
            const T u = dl_clamp(ctrl[a], m.ctrl_lo[a], m.ctrl_hi[a]);
            smooth[TP::act_dof(a)] += m.gear[a] * dl_clamp(u, m.force_lo[a], m.force_hi[a]);
        });
    }
    {
        TreeMat<T, TP> LM;
        static_for<TP::NV>([&](auto ii) {
            constexpr int i = ii.value;
            qacc_smooth[i] = smooth[i];
            static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; if constexpr (j <= i && TP::dof_anc(i, j)) LM.a[i][j] = M.a[i][j]; });
        });
        ltdl_factor<T, TP>(LM);
        ltdl_solve<T, TP>(LM, qacc_smooth);
    }
    make_constraints<T, TP>(m, k, q, v, mem, e);
    niter = 0;
    if (e.nefc == 0) {
        static_for<TP::NV>([&](auto ii) { qacc[ii.value] = qacc_smooth[ii.value]; });
        return;
    }
    // warmstart: the cheaper of qacc_warmstart and qacc_smooth
    {
        const T cw = candidate_cost<T, TP>(k, e, mem, M, warm, smooth, qacc_smooth);
        const T cs = candidate_cost<T, TP>(k, e, mem, M, qacc_smooth, smooth, qacc_smooth);
        const bool use_warm = !(cw > cs);
        static_for<TP::NV>([&](auto ii) { qacc[ii.value] = use_warm ? warm[ii.value] : qacc_smooth[ii.value]; });
    }
    T Ma[TP::NV], grad[TP::NV], search[TP::NV], Mv[TP::NV];
    treemat_mul<T, TP>(M, qacc, Ma);
    mul_J<T, TP, true>(k, e, mem, qacc, L::ROW_JAREF);        // JAREF: -aref -> J qacc - aref
    T cost, gauss;
    solver_update<T, TP>(k, e, mem, M, qacc, Ma, smooth, qacc_smooth, cost, gauss, grad, search);
    static_for<TP::NV>([&](auto ii) { search[ii.value] = -search[ii.value]; });
    const T nvf = T(TP::NV);
    const T scale = T(1) / (m.meaninertia * nvf);
    int iter = 0;
    while (iter < m.iterations) {
        T s2 = T(0);
        static_for<TP::NV>([&](auto ii) { s2 += search[ii.value] * search[ii.value]; });
        const T snorm = dl_sqrt(s2);
        if (snorm < T(1e-15)) break;
        treemat_mul<T, TP>(M, search, Mv);
        mul_J<T, TP, false>(k, e, mem, search, L::ROW_JV);
        T g1 = T(0), g2 = T(0);
        static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; g1 += search[i] * (Ma[i] - smooth[i]); g2 += T(0.5) * search[i] * Mv[i]; });
        const T gtol = m.tolerance * m.ls_tolerance * snorm * m.meaninertia * nvf + m.ls_reltol * dl_abs(g1);
        const T alpha = linesearch<T, TP>(mem, e.nefc, gauss, g1, g2, gtol, m.ls_iterations);
        if (alpha == T(0)) break;
        static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; qacc[i] += alpha * search[i]; Ma[i] += alpha * Mv[i]; });
        for (int r = 0; r < e.nefc; r++) mem(L::ROW_JAREF + r) += alpha * mem(L::ROW_JV + r);
        const T oldcost = cost;
        solver_update<T, TP>(k, e, mem, M, qacc, Ma, smooth, qacc_smooth, cost, gauss, grad, search);
        T gn = T(0);
        static_for<TP::NV>([&](auto ii) { gn += grad[ii.value] * grad[ii.value]; search[ii.value] = -search[ii.value]; });
        const T improvement = scale * (oldcost - cost), gradient = scale * dl_sqrt(gn);
        iter++;
        if (improvement < m.tolerance || gradient < m.tolerance) break;
    }
    niter = iter;
}

// [3P] mj_step, RK4 (mj_RungeKutta N=4).  Returns true on divergence (mj_checkPos/Vel/Acc).
template <typename T, typename TP>
DL_HD bool mj_step_rk4(const DevModel<T, TP>& m, const LaneMem<T>& mem, T (&q)[TP::NV], T (&v)[TP::NV], const T (&ctrl)[TP::NU], T (&warm)[TP::NV]) {
    bool bad = false;
    static_for<TP::NV>([&](auto ii) { bad = bad || dl_bad(q[ii.value]) || dl_bad(v[ii.value]); });
    if (bad) return true;
    const T h = m.timestep;
    T q0[TP::NV], v0[TP::NV], qs[TP::NV], vs[TP::NV], dq[TP::NV], dv[TP::NV], acc[TP::NV];
    static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; q0[i] = q[i]; v0[i] = v[i]; qs[i] = q[i]; vs[i] = v[i]; dq[i] = T(0); dv[i] = T(0); });
    EfcInfo<TP> e;
    int niter;
#pragma unroll 1
    for (int stage = 0; stage < 4; stage++) {
        forward<T, TP>(m, mem, qs, vs, ctrl, warm, acc, e, niter);
        static_for<TP::NV>([&](auto ii) { warm[ii.value] = acc[ii.value]; });
        if (stage == 0) {
            bool b2 = false;
            static_for<TP::NV>([&](auto ii) { b2 = b2 || dl_bad(acc[ii.value]); });
            if (b2) { bad = true; break; }
        }
        // classic tableau: stage weights 1/6 1/3 1/3 1/6, next-stage step 1/2 1/2 1
        const T wgt = (stage == 0 || stage == 3) ? T(1) / T(6) : T(1) / T(3);
        const T a = stage == 2 ? T(1) : T(0.5);
        static_for<TP::NV>([&](auto ii) {
            constexpr int i = ii.value;
            dq[i] += wgt * vs[i]; dv[i] += wgt * acc[i];
            const T vstage = vs[i];
            qs[i] = q0[i] + h * a * vstage;
            vs[i] = v0[i] + h * a * acc[i];
        });
    }
    if (bad) return true;
    static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; q[i] = q0[i] + h * dq[i]; v[i] = v0[i] + h * dv[i]; });
    return false;
}

}  // namespace dl
