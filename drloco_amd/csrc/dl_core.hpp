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

// out-of-line call boundary for the forward evaluation (see forward_call)
#if defined(DL_INLINE_FORWARD)
#define DL_NOINLINE DL_HD
#elif defined(__HIPCC__)
#define DL_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define DL_NOINLINE __attribute__((noinline))
#endif

// lane memory is LDS on the device: keep the address space in the pointer type so that ds_*
// instructions (not flat_*) are used even across the out-of-line call
#if defined(__HIP_DEVICE_COMPILE__)
#define DL_LDS __attribute__((address_space(3)))
#else
#define DL_LDS
#endif

// model parameters live in a device buffer that is immutable while kernels run: address it
// through the constant address space so that uniform parameter reads become scalar loads
#if defined(__HIP_DEVICE_COMPILE__)
#define DL_CONST __attribute__((address_space(4)))
#else
#define DL_CONST
#endif

// hide a value from the optimiser (stops it from peeling the solver's phase loop into one copy of
// the body per phase, which multiplies the code size beyond the instruction cache)
#if defined(__HIP_DEVICE_COMPILE__)
#define DL_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define DL_OPAQUE(x) asm volatile("" : "+r"(x))
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
    static constexpr int ENV_KIND = 0;             // DL_ENV_STRAIGHT
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


// topology of walker_165cm_65kg.xml (/root/reference/drloco/mujoco/xml/walker_165cm_65kg.xml:15-95):
// pelvis (3 slides + 3 hinges) -> torso (3 lumbar hinges), and two legs of hip (3 hinges), knee, ankle
struct TopoWalker165 {
    static constexpr int ENV_KIND = 1;             // DL_ENV_LOCO3D
    static constexpr int NB = 9, NV = 19, NU = 13, NG = 8, NS = 8, NLIM = 13;
    static constexpr int MAXCON = 24;              // 4 boxes x 4 + 4 capsules x 2
    static constexpr int MAXROW = 112;             // NLIM + 4 * MAXCON = 109, padded to a multiple of 4 (rows are processed four at a time)
    static constexpr int OBS = 47;                 // 8 joint-phase features + 2 desired velocities + 18 + 19
    static constexpr int body_parent_[NB] = {0, 0, 1, 1, 3, 4, 1, 6, 7};
    static constexpr int dof_body_[NV] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 5, 6, 6, 6, 7, 8};
    static constexpr int dof_type_[NV] = {0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    static constexpr int dof_axis_[NV] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 0, 2, 1, 1, 1, 0, 2, 1, 1};
    static constexpr int dof_sign_[NV] = {1, -1, 1, 1, -1, 1, 1, -1, 1, -1, 1, -1, -1, 1, -1, -1, -1, -1, -1};
    static constexpr int dof_parent_[NV] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 5, 9, 10, 11, 12, 5, 14, 15, 16, 17};
    static constexpr int dof_limited_[NV] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    static constexpr int geom_body_[NG] = {1, 2, 3, 4, 5, 6, 7, 8};
    static constexpr int geom_type_[NG] = {1, 1, 0, 0, 1, 0, 0, 1};
    static constexpr int site_body_[NS] = {5, 5, 5, 5, 8, 8, 8, 8};
    // motors: lumbar_extension, lumbar_bending, lumbar_rotation, then the legs (xml:81-95)
    static constexpr int act_dof_[NU] = {7, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18};
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
    static constexpr bool dof_anc(int i, int j) {
        while (i >= 0) { if (i == j) return true; i = dof_parent_[i]; }
        return false;
    }
    static constexpr bool body_anc(int b, int j) {
        while (b > 0) { if (dof_body_[j] == b) return true; b = body_parent_[b]; }
        return false;
    }
    static constexpr int body_last_dof(int b) {
        while (b > 0) {
            int last = -1;
            for (int j = 0; j < NV; j++) if (dof_body_[j] == b) last = j;
            if (last >= 0) return last;
            b = body_parent_[b];
        }
        return -1;
    }
    static constexpr uint32_t body_mask(int b) {
        uint32_t m = 0;
        for (int j = 0; j < NV; j++) if (body_anc(b, j)) m |= 1u << j;
        return m;
    }
    // joints whose (angle, velocity) phase plot replaces the phase variable (mimic_walker_165cm_65kg.py:40-43)
    static constexpr int phase_joint_[4] = {9, 12, 14, 17};
    static constexpr int phase_joint(int k) { return phase_joint_[k]; }
    // no policy mirroring for this walker (Loco3dReferenceTrajectories has no is_step_left)
    static constexpr int obs_perm(int k) { return k; }
    static constexpr int obs_neg(int) { return 0; }
    static constexpr int act_perm(int k) { return k; }
    static constexpr int act_neg(int) { return 0; }
};

// ------------------------------------------------------------------------------------------
// numeric model parameters (uniform across lanes: kernel argument -> SGPRs / scalar loads)
template <typename T, typename TP> struct DevModel {
    T body_pos[TP::NB][3], body_mass[TP::NB], body_ipos[TP::NB][3], body_inertia[TP::NB][3];
    T qpos0[TP::NV], range[TP::NV][2], damping[TP::NV], armature[TP::NV], dof_invw[TP::NV];
    T geom_pos[TP::NG][3], geom_mat[TP::NG][9], geom_size[TP::NG][3], geom_mu[TP::NG], body_invw[TP::NB];
    T site_pos[TP::NS][3];
    T ctrl_lo[TP::NU], ctrl_hi[TP::NU], force_lo[TP::NU], force_hi[TP::NU], gear[TP::NU];
    T timestep, gravity_z, solK, solB, solimp[5], meaninertia, tolerance, ls_tolerance, ls_reltol, tol_rel;
    int32_t iterations, ls_iterations, frame_skip;
};

// environment constants (drloco/config/hypers.py, config.py) + reference table view
template <typename T> struct DevCfg {
    T rew_w[3], rew_scale, alive_bonus, com_z_min, inv_ctrl_freq;
    int32_t ep_dur_max, mirror_policy, env_index_base;
    int32_t intended;          // dl_config.intended_semantics (DL_INTENDED_*): 0 = the reference's behaviour incl. its quirks Q1-Q4
    uint64_t seed;
    int32_t n_steps, total_len, stride, n_rows;
    const T* table;            // sample-major [total_len][2*NV]: the 2*NV reference values of one mocap sample are contiguous
    const int32_t* step_off;   // [n_steps+1]
    const int32_t* step_is_left;
    const T* step_vel;
    const double* pref;        // loco3d: prefix sums [2][total_len+1] of the reference pelvis x / z velocity rows
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
DL_HD float dl_atan2(float y, float x) { return atan2f(y, x); }
DL_HD double dl_atan2(double y, double x) { return atan2(y, x); }
DL_HD float dl_pow(float x, float y) { return powf(x, y); }
DL_HD double dl_pow(double x, double y) { return pow(x, y); }
// float sincos: Cody-Waite reduction by pi/2 (3 constants) + minimax polynomials on [-pi/4, pi/4].
// ~35 instructions instead of the ~260 of the inlined libm sinf+cosf pair; |error| < 2.5e-7 for
// |x| < 1e4 (joint angles are O(1); the root yaw of a tumbling walker stays far below that).
DL_HD void dl_sincos(float x, float& s, float& c) {
    const float kf = rintf(x * 0.63661977236758134f);        // 2/pi
    const int k = (int)kf;
    float r = fmaf(kf, -1.5703125f, x);                      // pi/2 split: hi
    r = fmaf(kf, -4.837512969970703125e-4f, r);              // mid
    r = fmaf(kf, -7.54978995489188e-8f, r);                  // lo
    const float r2 = r * r;
    float sp = fmaf(r2, 2.6083159809786593541503e-06f, -0.0001981069071916863322258f);
    sp = fmaf(sp, r2, 0.00833307858556509017944336f);
    sp = fmaf(sp, r2, -0.166666597127914428710938f);
    const float sr = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.44331571e-5f, -1.38873163e-3f);
    cp = fmaf(cp, r2, 4.16666418e-2f);
    const float cr = fmaf(cp * r2, r2, fmaf(r2, -0.5f, 1.0f));
    const bool swap = k & 1;
    const float ss = swap ? cr : sr, cc = swap ? sr : cr;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}
DL_HD void dl_sincos(double x, double& s, double& c) { s = sin(x); c = cos(x); }
template <typename T> DL_HD T dl_max(T a, T b) { return a > b ? a : b; }
template <typename T> DL_HD T dl_min(T a, T b) { return a < b ? a : b; }
template <typename T> DL_HD T dl_clamp(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
template <typename T> DL_HD bool dl_bad(T x) { return !(x == x) || x > T(1e10) || x < T(-1e10); }
// A float32 output of a step whose state left float32's range in its LAST substep (the state is caught by the next mj_checkPos / mj_checkVel, one
// control step later, exactly as in the reference -- but there, in float64, the value that becomes an observation is still a finite number):
// saturate at the largest finite float instead of handing inf / NaN to VecNormalize, whose moments would turn NaN for good
DL_HD float dl_sat_out(float x) { return x == x ? (x > 3.0e38f ? 3.0e38f : (x < -3.0e38f ? -3.0e38f : x)) : 3.0e38f; }

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
    DL_LDS T* base; int stride;
    DL_HD DL_LDS T& operator()(int idx) const { return base[idx * stride]; }
};
// per-lane view of a global (HBM) workspace laid out [word][N]
template <typename T> struct GlobalMem {
    T* base; int stride;
    DL_HD T& operator()(int idx) const { return base[(size_t)idx * stride]; }
};
// slot layout inside LaneMem
template <typename TP> struct MemLayout {
    static constexpr int ROW_D = 0, ROW_JAREF = TP::MAXROW, ROW_JV = 2 * TP::MAXROW, ROW_TMP = 3 * TP::MAXROW;
    static constexpr int CON_PX = 4 * TP::MAXROW;            // contact point relative to the root origin
    static constexpr int CON_PY = CON_PX + TP::MAXCON, CON_PZ = CON_PY + TP::MAXCON;
    static constexpr int CON_TX = CON_PZ + TP::MAXCON, CON_TY = CON_TX + TP::MAXCON;   // first tangent (unit, in the floor plane)
    static constexpr int CON_MU = CON_TY + TP::MAXCON, CON_DIST = CON_MU + TP::MAXCON;
    static constexpr int CON_BODY = CON_DIST + TP::MAXCON;    // body id of the contact (small integer stored as T)
    static constexpr int LIM_CODE = CON_BODY + TP::MAXCON;    // per limit row: dof | 32 * upper-side flag
    static constexpr int MAT = LIM_CODE + TP::NLIM;           // mass matrix, tree pattern, lower
    // position of M[i][j] (j ancestor-or-self of i) in the packed pattern
    static constexpr int mat_index(int i, int j) {
        int n = 0;
        for (int a = 0; a < TP::NV; a++)
            for (int b = 0; b <= a; b++) {
                if (!TP::dof_anc(a, b)) continue;
                if (a == i && b == j) return n;
                n++;
            }
        return -1;
    }
    static constexpr int mat_count() {
        int n = 0;
        for (int a = 0; a < TP::NV; a++)
            for (int b = 0; b <= a; b++) if (TP::dof_anc(a, b)) n++;
        return n;
    }
    template <int I, int J> static constexpr int MI = mat_index(I, J);   // forces compile-time evaluation
    // arguments / result of forward_call (kept in lane memory so that nothing crosses the call in
    // private scratch memory)
    static constexpr int IO_Q = MAT + mat_count(), IO_V = IO_Q + TP::NV, IO_WARM = IO_V + TP::NV, IO_QACC = IO_WARM + TP::NV;
    static constexpr int IO_CTRL = IO_QACC + TP::NV;
    static constexpr int TOTAL = IO_CTRL + TP::NU;            // elements of T per lane
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
DL_HD void kinematics(const DL_CONST DevModel<T, TP>& m, const T (&q)[TP::NV], Kin<T, TP>& k) {
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

template <typename T, typename TP> DL_HD V3<T> body_point(const Kin<T, TP>& k, int b, const DL_CONST T* local) {
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
// [3P] mj_crb + mj_rne(bias) in ONE depth-first sweep over the body tree.
// subtree<b>() receives the twist / velocity-product acceleration of b's parent, walks b's dofs,
// forms b's spatial inertia and inertial wrench, recurses into the children (accumulating their
// composite inertia and wrench -- same reference point, so plain sums), and then emits the rows of
// the mass matrix (into lane memory) and the bias forces of b's dofs.  Depth-first order keeps only
// one root-to-leaf chain of temporaries alive at a time (register pressure), unlike the textbook
// "all bodies down, all bodies up" arrangement.
template <typename T> struct SubtreeOut { SI<T> Ic; SV<T> W; };

template <typename T, typename TP, int b>
DL_HD SubtreeOut<T> crb_rne_subtree(const DL_CONST DevModel<T, TP>& m, const Kin<T, TP>& k, const T (&v)[TP::NV], const LaneMem<T>& mem,
                                    SV<T> vel, SV<T> acc, T (&bias)[TP::NV]) {
    using L = MemLayout<TP>;
    // chain through the dofs of body b (in dof order)
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value;
        if constexpr (TP::dof_body(j) == b) {
            const SV<T> S = dof_S<T, TP, j>(k);
            const SV<T> vJ = {v[j] * S.w, v[j] * S.v};
            acc = {acc.w + cross(vel.w, vJ.w), acc.v + cross(vel.w, vJ.v) + cross(vel.v, vJ.w)};
            vel = vel + vJ;
        }
    });
    // spatial inertia of b about the reference point (root origin), world orientation
    SubtreeOut<T> out;
    {
        const V3<T> c = body_point<T, TP>(k, b, m.body_ipos[b]);
        const T mass = m.body_mass[b];
        const T i0 = m.body_inertia[b][0], i1 = m.body_inertia[b][1], i2 = m.body_inertia[b][2];
        const V3<T> X = k.RX[b], Y = k.RY[b], Z = k.RZ[b];
        const T cc = dot(c, c);
        SI<T>& s = out.Ic;
        s.m = mass; s.h = mass * c;
        s.I.xx = i0 * X.x * X.x + i1 * Y.x * Y.x + i2 * Z.x * Z.x + mass * (cc - c.x * c.x);
        s.I.yy = i0 * X.y * X.y + i1 * Y.y * Y.y + i2 * Z.y * Z.y + mass * (cc - c.y * c.y);
        s.I.zz = i0 * X.z * X.z + i1 * Y.z * Y.z + i2 * Z.z * Z.z + mass * (cc - c.z * c.z);
        s.I.xy = i0 * X.x * X.y + i1 * Y.x * Y.y + i2 * Z.x * Z.y - mass * c.x * c.y;
        s.I.xz = i0 * X.x * X.z + i1 * Y.x * Y.z + i2 * Z.x * Z.z - mass * c.x * c.z;
        s.I.yz = i0 * X.y * X.z + i1 * Y.y * Y.z + i2 * Z.y * Z.z - mass * c.y * c.z;
        const SV<T> Iv = si_mul(s, vel);
        const SV<T> Ia = si_mul(s, acc);
        out.W = {Ia.w + cross(vel.w, Iv.w) + cross(vel.v, Iv.v), Ia.v + cross(vel.w, Iv.v)};
    }
    // children
    static_for<TP::NB>([&](auto ci) {
        constexpr int c = ci.value;
        if constexpr (c > b && TP::body_parent(c) == b) {
            const SubtreeOut<T> ch = crb_rne_subtree<T, TP, c>(m, k, v, mem, vel, acc, bias);
            si_add(out.Ic, ch.Ic);
            out.W = out.W + ch.W;
        }
    });
    // rows of M and bias entries of b's dofs
    static_for<TP::NV>([&](auto ii) {
        constexpr int i = ii.value;
        if constexpr (TP::dof_body(i) == b) {
            const SV<T> S = dof_S<T, TP, i>(k);
            bias[i] = sdot(S, out.W);
            const SV<T> f = si_mul(out.Ic, S);
            static_for<TP::NV>([&](auto ji) {
                constexpr int j = ji.value;
                if constexpr (j <= i && TP::dof_anc(i, j)) {
                    const SV<T> Sj = dof_S<T, TP, j>(k);
                    T mij = sdot(Sj, f);
                    if constexpr (i == j) mij += m.armature[i];
                    mem(L::MAT + L::template MI<i, j>) = mij;
                }
            });
        }
    });
    return out;
}

template <typename T, typename TP>
DL_HD void inertia_and_bias(const DL_CONST DevModel<T, TP>& m, const Kin<T, TP>& k, const T (&v)[TP::NV], const LaneMem<T>& mem, T (&bias)[TP::NV]) {
    const SV<T> vel0 = {mk<T>(0, 0, 0), mk<T>(0, 0, 0)};
    const SV<T> acc0 = {mk<T>(0, 0, 0), mk<T>(0, 0, -m.gravity_z)};      // gravity as a base acceleration of -g
    static_for<TP::NB>([&](auto bi) {
        constexpr int b = bi.value;
        if constexpr (b > 0 && TP::body_parent(b) == 0) (void)crb_rne_subtree<T, TP, b>(m, k, v, mem, vel0, acc0, bias);
    });
}

// ------------------------------------------------------------------------------------------
// constraint bookkeeping of one evaluation
template <typename TP> struct EfcInfo {
    int nlim, ncon, nefc;    // limit codes and contact bodies live in lane memory (LIM_CODE / CON_BODY)
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
        constexpr int bb = bi.value + 1, ld = TP::body_last_dof(bb);
        if (b == bb) r = vel[ld];
    });
    return r;
}

// [3P] solimp sigmoid (getimpedance); the kernels support the powers 1 and 2 (MuJoCo's default is 2;
// dl_create rejects anything else) so that no pow() expansion is inlined
template <typename T> DL_HD T impedance(const DL_CONST T* si, T pos) {
    const T x = dl_abs(pos) / si[2];
    T y;
    if (si[4] == T(1)) y = x;
    else y = (x <= si[3]) ? x * x / si[3] : T(1) - (T(1) - x) * (T(1) - x) / (T(1) - si[3]);
    const T imp = si[0] + y * (si[1] - si[0]);
    return x >= T(1) ? si[1] : (x <= T(0) ? si[0] : imp);
}

// [3P] mj_collision (plane vs capsule / box) + the position-dependent part of mj_makeConstraint /
// mj_makeImpedance: contacts and, per row, D and K*imp*pos (stored in ROW_JAREF; the velocity
// part B*(J v) of -aref is added by the solver's first pass).
template <typename T, typename TP>
DL_HD void make_constraints(const DL_CONST DevModel<T, TP>& m, const Kin<T, TP>& k, const T (&q)[TP::NV],
                            const LaneMem<T>& mem, EfcInfo<TP>& e) {
    using L = MemLayout<TP>;
    e.nlim = 0; e.ncon = 0;
    // joint limits: rows +-e_j (detected per dof; finished in the row loop below)
    static_for<TP::NV>([&](auto ji) {
        constexpr int j = ji.value;
        if constexpr (TP::dof_limited(j)) {
            const T dlo = q[j] - m.range[j][0], dhi = m.range[j][1] - q[j];
            const bool lo = dlo < T(0), hi = dhi < T(0);
            if (lo || hi) {
                const int r = e.nlim;
                mem(L::ROW_JAREF + r) = lo ? dlo : dhi;
                mem(L::ROW_D + r) = m.dof_invw[j];
                mem(L::LIM_CODE + r) = T(j | (lo ? 0 : 32));
                e.nlim = r + 1;
            }
        }
    });
    for (int r = 0; r < e.nlim; r++) {
        const T dist = mem(L::ROW_JAREF + r);
        const T imp = impedance(m.solimp, dist);
        const T R = dl_max(T(1e-15), (T(1) - imp) * mem(L::ROW_D + r) / imp);
        mem(L::ROW_D + r) = T(1) / R;
        mem(L::ROW_JAREF + r) = m.solK * imp * dist;
    }
    // contacts
    auto add_contact = [&](int body, T mu, V3<T> p, T dist, T tx, T ty) {
        const int c = e.ncon;
        if (c >= TP::MAXCON) return;
        mem(L::CON_PX + c) = p.x; mem(L::CON_PY + c) = p.y; mem(L::CON_PZ + c) = p.z;
        mem(L::CON_TX + c) = tx; mem(L::CON_TY + c) = ty; mem(L::CON_MU + c) = mu; mem(L::CON_DIST + c) = dist;
        mem(L::CON_BODY + c) = T(body);
        e.ncon = c + 1;
    };
    static_for<TP::NG>([&](auto gi) {
        constexpr int g = gi.value, b = TP::geom_body(g);
        const V3<T> gp = body_point<T, TP>(k, b, m.geom_pos[g]);
        const DL_CONST T* gm = m.geom_mat[g];
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
                if (dist < T(0)) add_contact(b, m.geom_mu[g], mk<T>(c.x, c.y, c.z - (rad + T(0.5) * dist)), dist, tx, ty);
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
                    add_contact(b, m.geom_mu[g], mk<T>(gp.x + corner.x, gp.y + corner.y, gp.z + corner.z - T(0.5) * dist), dist, T(0), T(1));
                    cnt++;
                }
            });
        }
    });
    // contact rows: 4 pyramid edges n +- mu t1, n +- mu t2; t1 = (tx, ty, 0), t2 = n x t1 = (-ty, tx, 0)
    for (int c = 0; c < e.ncon; c++) {
        const int body = (int)mem(L::CON_BODY + c);
        const T mu = mem(L::CON_MU + c), dist = mem(L::CON_DIST + c);
        T invw = T(0);
        static_for<TP::NB - 1>([&](auto bi) { if (body == bi.value + 1) invw = m.body_invw[bi.value + 1]; });
        const T imp = impedance(m.solimp, dist);
        const T diag = invw * (T(1) + mu * mu);
        const T R = T(2) * mu * mu * dl_max(T(1e-15), (T(1) - imp) * diag / imp);
        const T D = T(1) / R, kd = m.solK * imp * dist;
        const int r = e.nlim + 4 * c;
        mem(L::ROW_D + r) = D; mem(L::ROW_D + r + 1) = D; mem(L::ROW_D + r + 2) = D; mem(L::ROW_D + r + 3) = D;
        mem(L::ROW_JAREF + r) = kd; mem(L::ROW_JAREF + r + 1) = kd; mem(L::ROW_JAREF + r + 2) = kd; mem(L::ROW_JAREF + r + 3) = kd;
    }
    e.nefc = e.nlim + 4 * e.ncon;
}

// ROW_JV[r] = (J x)[r] for all rows, matrix free: body twists under x, then point velocities
template <typename T, typename TP>
DL_HD void mul_J(const Kin<T, TP>& k, const EfcInfo<TP>& e, const LaneMem<T>& mem, const T (&x)[TP::NV]) {
    using L = MemLayout<TP>;
    for (int r = 0; r < e.nlim; r++) {
        const int code = (int)mem(L::LIM_CODE + r), j = code & 31;
        T xj = T(0);
        static_for<TP::NV>([&](auto ji) { if constexpr (TP::dof_limited(ji.value)) if (j == ji.value) xj = x[ji.value]; });
        mem(L::ROW_JV + r) = (code & 32) ? -xj : xj;
    }
    if (e.ncon == 0) return;
    SV<T> vel[TP::NV];
    body_twists<T, TP>(k, x, vel);
    for (int c = 0; c < e.ncon; c++) {
        const int body = (int)mem(L::CON_BODY + c);
        const V3<T> p = mk<T>(mem(L::CON_PX + c), mem(L::CON_PY + c), mem(L::CON_PZ + c));
        const T tx = mem(L::CON_TX + c), ty = mem(L::CON_TY + c), mu = mem(L::CON_MU + c);
        const SV<T> tw = twist_of_body<T, TP>(vel, body);
        const V3<T> pv = tw.v + cross(tw.w, p);
        const T vn = pv.z, vt1 = mu * (tx * pv.x + ty * pv.y), vt2 = mu * (-ty * pv.x + tx * pv.y);
        const int r = L::ROW_JV + e.nlim + 4 * c;
        mem(r) = vn + vt1; mem(r + 1) = vn - vt1; mem(r + 2) = vn + vt2; mem(r + 3) = vn - vt2;
    }
}

// Jacobian of contact c in the contact frame: jn, jt1, jt2 (dense over dofs; zero where the dof
// does not move the body)
template <typename T, typename TP>
DL_HD void contact_jac(const Kin<T, TP>& k, int body, V3<T> p, T tx, T ty, T (&jn)[TP::NV], T (&jt1)[TP::NV], T (&jt2)[TP::NV]) {
    uint32_t mask = 0;
    static_for<TP::NB - 1>([&](auto bi) { constexpr uint32_t bm = TP::body_mask(bi.value + 1); if (body == bi.value + 1) mask = bm; });
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
// exact line search on the piecewise quadratic along the search direction (MuJoCo's scheme:
// Newton steps from one side until the derivative changes sign, then a bracketed search over
// {Newton from both ends, midpoint}).  Written as a state machine around ONE evaluation site so
// that the row loop is emitted once.
template <typename T> struct LsPoint { T alpha, cost, d1, d2; };

template <typename T> DL_HD bool ls_update_bracket(LsPoint<T>& p, const LsPoint<T> (&cand)[3]) {
    bool flag = false;
    for (int i = 0; i < 3; i++) {
        if (p.d1 < T(0) && cand[i].d1 < T(0) && p.d1 < cand[i].d1) { p = cand[i]; flag = true; }
        else if (p.d1 > T(0) && cand[i].d1 > T(0) && p.d1 > cand[i].d1) { p = cand[i]; flag = true; }
    }
    return flag;
}

template <typename T, typename TP>
DL_HD T linesearch(const LaneMem<T>& mem, int nefc, T g0, T g1, T g2, T gtol, int maxit) {
    using L = MemLayout<TP>;
    LsPoint<T> p0 = {T(0), T(0), T(0), T(1)}, p1 = p0, p2 = p0, pmid = p0, p1next = p0, p2next = p0;
    LsPoint<T> cand[3] = {p0, p0, p0};
    T dir = T(1), alpha = T(0);
    bool p2update = false, b1 = false, b2 = false;
    int it = 0, state = 0;
    for (;;) {
        // ---- the single evaluation site
        LsPoint<T> pe;
        {
            T cost = g0 + alpha * g1 + alpha * alpha * g2, d1 = g1 + T(2) * alpha * g2, d2 = T(2) * g2;
            // four rows per trip: the 12 LDS reads of a chunk are independent, so their latency overlaps
            for (int r = 0; r < nefc; r += 4) {
                T jv[4], ja[4], Dr[4];
                static_for<4>([&](auto ki) { constexpr int kk = ki.value; jv[kk] = mem(L::ROW_JV + r + kk); ja[kk] = mem(L::ROW_JAREF + r + kk); Dr[kk] = mem(L::ROW_D + r + kk); });
                static_for<4>([&](auto ki) {
                    constexpr int kk = ki.value;
                    const T x = ja[kk] + alpha * jv[kk];
                    if (r + kk < nefc && x < T(0)) { cost += T(0.5) * Dr[kk] * x * x; d1 += Dr[kk] * x * jv[kk]; d2 += Dr[kk] * jv[kk] * jv[kk]; }
                });
            }
            pe = {alpha, cost, d1, d2};
        }
        bool end_onesided = false, end_iter = false;
        if (state == 0) {                       // p0 = f(0)
            p0 = pe; alpha = -p0.d1 / p0.d2; state = 1;
            continue;
        } else if (state == 1) {                // p1 = Newton point from 0
            p1 = pe;
            if (p0.cost < p1.cost) p1 = p0;
            if (dl_abs(p1.d1) < gtol) return p1.alpha;
            dir = p1.d1 < T(0) ? T(1) : T(-1);
            p2 = p1;
            if (p1.d1 * dir <= -gtol && it < maxit) { p2 = p1; p2update = true; alpha = p1.alpha - p1.d1 / p1.d2; state = 2; continue; }
            end_onesided = true;
        } else if (state == 2) {                // one-sided Newton iterations
            p1 = pe; it++;
            if (dl_abs(p1.d1) < gtol) return p1.alpha;
            if (p1.d1 * dir <= -gtol && it < maxit) { p2 = p1; alpha = p1.alpha - p1.d1 / p1.d2; continue; }
            end_onesided = true;
        } else if (state == 3) {                // p1next after the bracket was found
            p1next = pe;
            end_iter = true;
        } else if (state == 4) {                // midpoint of the bracket
            pmid = pe; it++;
            cand[0] = p1next; cand[1] = p2next; cand[2] = pmid;
            int best = -1;
            T bestcost = T(0), bestalpha = T(0);
            for (int i = 0; i < 3; i++)
                if (dl_abs(cand[i].d1) < gtol && (best < 0 || cand[i].cost < bestcost)) { best = i; bestcost = cand[i].cost; bestalpha = cand[i].alpha; }
            if (best >= 0) return bestalpha;
            b1 = ls_update_bracket(p1, cand);
            if (b1) { alpha = p1.alpha - p1.d1 / p1.d2; state = 5; continue; }
            b2 = ls_update_bracket(p2, cand);
            if (b2) { alpha = p2.alpha - p2.d1 / p2.d2; state = 6; continue; }
            end_iter = true;
        } else if (state == 5) {                // Newton point from the updated p1
            p1next = pe;
            b2 = ls_update_bracket(p2, cand);
            if (b2) { alpha = p2.alpha - p2.d1 / p2.d2; state = 6; continue; }
            end_iter = true;
        } else {                                // state 6: Newton point from the updated p2
            p2next = pe;
            end_iter = true;
        }
        if (end_onesided) {
            if (it >= maxit || !p2update) return p1.alpha;
            p2next = p1;
            alpha = p1.alpha - p1.d1 / p1.d2;
            state = 3;
            continue;
        }
        if (end_iter) {
            if (state != 3 && !b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : T(0);
            if (it >= maxit) break;
            b1 = false; b2 = false;
            alpha = T(0.5) * (p1.alpha + p2.alpha);
            state = 4;
        }
    }
    if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
    if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
    return T(0);
}

// ------------------------------------------------------------------------------------------
// [3P] mj_forward: returns qacc; `warm` is qacc_warmstart (input only).
//
// After kinematics / inertia / bias / collision the rest is ONE loop whose body is emitted once:
//   phase -1  x = qvel      : J x completes -aref;  H = M -> factor -> qacc_smooth = M^-1 qfrc_smooth
//   phase  0  x = qacc_smooth: candidate cost
//   phase  1  x = warmstart  : candidate cost, choose the cheaper start, initial Hessian
//   phase >=2 x = search     : exact line search, move, incremental Hessian update
// every phase >= 1 ends with cost / gradient / factorisation of H / Newton direction.
template <typename T, typename TP>
DL_HD void forward(const DL_CONST DevModel<T, TP>& m, const LaneMem<T>& mem, const T (&q)[TP::NV], const T (&v)[TP::NV],
                   const T (&ctrl)[TP::NU], const T (&warm)[TP::NV], T (&qacc)[TP::NV], EfcInfo<TP>& e, int& niter) {
    using L = MemLayout<TP>;
    Kin<T, TP> k;
    kinematics<T, TP>(m, q, k);
    T smooth[TP::NV], qacc_smooth[TP::NV];
    {
        T bias[TP::NV];
        inertia_and_bias<T, TP>(m, k, v, mem, bias);      // M goes to lane memory
        static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; smooth[j] = -m.damping[j] * v[j] - bias[j]; });
        static_for<TP::NU>([&](auto ai) {
            constexpr int a = ai.value;
            const T u = dl_clamp(ctrl[a], m.ctrl_lo[a], m.ctrl_hi[a]);
            smooth[TP::act_dof(a)] += m.gear[a] * dl_clamp(u, m.force_lo[a], m.force_hi[a]);
        });
    }
    make_constraints<T, TP>(m, k, q, mem, e);
    const int nefc = e.nefc;
    niter = 0;

    TreeMat<T, TP> H;                    // M + sum_active D row row^T, kept in registers across the Newton iterations
    uint64_t act_lo = 0, act_hi = 0;     // per-row "quadratic" state
    T x[TP::NV], Mx[TP::NV], Ma[TP::NV], rhs[TP::NV];
    T cost = T(0), gauss = T(0), cost_s = T(0);
    const T nvf = T(TP::NV);
    const T scale = T(1) / (m.meaninertia * nvf);
    int phase = -1, iter = 0;
    static_for<TP::NV>([&](auto ii) { x[ii.value] = v[ii.value]; Mx[ii.value] = T(0); Ma[ii.value] = T(0); rhs[ii.value] = T(0); qacc[ii.value] = T(0); });
    for (;;) {
        // ---- x -> M x (phases >= 0) and J x (whenever there are rows)
        if (phase == 0) {
            static_for<TP::NV>([&](auto ir) { Mx[ir.value] = smooth[ir.value]; });      // M qacc_smooth = qfrc_smooth
        } else if (phase > 0) {
            static_for<TP::NV>([&](auto ir) { Mx[ir.value] = mem(L::MAT + L::template MI<ir.value, ir.value>) * x[ir.value]; });
            static_for<TP::NV>([&](auto ir) {
                constexpr int i = ir.value;
                static_for<TP::NV>([&](auto jr) {
                    constexpr int j = jr.value;
                    if constexpr (j < i && TP::dof_anc(i, j)) { const T mij = mem(L::MAT + L::template MI<i, j>); Mx[i] += mij * x[j]; Mx[j] += mij * x[i]; }
                });
            });
        }
        if (nefc > 0) mul_J<T, TP>(k, e, mem, x);
        T alpha = T(0);
        bool stop = false;
        if (phase == -1) {
            for (int r = 0; r < nefc; r += 4) {                                                        // -aref complete
                T a[4], b[4];
                static_for<4>([&](auto ki) { a[ki.value] = mem(L::ROW_JAREF + r + ki.value); b[ki.value] = mem(L::ROW_JV + r + ki.value); });
                static_for<4>([&](auto ki) { mem(L::ROW_JAREF + r + ki.value) = a[ki.value] + m.solB * b[ki.value]; });
            }
        } else if (phase <= 1) {
            // cost of the candidate start x
            T c = T(0);
            for (int r = 0; r < nefc; r += 4) {
                T jv[4], ja[4], Dr[4];
                static_for<4>([&](auto ki) { constexpr int kk = ki.value; jv[kk] = mem(L::ROW_JV + r + kk); ja[kk] = mem(L::ROW_JAREF + r + kk); Dr[kk] = mem(L::ROW_D + r + kk); });
                static_for<4>([&](auto ki) {
                    constexpr int kk = ki.value;
                    const T jar = jv[kk] + ja[kk];
                    if (r + kk < nefc && jar < T(0)) c += T(0.5) * Dr[kk] * jar * jar;
                });
            }
            static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; c += T(0.5) * (Mx[i] - smooth[i]) * (x[i] - qacc_smooth[i]); });
            if (phase == 0) {
                cost_s = c;
                for (int r = 0; r < nefc; r += 4) {
                    T a[4];
                    static_for<4>([&](auto ki) { a[ki.value] = mem(L::ROW_JV + r + ki.value); });
                    static_for<4>([&](auto ki) { mem(L::ROW_TMP + r + ki.value) = a[ki.value]; });
                }
            } else {
                const bool use_warm = !(c > cost_s);
                // M qacc_smooth = qfrc_smooth by construction
                static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; qacc[i] = use_warm ? warm[i] : qacc_smooth[i]; Ma[i] = use_warm ? Mx[i] : smooth[i]; });
                for (int r = 0; r < nefc; r += 4) {
                    T a[4], b[4], t[4];
                    static_for<4>([&](auto ki) { constexpr int kk = ki.value; a[kk] = mem(L::ROW_JAREF + r + kk); b[kk] = mem(L::ROW_JV + r + kk); t[kk] = mem(L::ROW_TMP + r + kk); });
                    static_for<4>([&](auto ki) { constexpr int kk = ki.value; mem(L::ROW_JAREF + r + kk) = a[kk] + (use_warm ? b[kk] : t[kk]); });
                }
            }
        } else {
            T s2 = T(0), g1 = T(0), g2 = T(0);
            static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; s2 += x[i] * x[i]; g1 += x[i] * (Ma[i] - smooth[i]); g2 += T(0.5) * x[i] * Mx[i]; });
            const T snorm = dl_sqrt(s2);
            if (snorm < T(1e-15)) stop = true;
            else {
                const T gtol = m.tolerance * m.ls_tolerance * snorm * m.meaninertia * nvf + m.ls_reltol * dl_abs(g1);
                alpha = linesearch<T, TP>(mem, nefc, gauss, g1, g2, gtol, m.ls_iterations);
                if (alpha == T(0)) stop = true;
                else {
                    static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; qacc[i] += alpha * x[i]; Ma[i] += alpha * Mx[i]; });
                    for (int r = 0; r < nefc; r += 4) {
                        T a[4], b[4];
                        static_for<4>([&](auto ki) { a[ki.value] = mem(L::ROW_JAREF + r + ki.value); b[ki.value] = mem(L::ROW_JV + r + ki.value); });
                        static_for<4>([&](auto ki) { mem(L::ROW_JAREF + r + ki.value) = a[ki.value] + alpha * b[ki.value]; });
                    }
                }
            }
        }
        if (stop) break;
        if (phase == 0) { phase = 1; static_for<TP::NV>([&](auto ii) { x[ii.value] = warm[ii.value]; }); continue; }

        // ---- Hessian: H = M at the start, then add/remove rows whose state flipped
        const T oldcost = cost;
        if (phase <= 1) {
            static_for<TP::NV>([&](auto ii) {
                constexpr int i = ii.value;
                static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; if constexpr (j <= i && TP::dof_anc(i, j)) H.a[i][j] = mem(L::MAT + L::template MI<i, j>); });
            });
        }
        SV<T> W[TP::NB];                  // constraint wrench per body (about the reference point)
        T fcon[TP::NV];
        T c = T(0);
        if (phase >= 1) {
            static_for<TP::NB>([&](auto bi) { W[bi.value] = {mk<T>(0, 0, 0), mk<T>(0, 0, 0)}; });
            static_for<TP::NV>([&](auto ii) { fcon[ii.value] = T(0); });
            for (int r = 0; r < e.nlim; r++) {
                const T jar = mem(L::ROW_JAREF + r), D = mem(L::ROW_D + r);
                const bool on = jar < T(0), was = (act_lo >> r) & 1ull;
                const int code = (int)mem(L::LIM_CODE + r), j = code & 31;
                const T f = on ? ((code & 32) ? D * jar : -D * jar) : T(0);
                if (on) c += T(0.5) * D * jar * jar;
                const T dH = (on == was) ? T(0) : (on ? D : -D);
                static_for<TP::NV>([&](auto ji) {
                    if constexpr (TP::dof_limited(ji.value)) if (j == ji.value) { fcon[ji.value] += f; H.a[ji.value][ji.value] += dH; }
                });
                if (on != was) act_lo ^= 1ull << r;
            }
            for (int cc = 0; cc < e.ncon; cc++) {
                const int r0 = e.nlim + 4 * cc;
                const int body = (int)mem(L::CON_BODY + cc);
                const V3<T> p = mk<T>(mem(L::CON_PX + cc), mem(L::CON_PY + cc), mem(L::CON_PZ + cc));
                const T tx = mem(L::CON_TX + cc), ty = mem(L::CON_TY + cc), mu = mem(L::CON_MU + cc);
                T fs[4], Dr[4];
                unsigned flips = 0, ons = 0;
                for (int s = 0; s < 4; s++) {
                    const int r = r0 + s;
                    const T jar = mem(L::ROW_JAREF + r);
                    Dr[s] = mem(L::ROW_D + r);
                    const bool on = jar < T(0);
                    const bool was = r < 64 ? ((act_lo >> r) & 1ull) : ((act_hi >> (r - 64)) & 1ull);
                    fs[s] = on ? -Dr[s] * jar : T(0);
                    if (on) { c += T(0.5) * Dr[s] * jar * jar; ons |= 1u << s; }
                    if (on != was) { flips |= 1u << s; if (r < 64) act_lo ^= 1ull << r; else act_hi ^= 1ull << (r - 64); }
                }
                if (ons) {
                    // world force of the pyramid edges and its moment about the reference point
                    const T fn = fs[0] + fs[1] + fs[2] + fs[3], f1 = mu * (fs[0] - fs[1]), f2 = mu * (fs[2] - fs[3]);
                    const V3<T> F = mk<T>(f1 * tx - f2 * ty, f1 * ty + f2 * tx, fn);
                    const V3<T> Nm = cross(p, F);
                    static_for<TP::NB - 1>([&](auto bi) {
                        constexpr int b = bi.value + 1;
                        if (body == b) { W[b].w = W[b].w + Nm; W[b].v = W[b].v + F; }
                    });
                }
                if (flips) {
                    T jn[TP::NV], jt1[TP::NV], jt2[TP::NV];
                    contact_jac<T, TP>(k, body, p, tx, ty, jn, jt1, jt2);
                    for (int s = 0; s < 4; s++) {
                        if (!((flips >> s) & 1u)) continue;
                        const T D = ((ons >> s) & 1u) ? Dr[s] : -Dr[s];
                        const T sg = (s & 1) ? -mu : mu;
                        T row[TP::NV];
                        static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; row[j] = jn[j] + sg * (s < 2 ? jt1[j] : jt2[j]); });
                        static_for<TP::NV>([&](auto ii) {
                            constexpr int i = ii.value;
                            const T di = D * row[i];
                            static_for<TP::NV>([&](auto ji) { constexpr int j = ji.value; if constexpr (j <= i && TP::dof_anc(i, j)) H.a[i][j] += di * row[j]; });
                        });
                    }
                }
            }
            // J^T f: wrenches to the root, then project on the motion subspaces
            static_for<TP::NB - 2>([&](auto bi) {
                constexpr int b = TP::NB - 1 - bi.value, pb = TP::body_parent(b);
                if constexpr (pb > 0) W[pb] = W[pb] + W[b];
            });
            T g = T(0);
            static_for<TP::NV>([&](auto ii) {
                constexpr int i = ii.value;
                fcon[i] += sdot(dof_S<T, TP, i>(k), W[TP::dof_body(i)]);
                g += T(0.5) * (Ma[i] - smooth[i]) * (qacc[i] - qacc_smooth[i]);
                rhs[i] = Ma[i] - smooth[i] - fcon[i];           // gradient
            });
            gauss = g;
            cost = c + g;
        } else {
            static_for<TP::NV>([&](auto ii) { rhs[ii.value] = smooth[ii.value]; });
        }
        // termination tests of the previous Newton step (need the new cost and gradient)
        if (phase >= 2) {
            T gn = T(0);
            static_for<TP::NV>([&](auto ii) { gn += rhs[ii.value] * rhs[ii.value]; });
            const T improvement = scale * (oldcost - cost), gradient = scale * dl_sqrt(gn);
            iter++;
            // float32 (tol_rel > 0): relative terms absorb the rounding noise of cost / gradient
            T gmag = T(0);
            static_for<TP::NV>([&](auto ii) { gmag += dl_abs(Ma[ii.value]) + dl_abs(smooth[ii.value]); });
            if (improvement < m.tolerance + m.tol_rel * scale * dl_abs(cost) || gradient < m.tolerance + m.tol_rel * scale * gmag || iter >= m.iterations) break;
        }
        // ---- factorise a copy of H and solve
        {
            TreeMat<T, TP> Hf;
            static_for<TP::NV>([&](auto ii) {
                constexpr int i = ii.value;
                static_for<TP::NV>([&](auto ji) {
                    constexpr int j = ji.value;
                    if constexpr (j <= i && TP::dof_anc(i, j)) Hf.a[i][j] = H.a[i][j];
                });
            });
            ltdl_factor<T, TP>(Hf);
            ltdl_solve<T, TP>(Hf, rhs);
        }
        if (phase == -1) {
            static_for<TP::NV>([&](auto ii) { qacc_smooth[ii.value] = rhs[ii.value]; x[ii.value] = rhs[ii.value]; });
            if (nefc == 0) { static_for<TP::NV>([&](auto ii) { qacc[ii.value] = qacc_smooth[ii.value]; }); break; }
            phase = 0;
        } else {
            static_for<TP::NV>([&](auto ii) { x[ii.value] = -rhs[ii.value]; });     // Newton direction
            phase = phase + 1;
        }
    }
    niter = iter;
}

// The forward evaluation behind a real call: keeps the optimiser from hoisting its (many)
// loop-invariant sub-expressions out of the RK4 / frame-skip loops of the callers, which would
// keep hundreds of values alive across the whole evaluation and spill them.
template <typename T, typename TP>
DL_NOINLINE int forward_call(const DL_CONST DevModel<T, TP>* m, DL_LDS T* lane_base, int lane_stride) {
    using L = MemLayout<TP>;
#if defined(__HIP_DEVICE_COMPILE__) && defined(DL_INLINE_FORWARD)
    asm volatile("" : "+s"(m));     // launder the (uniform) model pointer: parameter-derived values must not be hoisted out of the callers' loops
#elif defined(__HIP_DEVICE_COMPILE__)
    {   // pointer arguments arrive in VGPRs; the model pointer is wave-uniform: move it to SGPRs so
        // that parameter reads are scalar loads
        const uint64_t p = (uint64_t)m;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        m = (const DL_CONST DevModel<T, TP>*)(((uint64_t)hi << 32) | lo);
        lane_stride = __builtin_amdgcn_readfirstlane(lane_stride);
    }
#endif
    const LaneMem<T> mem{lane_base, lane_stride};
    T ql[TP::NV], vl[TP::NV], wl[TP::NV], ul[TP::NU], al[TP::NV];
    static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; ql[i] = mem(L::IO_Q + i); vl[i] = mem(L::IO_V + i); wl[i] = mem(L::IO_WARM + i); });
    static_for<TP::NU>([&](auto ii) { ul[ii.value] = mem(L::IO_CTRL + ii.value); });
    EfcInfo<TP> e;
    int niter;
    forward<T, TP>(*m, mem, ql, vl, ul, wl, al, e, niter);
    static_for<TP::NV>([&](auto ii) { mem(L::IO_QACC + ii.value) = al[ii.value]; });
    return e.ncon | (e.nefc << 8) | (niter << 16);
}
// convenience wrapper: arrays in, arrays out (through the lane-memory IO slots)
template <typename T, typename TP>
DL_HD int forward_io(const DL_CONST DevModel<T, TP>& m, const LaneMem<T>& mem, const T (&q)[TP::NV], const T (&v)[TP::NV], const T (&ctrl)[TP::NU],
                     const T (&warm)[TP::NV], T (&qacc)[TP::NV]) {
    using L = MemLayout<TP>;
    static_for<TP::NV>([&](auto ii) { constexpr int i = ii.value; mem(L::IO_Q + i) = q[i]; mem(L::IO_V + i) = v[i]; mem(L::IO_WARM + i) = warm[i]; });
    static_for<TP::NU>([&](auto ii) { mem(L::IO_CTRL + ii.value) = ctrl[ii.value]; });
    const int info = forward_call<T, TP>(&m, mem.base, mem.stride);
    static_for<TP::NV>([&](auto ii) { qacc[ii.value] = mem(L::IO_QACC + ii.value); });
    return info;
}

// [3P] mj_step, RK4 (mj_RungeKutta N=4).  Returns true on divergence (mj_checkPos/Vel/Acc).
// The RK4 bookkeeping (start state, weighted sums) is staged in the lane's global workspace `gw`
// (4*NV words, coalesced [word][N]) instead of being held in registers across the four forward
// evaluations.
template <typename T, typename TP>
DL_HD bool mj_step_rk4(const DL_CONST DevModel<T, TP>& m, const LaneMem<T>& mem, const GlobalMem<T>& gw, T (&q)[TP::NV], T (&v)[TP::NV], const T (&ctrl)[TP::NU], T (&warm)[TP::NV]) {
    constexpr int NV = TP::NV;
    bool bad = false;
    static_for<NV>([&](auto ii) { bad = bad || dl_bad(q[ii.value]) || dl_bad(v[ii.value]); });
    if (bad) return true;
    const T h = m.timestep;
    static_for<NV>([&](auto ii) { constexpr int i = ii.value; gw(i) = q[i]; gw(NV + i) = v[i]; gw(2 * NV + i) = T(0); gw(3 * NV + i) = T(0); });
#pragma unroll 1
    for (int stage = 0; stage < 4; stage++) {
        T acc[NV];
        (void)forward_io<T, TP>(m, mem, q, v, ctrl, warm, acc);
        static_for<NV>([&](auto ii) { warm[ii.value] = acc[ii.value]; });
        if (stage == 0) {
            bool b2 = false;
            static_for<NV>([&](auto ii) { b2 = b2 || dl_bad(acc[ii.value]); });
            if (b2) { bad = true; break; }
        }
        // classic tableau: stage weights 1/6 1/3 1/3 1/6, next-stage step 1/2 1/2 1
        const T wgt = (stage == 0 || stage == 3) ? T(1) / T(6) : T(1) / T(3);
        const T a = stage == 2 ? T(1) : T(0.5);
        static_for<NV>([&](auto ii) {
            constexpr int i = ii.value;
            const T dq = gw(2 * NV + i) + wgt * v[i], dv = gw(3 * NV + i) + wgt * acc[i];
            gw(2 * NV + i) = dq; gw(3 * NV + i) = dv;
            const T vstage = v[i];
            if (stage < 3) { q[i] = gw(i) + h * a * vstage; v[i] = gw(NV + i) + h * a * acc[i]; }
            else { q[i] = gw(i) + h * dq; v[i] = gw(NV + i) + h * dv; }
        });
    }
    return bad;
}

}  // namespace dl
