// dl_group.hpp -- "16 lanes per walker" formulation of the forward dynamics (device only): the product path.
//
// The lane-per-walker kernels (dl_core.hpp) leave 15/16 of the chip idle at the benchmark size
// (4096 walkers = 64 waves on 1024 SIMDs) and their run time is the serial instruction stream of
// one walker.  Here a walker is spread over a 16-lane DPP row: a lane owns one degree of freedom (and the
// body whose last dof it is, the collision candidates j, j + 16, ..., contact j, its own limit row), four
// walkers share a wave, 4096 walkers are 1024 waves -- one per SIMD.  At one wave per SIMD nothing hides
// latency, so the design goal is few instructions and few dependent round trips:
//   * the dynamics run in registers; lanes exchange data through DPP (row_newbcast broadcasts, row_shr /
//     row_shl segmented scans over the dof tree, row_ror sums), mostly folded into v_fmac_f32_dpp;
//   * LDS holds only what is indexed dynamically (contacts, constraint rows, contact Jacobians, body frames
//     for the collision stage), in 16-byte groups; a workgroup is one wave, so LDS needs no barrier;
//   * the kinematic tree is a compile-time constant (GTopo<TP>), all numeric parameters are data.
//
// Models with more than 16 dofs (the 19-dof walker of BASELINE configs[3]) keep the 16-lane row: their NX = nv - 16
// leading dofs -- the world-aligned root translations, ancestors of every other dof -- are carried REPLICATED in every
// lane of the row (GX): their rows of the mass matrix / Hessian are uniform values or one coupling entry per lane,
// their Jacobian columns are the contact frame itself, and they are eliminated first in the Cholesky factorisation.
// NX = 0 (straight walker) compiles to exactly the lane-only code.
//
// Results follow dl_core.hpp / the CPU restatement (same model, same minimiser); the parity tests run both device
// paths against the CPU oracle, and tests/host_emu runs THIS source on the host (DL_GROUP_EMU: a wave as 64 fibers).
#pragma once

#if defined(DL_GROUP_EMU)
#include "dl_group_emu.hpp"       // tests/host_emu: host stand-ins for the wave-level builtins (test infrastructure)
#define DL_VPIN(x) ((void)0)
#define DL_SPIN(x) ((void)0)
#define DL_CLOCK() 0ll
#define DL_SLEEP() dlemu::sync(dlemu::TAG_SLEEP)          // a poll's sleep is where the emulated wave yields to its partner (run_pair)
#define DL_WAKE() dlemu::sync(dlemu::TAG_YIELD)          // every flag post is followed by s_wakeup: in the emulation the point where the posting wave may lose the SIMD to its partner
#define DL_WG_RELEASE() ((void)0)
#define DL_WG_ACQUIRE() ((void)0)
#define DL_FAULT_OR(p, code) ((void)(*(p) |= (code)))
#define DL_UNIFORM(x) (x)
#else
#include <hip/hip_runtime.h>
#define DL_VPIN(x) asm volatile("" : "+v"(x))
#define DL_SPIN(x) asm volatile("" : "+s"(x))
#define DL_CLOCK() ((long long)__builtin_readcyclecounter())
#ifndef DL_SLEEP_N
#define DL_SLEEP_N 1        // s_sleep argument of a poll (64 cycles per unit).  Round 5, A/B on one box against 16 (rounds 2-4) / 4 / 2 / 48: 1 is best on every line (+0.6 % headline, +1.4 % 19-dof walker, +0.8 / +1.1 % policy lines): the partner's reaction time is on the dynamics wave's critical path at every commit
#endif
#define DL_SLEEP() __builtin_amdgcn_s_sleep(DL_SLEEP_N)        // a waiting wave of a split workgroup: 64 x N cycles, cut short by the partner's s_wakeup
#define DL_WAKE() asm volatile("s_wakeup")          // NB (round 5, tools/ubench/snop_wakeup.hip): an s_wakeup also ENDS THE s_nop another wave of the workgroup is in, after one wait state.  The hand-written hazard padding of these kernels is the DPP one (DL_DPP_NOP below: one state, `s_nop 0`, which cannot be shortened); the policy kernels, whose MFMA results need 3 .. 10 states, wait with v_nop (dl_policy.hpp)
// the hand-over between the two waves of a split pair is the one place where DIFFERENT waves exchange data through LDS: workgroup-scope
// release before the flag store, acquire after the successful poll (g_sync's wavefront scope orders a wave against itself only)
#ifdef DL_EXP_NO_WG_FENCE       // experiment switch: wavefront scope only (the round-2 form)
#define DL_WG_RELEASE() ((void)0)
#define DL_WG_ACQUIRE() ((void)0)
#else
#define DL_WG_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#define DL_WG_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")
#endif
// The fault word's system-scope OR, hand-written.  As a builtin (rounds 3-5) it compiled to  v_readlane s[a:a+1] <- spilled pointer ; s_nop 3 ; global_atomic_or v0, v, s[a:a+1] sc1:
// hipcc's padding of the "VALU writes an SGPR -> VMEM reads it as its base: 5 wait states" hazard is ONE multi-state s_nop, and an s_wakeup executed by another wave of the workgroup
// ends the s_nop a wave is in after one state (tools/ubench/snop_wakeup.hip; dl_hwprobe.hpp).  The atomic then went out with whatever the SGPR pair held before the restore -- in
// round 6 a saved lane mask: a memory access fault at 0xffffffff........ in one of five runs of tests/test_gpu_bench_shapes.py::test_split_handover_timeout_raises (found under rocgdb,
// bisected with -DDL_EXP_NO_SRV_FAULT: EXPERIMENTS.md round 6).  Round 5's survey (tools/survey_snop.py) had exempted exactly these sites as "time-out paths".  Here the pointer is an
// operand of the statement (restored before it; the compiler pads nothing around inline asm) and the five states are v_nop, which nothing shortens.
__device__ __forceinline__ void dl_fault_or(int32_t* p, int code) {
    asm volatile("v_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tv_nop\n\tglobal_atomic_or %0, %1, %2 sc1" : : "v"(0), "v"(code), "s"(p) : "memory");
}
#define DL_FAULT_OR(p, code) ::dl_fault_or((p), (code))
#define DL_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)      // a wave-uniform int as a scalar (SGPR): scalar branches instead of exec-mask regions
#endif
// (the reasons in a handle's fault word, DL_FAULT_*: include/drloco_hip.h)

#include <type_traits>

#include "dl_env.hpp"

#ifndef DL_PIN_STRAIGHT
#define DL_PIN_STRAIGHT 1       // experiment switch: 0 builds the straight walker with the fetch-at-use policy of the 19-dof walker
#endif
// experiment switches: unroll factor of the contact-pair loops (J^T f / Hessian, J dir); 1 = as written
// experiment switches of the round-3 dependent-chain work in the Newton loop (1 = product)
#ifndef DL_OPT_JAC_PIPE
#define DL_OPT_JAC_PIPE 0
#endif
#ifndef DL_OPT_ARMIJO
#define DL_OPT_ARMIJO 1     // 0: always the exact line search (round-2 behaviour)
#endif
#ifndef DL_OPT_ARMIJO_C
#define DL_OPT_ARMIJO_C 0.1
#endif
#ifndef DL_OPT_GRADSQ
#define DL_OPT_GRADSQ 1
#endif
#ifndef DL_OPT_MX2
#define DL_OPT_MX2 1
#endif
#ifndef DL_OPT_FSQRT
#define DL_OPT_FSQRT 1
#endif
#ifndef DL_CHOL_SHORT_CHAIN
#define DL_CHOL_SHORT_CHAIN 1   // 0: round-2 form of the leaf-first factorisation / substitutions (experiment switch)
#endif
#ifndef DL_CHOL_X_LAST
#define DL_CHOL_X_LAST 1        // walkers with replicated root translations: eliminate them LAST, the lane block leaf-first (0: round-2 form -- replicated dofs first, dense lane block)
#endif
#ifndef DL_CHOL_LEAF_FIRST
#define DL_CHOL_LEAF_FIRST 1    // 0: the root-first dense row-per-lane Cholesky for every model (experiment switch)
#endif
#ifndef DL_PREFETCH_REFS
#define DL_PREFETCH_REFS 0     // experiment switch: 1 advances a copy of the cursor and requests the step's reference sample BEFORE the physics (measured: no gain -- the values wait in scratch, 68 instead of 44 spilled registers)
#endif
#ifndef DL_JAC_ON_PARTNER
#define DL_JAC_ON_PARTNER 0      // experiment switch: 1 = the partner wave of a split workgroup also builds the contact Jacobians (with its commit)
#endif
#ifndef DL_PREFETCH_ACTIONS
#define DL_PREFETCH_ACTIONS 1   // the constraint wave of a split workgroup touches the next control step's action row (0: off, experiment switch)
#endif
#ifndef DL_UNROLL_JTF
#define DL_UNROLL_JTF 1
#endif
#ifndef DL_UNROLL_APPLY
#define DL_UNROLL_APPLY 1
#endif
#define DL_PRAGMA_(x) _Pragma(#x)
#define DL_UNROLL(n) DL_PRAGMA_(unroll n)

namespace dl {

constexpr int GL = 16;          // lanes per walker
constexpr int GW = 4;           // walkers per wave

// sizes of the 16-lane formulation of model TP
template <typename TP> struct GD {
    static constexpr int NV = TP::NV;
    static constexpr int NX = NV > GL ? NV - GL : 0;      // leading root-translation dofs carried replicated
    static constexpr int NXA = NX > 0 ? NX : 1;           // array extent (no zero-length arrays)
    static constexpr int NL = NV - NX;                    // dofs that own a lane: lane l <-> dof l + NX
    static constexpr int MAXB = TP::NB;                   // bodies incl. world
    static constexpr int MAXCON = (TP::MAXCON + 1) & ~1;     // even: contacts are processed in pairs (straight walker 18: 29.3 KB of LDS per wave, five waves per CU)
    static constexpr int MAXROW = ((TP::NLIM + 4 * TP::MAXCON + 3) / 4) * 4;
    static constexpr int ncand_() { int n = 0; for (int g = 0; g < TP::NG; g++) n += TP::geom_type(g) ? 8 : 2; return n; }
    static constexpr int NCAND = ncand_();                // capsule end points / box corners in contact order
    static constexpr int NPASS = (NCAND + GL - 1) / GL;   // candidate c = lane + 16 * pass
    static constexpr bool slides_ok_() {                  // the replicated dofs: world-aligned slides of the root body, unlimited, not actuated
        for (int t = 0; t < NX; t++) if (TP::dof_type(t) != 0 || TP::dof_body(t) != 1 || TP::dof_limited(t) || TP::dof_parent(t) != t - 1) return false;
        for (int a = 0; a < TP::NU; a++) if (TP::act_dof(a) < NX) return false;
        return true;
    }
    // Register budget: the straight walker pins every model constant a lane touches in VGPRs (nothing of the model is loaded
    // inside an evaluation).  The 19-dof walker does not have the registers for that (three candidate passes, nine bodies, the
    // replicated dofs' state): what is used ONCE per evaluation -- collision candidates, body offsets, solimp -- is fetched at
    // its point of use from the L1-resident model block instead (a few hundred cycles per evaluation against scratch spills).
    static constexpr bool PIN_ALL = (NX == 0) && DL_PIN_STRAIGHT;
    // Contact Jacobians in LDS: one 16-byte record (normal, tangent 1, tangent 2, owner) per contact and dof lane.  A contact's column is non-zero only for
    // the lanes on the chain root -> the contact's body, and the lanes of a chain have distinct depths: the 19-dof walker stores a contact's records BY DEPTH
    // (JCL = deepest chain = 8 slots instead of 16 lanes; the record's fourth word names the lane that owns the slot, a lane off the chain finds another
    // owner -- or none -- at its depth and takes zero).  24 contacts x 16 lanes were 6 KB of the walker's 9.98 KB; packed, a split workgroup of sixteen
    // 19-dof walkers fits the CU's LDS.  The lane-only walker keeps the lane-major layout (JCL = 16: nothing to mask).
    static constexpr int lane_depth(int l) { int d = 0; for (int p = TP::dof_parent(l + NX) - NX; p >= 0; p = TP::dof_parent(p + NX) - NX) d++; return d; }
    static constexpr int max_depth_() { int m = 0; for (int l = 0; l < NL; l++) if (lane_depth(l) > m) m = lane_depth(l); return m; }
    static constexpr bool JC_PACKED = NX > 0 && max_depth_() + 1 <= GL / 2;
    static constexpr int JCL = JC_PACKED ? ((max_depth_() + 2) & ~1) : GL;
    static_assert(slides_ok_(), "the dofs beyond 16 must be leading root translations");
    static_assert(NL <= GL && MAXB <= 16 && NCAND <= 64 && TP::NS <= GL, "model too large for a 16-lane row");
};
// a value per replicated dof (uniform over the lanes of a walker's row)
template <typename T, int NX> struct GX { T x[NX > 0 ? NX : 1]; };
template <int AX, typename T> __device__ __forceinline__ T& vcomp(V3<T>& a) { if constexpr (AX == 0) return a.x; else if constexpr (AX == 1) return a.y; else return a.z; }
template <int AX, typename T> __device__ __forceinline__ T vcomp(const V3<T>& a) { if constexpr (AX == 0) return a.x; else if constexpr (AX == 1) return a.y; else return a.z; }

// what a lane needs about ITS OWN dof / body / collision candidates; built once on the host (g_load_lane) for the 16
// lanes and kept in the model block, so that a kernel fetches its lane's record with a handful of wide loads
// the collision candidates of a lane (capsule end points / box corners in contact order, c = j + 16 * pass)
// as body-local constants: point, radius (0 for a box corner), capsule axis (tangent direction), corner relative
// to the box centre (mjc_PlaneBox keeps only corners below the centre), friction
template <typename T, int NP> struct GLaneCand {
    int cinfo[NP];                                     // bit0 valid, 1 box, 2-4 sub index, 5-8 body
    T cpl[NP][3], crad[NP], cal[NP][3], crl[NP][3], cmu[NP], cinvw[NP];   // cinvw: body_invweight0 of the candidate's body (diagApprox of its contact rows)
};
template <typename T, int NP> struct GLane {
    int type, axis, body, limited, act;
    T sign, qpos0, range_lo, range_hi, damping, armature, invw, ctrl_lo, ctrl_hi, force_lo, force_hi, gear;
    // inertial parameters of the body of this dof
    T mass, ipos[3], inertia[3];
    GLaneCand<T, NP> cand;
};

// model as data (host-built from dl_model_desc), read through the constant address space.  Per-dof arrays are indexed
// by LANE (dof - NX); the replicated dofs have their own small arrays (xs_*).
template <typename T, typename TP> struct GModel {
    using D = GD<TP>;
    int32_t nv, nb, nu, ngeom, nsite, frame_skip, iterations, ls_iterations, ncand, root_last_dof;
    T timestep, gravity_z, solK, solB, solimp[5], solimp_inv[3], meaninertia, tolerance, ls_tolerance, ls_reltol, tol_rel, root_z0;
    int32_t dof_body[GL], dof_type[GL], dof_axis[GL], dof_limited[GL];
    T dof_sign[GL], qpos0[GL], range_lo[GL], range_hi[GL], damping[GL], armature[GL], dof_invw[GL];
    int32_t dof_act[GL];                       // actuator index driving this dof, or -1
    T ctrl_lo[GL], ctrl_hi[GL], force_lo[GL], force_hi[GL], gear[GL];   // indexed by lane
    T xs_qpos0[D::NXA], xs_damping[D::NXA], xs_armature[D::NXA];        // replicated root translations
    T body_pos[D::MAXB][3], body_mass[D::MAXB], body_ipos[D::MAXB][3], body_inertia[D::MAXB][3], body_invw[D::MAXB];
    int32_t cand_geom[GL * D::NPASS], cand_sub[GL * D::NPASS];
    int32_t geom_body[D::MAXB], geom_type[D::MAXB];
    T geom_pos[D::MAXB][3], geom_mat[D::MAXB][9], geom_size[D::MAXB][3], geom_friction[D::MAXB], floor_friction;
    int32_t site_body[GL];
    T site_pos[GL][3];
    GLane<T, D::NPASS> lanes[GL];              // per-lane records (g_load_lane of the fields above), filled by fill_group_model
};

// per-walker LDS layout (in elements of T).  Regions that are never live at the same time share their space: the mirror
// block of the mass matrix (MM, smooth dynamics) sits in the constraint rows (first written by the constraint stage), the
// body frames (BFR, dead once the contacts exist) in the contact Jacobians, the q / v staging of the observation writer
// (outside the forward evaluation) in the contact forces.  10 KB per walker for the 19-dof walker: four waves per CU.
template <typename TP> struct GLds {
    using D = GD<TP>;
    static constexpr int MAXCON = D::MAXCON, MAXROW = D::MAXROW;
    static constexpr int ROW = 0;                                    // rows [4][MAXROW]: D, JAREF, JV, TMP; contact c owns rows 4c..4c+3 (16-byte groups), limits follow
    static constexpr int R_D = 0, R_JAREF = 1, R_JV = 2, R_TMP = 3;
    static constexpr int MS = GL + 4;                                // row stride of M: rows are written as four 16-byte groups, columns read with consecutive lanes
    static constexpr int MM = ROW;                                   // M [16 rows][MS]: every lane's chain part of its row (transposed read-back gives the rest)
    static constexpr int CON_W = 8;                                  // contact record: (px py pz body) (tx ty mu dist): two 16-byte groups
    static constexpr int C_P = 0, C_BODY = 3, C_TX = 4, C_TY = 5, C_MU = 6, C_DIST = 7;
    static constexpr int CON = ROW + 4 * MAXROW;                     // contacts [MAXCON][CON_W]
    static constexpr int FC_W = 12;                                  // per contact: Fn F1 F2 flip | w00 w01 w02 w11 | w22 Fx Fy Fz (world-frame force: the replicated dofs' J^T f)
    static constexpr int FC = CON + CON_W * MAXCON;                  // contact-frame force and Hessian weights [MAXCON][FC_W]
    static constexpr int Q = FC, V = Q + D::NV;                      // q, v staged for the observation writer
    static constexpr int MISC = FC + FC_W * MAXCON;                  // rootz ... [8]
    static constexpr int JCL = D::JCL;                               // record slots per contact: 16 lanes, or (GD::JC_PACKED) the depth of the deepest chain
    static constexpr int JC = MISC + 8;                              // contact Jacobians [MAXCON][JCL][4]: normal, tangent 1, tangent 2, owner lane (packed form)
    static constexpr int BFR_W = 12;                                 // body frame record: X (3) Y (3) Z (3) pos (3) = three 16-byte groups
    static constexpr int BFR = JC;                                   // body frames [MAXB][BFR_W]
    static constexpr int TOTAL_RAW = JC + MAXCON * JCL * 4;
    // walker regions are offset by 16 (mod 32) words so that the two rows of a half-wave use disjoint banks
    static constexpr int TOTAL = ((TOTAL_RAW + 31) / 32) * 32 + 16;
    static_assert(GL * MS <= 4 * MAXROW && 2 * D::NV <= FC_W * MAXCON && D::MAXB * BFR_W <= MAXCON * JCL * 4, "aliased regions must fit");
    static_assert(MS % 4 == 0 && CON % 4 == 0 && FC % 4 == 0 && JC % 4 == 0 && MAXROW % 4 == 0 && TOTAL % 4 == 0, "16-byte groups must stay aligned");
};

// everything a lane needs about ITS OWN dof / body / collision candidates (incl. the inverse weight of the candidates'
// bodies) is preloaded into registers (GLane), uniform scalars are pinned in VGPRs (GConst); nothing of the model is
// indexed dynamically inside the loops, so there is no shared model block in LDS.

template <typename T, typename TP, typename MODEL>       // MODEL: GModel<T, TP> in whatever address space the caller holds it
__host__ __device__ __forceinline__ void g_load_lane(const MODEL& m, int j, GLane<T, GD<TP>::NPASS>& ln) {
    constexpr int NL = GD<TP>::NL;
    const int jj = j < NL ? j : 0;
    ln.type = m.dof_type[jj]; ln.axis = m.dof_axis[jj]; ln.body = m.dof_body[jj]; ln.limited = m.dof_limited[jj]; ln.act = m.dof_act[jj];
    ln.sign = m.dof_sign[jj]; ln.qpos0 = m.qpos0[jj]; ln.range_lo = m.range_lo[jj]; ln.range_hi = m.range_hi[jj];
    ln.damping = m.damping[jj]; ln.armature = m.armature[jj]; ln.invw = m.dof_invw[jj];
    ln.ctrl_lo = m.ctrl_lo[jj]; ln.ctrl_hi = m.ctrl_hi[jj]; ln.force_lo = m.force_lo[jj]; ln.force_hi = m.force_hi[jj]; ln.gear = m.gear[jj];
    const int b = ln.body;                            // inertial parameters of the body this dof belongs to
    ln.mass = m.body_mass[b];
    for (int k = 0; k < 3; k++) { ln.ipos[k] = m.body_ipos[b][k]; ln.inertia[k] = m.body_inertia[b][k]; }
    for (int pass = 0; pass < GD<TP>::NPASS; pass++) {
        const int c = j + GL * pass;
        const bool ok = c < m.ncand;
        const int ge = ok ? m.cand_geom[c] : 0, sub = ok ? m.cand_sub[c] : 0;
        const bool box = m.geom_type[ge] != 0;
        ln.cand.cinfo[pass] = (ok ? 1 : 0) | (box ? 2 : 0) | (sub << 2) | (m.geom_body[ge] << 5);
        const auto* mat = m.geom_mat[ge];
        T rel[3];
        if (box) {
            const T sx = (sub & 1) ? m.geom_size[ge][0] : -m.geom_size[ge][0];
            const T sy = (sub & 2) ? m.geom_size[ge][1] : -m.geom_size[ge][1];
            const T sz = (sub & 4) ? m.geom_size[ge][2] : -m.geom_size[ge][2];
            for (int k = 0; k < 3; k++) { rel[k] = mat[3 * k] * sx + mat[3 * k + 1] * sy + mat[3 * k + 2] * sz; ln.cand.cal[pass][k] = T(0); ln.cand.crl[pass][k] = rel[k]; }
            ln.cand.crad[pass] = T(0);
        } else {
            const T hs = sub == 0 ? m.geom_size[ge][1] : -m.geom_size[ge][1];
            for (int k = 0; k < 3; k++) { ln.cand.cal[pass][k] = mat[3 * k + 2]; rel[k] = hs * mat[3 * k + 2]; ln.cand.crl[pass][k] = T(0); }
            ln.cand.crad[pass] = m.geom_size[ge][0];
        }
        for (int k = 0; k < 3; k++) ln.cand.cpl[pass][k] = m.geom_pos[ge][k] + rel[k];
        ln.cand.cmu[pass] = m.geom_friction[ge];          // the contact uses max(geom, floor) with the walker's floor friction
        ln.cand.cinvw[pass] = m.body_invw[m.geom_body[ge]];
    }
}

// ------------------------------------------------------------------------------------------
// DPP row (16-lane) primitives
// row_ror / row_newbcast read a valid lane of the own row for every lane, so there is no "old" value to keep:
// bound_ctrl lets the compiler fold the DPP read into the consuming VOP2 instruction (v_add/v_mul/v_fmac ..._dpp)
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_f(double x) {
    const uint64_t u = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// sum over the 16 lanes of a row, result in every lane (row_ror:8,4,2,1).  Every lane must get the SAME
// bits (the solver's stopping decisions are taken per lane from these sums): with rotations the partial sums
// have period 8, 4, 2, 1, so commutativity alone guarantees it -- provided the compiler does not contract a
// multiply feeding x into the first add (fma(a_i, b_i, t_{i+8}) != fma(a_{i+8}, b_{i+8}, t_i)).  The empty
// asm hides the producer of x; the pragma keeps the adds themselves un-contracted.
template <typename T> __device__ __forceinline__ T gsum(T x) {
#pragma clang fp contract(off)
    DL_VPIN(x);
    x += dpp_f<0x128>(x);
    x += dpp_f<0x124>(x);
    x += dpp_f<0x122>(x);
    x += dpp_f<0x121>(x);
    return x;
}
// the same reduction for NV values at once: the NV dependent DPP chains are interleaved, which hides the two wait
// states every DPP read of a freshly written VGPR costs
template <int NV, typename T> __device__ __forceinline__ void gsum_n(T (&x)[NV]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < NV; i++) DL_VPIN(x[i]);
#pragma unroll
    for (int i = 0; i < NV; i++) x[i] += dpp_f<0x128>(x[i]);
#pragma unroll
    for (int i = 0; i < NV; i++) x[i] += dpp_f<0x124>(x[i]);
#pragma unroll
    for (int i = 0; i < NV; i++) x[i] += dpp_f<0x122>(x[i]);
#pragma unroll
    for (int i = 0; i < NV; i++) x[i] += dpp_f<0x121>(x[i]);
}
__device__ __forceinline__ bool gany(bool p) { return gsum(p ? 1.0f : 0.0f) > 0.0f; }

// A workgroup is ONE wave and a wave's LDS operations execute in order, so lanes exchange data through LDS
// without s_barrier: what is needed is that the compiler keeps the program order of the LDS accesses.
template <typename T> __device__ __forceinline__ void g_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T> __device__ __forceinline__ void g_pin(T& x) { DL_VPIN(x); }

template <typename T> __device__ __forceinline__ V3<T> ld3(DL_LDS T* p, int stride) { return {p[0], p[stride], p[2 * stride]}; }
// four consecutive, 16-byte aligned LDS words as one ds_read_b128 / ds_write_b128 (float); plain accesses for double
template <typename T> struct Q4 { T a, b, c, d; };
#if !defined(DL_GROUP_EMU)
__device__ __forceinline__ Q4<float> ld4(const DL_LDS float* p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = *(const DL_LDS f4*)p;
    return {v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(DL_LDS float* p, float a, float b, float c, float d) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v; v.x = a; v.y = b; v.z = c; v.w = d;
    *(DL_LDS f4*)p = v;
}
#else
__device__ __forceinline__ Q4<float> ld4(const DL_LDS float* p) { return {p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void st4(DL_LDS float* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
#endif
__device__ __forceinline__ Q4<double> ld4(const DL_LDS double* p) { return {p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void st4(DL_LDS double* p, double a, double b, double c, double d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }

// value of lane K of the row in every lane: ONE DPP instruction (row_newbcast:K, gfx90a+), usually folded
// into the consuming VALU instruction
template <int K> __device__ __forceinline__ float rbcast(float x) { return dpp_f<0x150 + K>(x); }
template <int K> __device__ __forceinline__ double rbcast(double x) { return dpp_f<0x150 + K>(x); }

#if !defined(DL_GROUP_EMU)
// d += bcast_K(a) * b  (SIGN = +1)  or  d -= bcast_K(a) * b  (SIGN = -1).  float: ONE instruction, v_fmac_f32_dpp with the
// row broadcast folded into src0 (the compiler only folds DPP into add/mul).  The DPP read of `a` needs two wait states
// after the VALU write of `a`; inline asm is opaque to the hazard recogniser, so callers fence with g_dpp_ready(a).
template <int K, int SIGN> __device__ __forceinline__ void fmac_bcast(float& d, float a, float b) {
    if constexpr (SIGN > 0) asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "v"(b), "n"(K));
    else asm("v_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "v"(b), "n"(K));
}
// d += dpp<CTRL>(a) * b for the row shifts of the segmented scans (lanes without a source lane read 0)
#define DL_FMAC_DPP_CASE(code, text) \
    if constexpr (CTRL == code) asm("v_fmac_f32_dpp %0, %1, %2 " text " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(d) : "v"(a), "v"(b));
template <int CTRL> __device__ __forceinline__ void fmac_dpp(float& d, float a, float b) {
    DL_FMAC_DPP_CASE(0x111, "row_shr:1") DL_FMAC_DPP_CASE(0x112, "row_shr:2") DL_FMAC_DPP_CASE(0x114, "row_shr:4") DL_FMAC_DPP_CASE(0x118, "row_shr:8")
    DL_FMAC_DPP_CASE(0x101, "row_shl:1") DL_FMAC_DPP_CASE(0x102, "row_shl:2") DL_FMAC_DPP_CASE(0x104, "row_shl:4") DL_FMAC_DPP_CASE(0x108, "row_shl:8")
}
#undef DL_FMAC_DPP_CASE
// x += dpp<CTRL>(x) * b with x as destination AND DPP source of the same instruction (no register copy)
#define DL_FMAC_SELF_CASE(code, text) \
    if constexpr (CTRL == code) asm("v_fmac_f32_dpp %0, %0, %1 " text " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(b));
template <int CTRL> __device__ __forceinline__ void fmac_dpp_self(float& x, float b) {
    DL_FMAC_SELF_CASE(0x111, "row_shr:1") DL_FMAC_SELF_CASE(0x112, "row_shr:2") DL_FMAC_SELF_CASE(0x114, "row_shr:4") DL_FMAC_SELF_CASE(0x118, "row_shr:8")
    DL_FMAC_SELF_CASE(0x101, "row_shl:1") DL_FMAC_SELF_CASE(0x102, "row_shl:2") DL_FMAC_SELF_CASE(0x104, "row_shl:4") DL_FMAC_SELF_CASE(0x108, "row_shl:8")
}
#undef DL_FMAC_SELF_CASE
template <int K> __device__ __forceinline__ void fmac_bcast_self(float& x, float b) {
    asm("v_fmac_f32_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b), "n"(K));
}
// Wait states between a VALU write of a register and its read through DPP in the hand-written statements.  The ISA manual asks for TWO and hipcc pads its own DPP
// instructions with `s_nop 1`; measured on gfx950 (dl_hwprobe.hpp / tools/ubench/dpp_wait.hip, profiles/r05_dpp_wait.txt: seven producers x six DPP forms, alone, beside
// s_wakeup, beside VALU + DPP work, beside MFMAs): with NO wait the read is stale, with ONE state never (0 of 11 G lane-reads).  Two code objects are built (drloco_amd/lib.py):
//   libdrloco_hip.so       DL_DPP_WAIT = 2, the DEFAULT: the manual's two states in their wakeup-proof form, `s_nop 0` twice (an `s_nop 1` is worth ONE state once another
//                          wave's s_wakeup ends it: tools/ubench/snop_wakeup.hip) -- spec-conformant on any gfx9 part;
//   libdrloco_hip_dpp1.so  -DDL_DPP_WAIT=1: one state, `s_nop 0` (+2.5 % headline, +3.6 % 19-dof walker).  Used only on a device that has just PROVEN, in this process, that one
//                          state is enough and that the test can fail (dl_hw_probe: stale reads with no wait, none with one): drloco_amd.lib.load() asks the conformant
//                          library to run the probe and switches only on that evidence; dl_create of this build runs the probe itself and refuses otherwise (DL_E_HIP).
// tools/check_dpp_hazards.py checks each listing against its DL_DPP_WAIT (--need).
#ifndef DL_DPP_WAIT
#define DL_DPP_WAIT 2
#endif
#if DL_DPP_WAIT == 2
#define DL_DPP_NOP "s_nop 0\n\ts_nop 0"
#else
#define DL_DPP_NOP "s_nop 0"
#endif
__device__ __forceinline__ void g_dpp_ready(float& a) { asm volatile(DL_DPP_NOP : "+v"(a)); }
// max(bcast_K(x), lo): the row broadcast folded into the v_max.  The wait between the VALU write of x and its DPP read (DL_DPP_NOP) is
// part of the statement: the register allocator may place a copy of x right before an asm statement, behind a separate g_dpp_ready.
template <int K> __device__ __forceinline__ float max_bcast(float x, float lo) {
    float d;
    asm(DL_DPP_NOP "\n\tv_max_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(x), "v"(lo), "n"(K));
    return d;
}
// x += bcast_K(x) * b for a chain of such steps on one register (each reads what the previous one wrote: wait states included, see max_bcast)
template <int K> __device__ __forceinline__ void fmac_bcast_chain(float& x, float b) {
    asm(DL_DPP_NOP "\n\tv_fmac_f32_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b), "n"(K));
}
// x in the lanes whose dof index (lane & 15) has its bit set in the COMPILE-TIME pattern M16, zero in the others.  Written as `(j < k) ? x : 0` the thirteen lane
// predicates of the factorisation were computed once per launch and kept as thirteen 64-bit lane masks -- 26 SGPRs of a kernel that has none to spare: every use
// reloaded its pair from a VGPR lane (two v_readlane per select: 26 VALU instructions per Newton iteration, profiles/r06_asm_mix.txt).  As a literal the mask is two s_mov_b32
// on the scalar pipe, rematerialised where it is used.
#ifndef DL_OPT_LITERAL_MASKS
// The thirteen (j < k) lane predicates of the leaf-first factorisation, round 6 (profiles/r06_asm_mix.txt, EXPERIMENTS.md): written plainly they are computed once per launch, kept as 64-bit
// lane masks and reloaded from VGPR lanes -- two v_readlane per use, the 26 lane spills of the Newton loop.  0: that form.  1: literal SGPR masks in an inline-asm v_cndmask (lane_keep):
// bit-identical, 3.3 % SLOWER (the masks stay live and push other values out).  2 (default): the predicates are formed per factorisation from a lane index made opaque at the top of
// g_chol_rev / the solve: one v_cmp per use, no spill in the Newton loop, 65 fewer in the control-step body; bit-identical, +0.1 % headline, +0.4 % 19-dof walker, +1.0 % --policy.
#define DL_OPT_LITERAL_MASKS 2
#endif
template <uint32_t M16> __device__ __forceinline__ float lane_keep(float x, int j) {
#if DL_OPT_LITERAL_MASKS == 1
    constexpr uint64_t M = 0x0001000100010001ull * (uint64_t)(M16 & 0xffffu);
    float d;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(d) : "v"(x), "s"(M));
    return d;
#else
    return ((M16 >> j) & 1u) ? x : 0.0f;
#endif
}
// one wait for a whole group of values that are about to be read through DPP
template <int NV> __device__ __forceinline__ void g_dpp_ready_n(float (&x)[NV]) {
    if constexpr (NV == 2) asm volatile(DL_DPP_NOP : "+v"(x[0]), "+v"(x[1]));
    else if constexpr (NV == 3) asm volatile(DL_DPP_NOP : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
    else if constexpr (NV == 4) asm volatile(DL_DPP_NOP : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else if constexpr (NV == 6) asm volatile(DL_DPP_NOP : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
    else if constexpr (NV == 16) asm volatile(DL_DPP_NOP : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
    else { for (int i = 0; i < NV; i++) asm volatile(DL_DPP_NOP : "+v"(x[i])); }
}
#else
// host emulation: the generic forms (a DPP read followed by a multiply-add)
template <int K, int SIGN> __device__ __forceinline__ void fmac_bcast(float& d, float a, float b) { if constexpr (SIGN > 0) d += rbcast<K>(a) * b; else d -= rbcast<K>(a) * b; }
template <int CTRL> __device__ __forceinline__ void fmac_dpp(float& d, float a, float b) { d += dpp_f<CTRL>(a) * b; }
template <int CTRL> __device__ __forceinline__ void fmac_dpp_self(float& x, float b) { x += dpp_f<CTRL>(x) * b; }
template <int K> __device__ __forceinline__ void fmac_bcast_self(float& x, float b) { x += rbcast<K>(x) * b; }
__device__ __forceinline__ void g_dpp_ready(float&) {}
template <int K> __device__ __forceinline__ float max_bcast(float x, float lo) { const float b = rbcast<K>(x); return b > lo ? b : lo; }
template <int K> __device__ __forceinline__ void fmac_bcast_chain(float& x, float b) { x += rbcast<K>(x) * b; }
template <uint32_t M16> __device__ __forceinline__ float lane_keep(float x, int j) { return ((M16 >> j) & 1u) ? x : 0.0f; }
template <int NV> __device__ __forceinline__ void g_dpp_ready_n(float (&)[NV]) {}
#endif
template <int K, int SIGN> __device__ __forceinline__ void fmac_bcast(double& d, double a, double b) {
    if constexpr (SIGN > 0) d += rbcast<K>(a) * b; else d -= rbcast<K>(a) * b;
}
template <int CTRL> __device__ __forceinline__ void fmac_dpp(double& d, double a, double b) { d += dpp_f<CTRL>(a) * b; }
template <int CTRL> __device__ __forceinline__ void fmac_dpp_self(double& x, double b) { x += dpp_f<CTRL>(x) * b; }
template <int K> __device__ __forceinline__ void fmac_bcast_self(double& x, double b) { x += rbcast<K>(x) * b; }
template <int NV> __device__ __forceinline__ void g_dpp_ready_n(double (&)[NV]) {}
__device__ __forceinline__ void g_dpp_ready(double&) {}
template <int K> __device__ __forceinline__ double max_bcast(double x, double lo) { const double b = rbcast<K>(x); return b > lo ? b : lo; }
template <int K> __device__ __forceinline__ void fmac_bcast_chain(double& x, double b) { x += rbcast<K>(x) * b; }
template <uint32_t M16> __device__ __forceinline__ double lane_keep(double x, int j) { return ((M16 >> j) & 1u) ? x : 0.0; }

// ------------------------------------------------------------------------------------------
// Uniform model scalars of the hot path, read ONCE per kernel and pinned in VGPRs (the empty asm makes the values
// opaque: the compiler would otherwise re-issue the scalar loads inside the solver loops and wait for each --
// s_waitcnt lgkmcnt(0) also drains the LDS queue).  VGPRs are plentiful at one wave per SIMD.
template <typename T, typename TP> struct GConst {
    using D = GD<TP>;
    T body_pos[D::MAXB][3], root_z0, gravity_z, solK, solB, solimp[5], solimp_inv[3], meaninertia, tolerance, ls_tolerance, ls_reltol, tol_rel, scale, nvf;
    T xs_qpos0[D::NXA], xs_damping[D::NXA], xs_armature[D::NXA];
    int iterations, ls_iterations;
};
// PIN: the register policy of the caller -- everything a lane touches pinned in VGPRs (the straight walker's dynamics wave), or body offsets / solimp / collision
// candidates fetched where they are used (the 19-dof walker; the partner wave of a split workgroup, which has time but needs its registers for what it holds
// from one evaluation to the next)
template <typename T, typename TP, bool PIN = GD<TP>::PIN_ALL>
__device__ __forceinline__ void g_load_const(const DL_CONST GModel<T, TP>& m, GConst<T, TP>& c) {
    if constexpr (PIN) for (int b = 0; b < GD<TP>::MAXB; b++) for (int k = 0; k < 3; k++) { c.body_pos[b][k] = b < m.nb ? m.body_pos[b][k] : T(0); g_pin(c.body_pos[b][k]); }
    c.root_z0 = m.root_z0; c.gravity_z = m.gravity_z; c.solK = m.solK; c.solB = m.solB; c.meaninertia = m.meaninertia;
    c.tolerance = m.tolerance; c.ls_tolerance = m.ls_tolerance; c.ls_reltol = m.ls_reltol; c.tol_rel = m.tol_rel;
    c.nvf = T(m.nv); c.scale = T(1) / (m.meaninertia * c.nvf);
    if constexpr (PIN) {
        for (int k = 0; k < 5; k++) { c.solimp[k] = m.solimp[k]; g_pin(c.solimp[k]); }
        for (int k = 0; k < 3; k++) { c.solimp_inv[k] = m.solimp_inv[k]; g_pin(c.solimp_inv[k]); }
    }
    for (int t = 0; t < GD<TP>::NXA; t++) {
        c.xs_qpos0[t] = m.xs_qpos0[t]; c.xs_damping[t] = m.xs_damping[t]; c.xs_armature[t] = m.xs_armature[t];
        g_pin(c.xs_qpos0[t]); g_pin(c.xs_damping[t]); g_pin(c.xs_armature[t]);
    }
    c.iterations = m.iterations; c.ls_iterations = m.ls_iterations;
    g_pin(c.root_z0); g_pin(c.gravity_z); g_pin(c.solK); g_pin(c.solB); g_pin(c.meaninertia); g_pin(c.tolerance); g_pin(c.ls_tolerance);
    g_pin(c.ls_reltol); g_pin(c.tol_rel); g_pin(c.nvf); g_pin(c.scale); g_pin(c.iterations); g_pin(c.ls_iterations);
}

// build-defined dynamics randomisation of one walker (the reference's dynamics_randomization is a stub,
// drloco/mujoco/mimic_env.py:492-524; BASELINE config 5): scale of all body masses / inertias, sliding friction of
// the floor, world-frame push force at the torso's centre of mass ([3P] xfrc_applied)
template <typename T> struct GWalk { T mscale, floor_mu; V3<T> push; bool pushed; };

template <typename T, typename TP> struct GCtx {
    DL_LDS T* wb;                                // walker's LDS region
    const DL_CONST GModel<T, TP>* m;             // uniform scalars only on the hot path
    int j;                                       // lane in the row
    const GLane<T, GD<TP>::NPASS>* ln;           // this lane's preloaded model data
    const GConst<T, TP>* c;                      // pinned uniform scalars
    const GWalk<T>* wk;                          // this walker's randomisation
    DL_LDS T* mm;                                // mirror block of the mass matrix (GLds::MM inside the rows, or its own space in the split workgroup)
    DL_LDS T* mbox;                              // split workgroup: this walker's mailbox between the dynamics wave and the constraint wave (else null)
    DL_LDS T* mbox0;                             //                  the mailbox of the wave's first walker (sequence numbers of the wave pair)
    int32_t* fault;                              //                  the handle's fault word (or null) and the poll budget of this wave's waits
    int spin_limit;
    DL_LDS T* bfr;                               // body frames [MAXB][BFR_W] and root height as g_fk publishes them / the collision stage reads them: GLds::BFR and
    DL_LDS T* rz;                                //   GLds::MISC inside the walker's region, or (partner wave of a split workgroup) GSplit::BFRX / RZX, its own space
    int strict = 0;                              // dl_config.strict_solver (one-wave form): the Newton solver takes [3P] mj_solNewton's decisions instead of the product's shortcuts
};
// Split workgroup: a second wave builds the constraints of the same four walkers while the first runs the smooth dynamics.
// Per walker, behind the regular region: the mirror block of M (which may no longer share the rows' space) and the mailbox.
template <typename TP> struct GSplit {
    using Ld = GLds<TP>;
    static constexpr int MMX = Ld::TOTAL;                        // M mirror [16][MS]
    static constexpr int MB = MMX + GL * Ld::MS;                 // mailbox
    static constexpr int MB_Q = 0, MB_X0 = 16, MB_LIM = 32, MB_SGN = 48, MB_NCON = 64, MB_NLIM = 65, MB_CMDSEQ = 66, MB_CMD = 67, MB_DONESEQ = 68;      // (CMDSEQ, CMD): one aligned 8-byte word
    // replicated root translations (NX > 0): the request's words for them -- configuration (command 2), solver start point, NEXT configuration -- one 16-byte group each
    static constexpr int MB_QX = 96, MB_X0X = 100, MB_QNX = 104;
    static constexpr int MB_SIZE = GD<TP>::NX > 0 ? 112 : 96;
    // look-ahead of the mass matrix: MB_QN = the configuration of the NEXT evaluation (known when this one is requested); MB_MOK = sequence number of
    // the request whose mass matrix is in the mirror block (posted by the partner once it has checked that what it precomputed is for this request's
    // configuration); MB_MFREE = sequence number of the last request whose mass matrix the dynamics wave has taken into registers
    static constexpr int MB_MOK = 69, MB_MFREE = 70, MB_QN = 72, MB_PRE = 71;          // MB_PRE: sequence number of the last request whose look-ahead is complete (-1 at launch; 0: the state in memory)
    static_assert(MB_QN % 4 == 0 && MB_QN + GL <= MB_SIZE, "mailbox layout");
    // the two epoch counters through which the waves of a pair meet in k_rollout_pairs (pair_sync): words of walker 0's mailbox behind the announced configuration
    static constexpr int MB_PAIR0 = 90, MB_PAIR1 = 91;
    static_assert(MB_PAIR0 >= MB_QN + GL && MB_PAIR1 == MB_PAIR0 + 1 && MB_PAIR1 < MB_SIZE, "the pair counters lie behind every other mailbox field");
    // what the partner wave computes one evaluation ahead and hands over through LDS: the body frames (its own copy: GLds::BFR shares the space of the contact
    // Jacobians, which are live while it works) and, per dof lane, (joint axis x y z, root height)
    static constexpr int BFRX = MB + MB_SIZE;                    // body frames [MAXB][BFR_W]
    static constexpr int AXX = BFRX + ((GD<TP>::MAXB * Ld::BFR_W + 31) / 32) * 32;      // [16 lanes][4]: axis, root height
    static constexpr int RZX = AXX + 3;                          // the root height of lane 0's record
    // replicated root translations: per dof lane (M[j][t] for the NX translations, M[t][t] of translation 0 = the walker's mass + armature), the partner's g_mass_rows
    static constexpr int MXL = AXX + GL * 4;                     // [16 lanes][4]
    // the dynamics wave's lane file (19-dof walker): per dof lane (motor force of this control step, joint damping, -, -).  Values used once per evaluation that
    // the 256-register wave cannot hold across the solver: as registers they were spilled to scratch and reloaded one by one, each behind its own s_waitcnt
    // (five dependent memory round trips per evaluation); one ds_read_b128 instead.  The replicated dofs' damping sits in GLds::MISC + 4 .. 6.
    static constexpr int LSP = MXL + (GD<TP>::NX > 0 ? GL * 4 : 0);
    static constexpr bool LANE_FILE = GD<TP>::NX > 0;
    static constexpr int TOTAL_RAW = LSP + (LANE_FILE ? GL * 4 : 0);
    static constexpr int TOTAL = ((TOTAL_RAW - 16 + 31) / 32) * 32 + 16;          // per walker; 16 (mod 32) like GLds::TOTAL
    static_assert(TOTAL % 32 == 16 && MB % 4 == 0 && TOTAL >= TOTAL_RAW && GD<TP>::NX <= 3, "walker regions keep their bank offset");
    // polls (s_sleep 16: ~1000 cycles each, ~30 ms in all) before a wave gives up waiting for its partner: no hang on a protocol error -- the
    // wave sets the handle's fault word, its walkers take the reference's exception path (mimic_env.py:86-91) and the host raises DL_E_FAULT
#if defined(DL_GROUP_EMU)
    static constexpr int SPIN_LIMIT = 1 << 16;
#else
    static constexpr int SPIN_LIMIT = (1 << 20) / (DL_SLEEP_N > 0 ? DL_SLEEP_N : 1);          // the same ~30 ms of patience whatever the length of a poll's sleep
#endif
};

// ------------------------------------------------------------------------------------------
// Compile-time topology for the lane-per-dof layout (lane l <-> dof l + NX).  The dof order of a model is topological
// and mostly contiguous: a "run" is a maximal sequence ..., l-1, l with parent(l) == l-1 (straight walker: root + right
// leg = lanes 0..9, left leg = 10..13 hanging off lane 5; 19-dof walker: root rotations + lumbar = lanes 0..5, the legs
// = 6..10 and 11..15 hanging off lane 2).  Sums over a dof's ancestor chain / over its subtree are then segmented
// scans inside the runs (row_shr / row_shl DPP steps) plus one broadcast per run boundary.
template <typename TP> struct GTopoTab {
    uint32_t anc[GL];        // bit a: lane a is on the root -> l chain (incl. l)
    uint32_t bodies[GL];     // bit b: the dof of lane l moves body b
    int32_t rs[GL], re[GL];  // first / last lane of the run of l
    int32_t last[GL];        // l is the last dof of its body
};
template <typename TP> constexpr GTopoTab<TP> g_make_topo() {
    constexpr int NX = GD<TP>::NX, NL = GD<TP>::NL;
    GTopoTab<TP> t{};
    for (int j = 0; j < GL; j++) { t.anc[j] = 0; t.bodies[j] = 0; t.rs[j] = j; t.re[j] = j; t.last[j] = 0; }
    for (int j = 0; j < NL; j++) {
        for (int a = 0; a < NL; a++) if (TP::dof_anc(j + NX, a + NX)) t.anc[j] |= 1u << a;
        for (int b = 1; b < TP::NB; b++) if (TP::body_anc(b, j + NX)) t.bodies[j] |= 1u << b;
        int r = j; while (r > 0 && TP::dof_parent(r + NX) == r + NX - 1) r--;
        t.rs[j] = r;
        int e = j; while (e + 1 < NL && TP::dof_parent(e + 1 + NX) == e + NX) e++;
        t.re[j] = e;
        t.last[j] = (j == NL - 1 || TP::dof_body(j + 1 + NX) != TP::dof_body(j + NX)) ? 1 : 0;
    }
    return t;
}
template <typename TP> struct GTopo {
    static constexpr int NX = GD<TP>::NX, NL = GD<TP>::NL;
    static constexpr GTopoTab<TP> tab = g_make_topo<TP>();
    // per-lane views of the dof tables
    static constexpr int l_type(int l) { return TP::dof_type(l + NX); }
    static constexpr int l_axis(int l) { return TP::dof_axis(l + NX); }
    static constexpr int l_body(int l) { return TP::dof_body(l + NX); }
    static constexpr int l_parent(int l) { const int p = TP::dof_parent(l + NX) - NX; return p < 0 ? -1 : p; }
    static constexpr bool dof_first(int l) { return l + NX == 0 || TP::dof_body(l + NX - 1) != TP::dof_body(l + NX); }
    static constexpr bool run_start(int l) { return l == 0 || l_parent(l) != l - 1; }
    static constexpr int max_run() { int m = 1; for (int j = 0; j < NL; j++) { const int l = tab.re[j] - tab.rs[j] + 1; if (l > m) m = l; } return m; }
    // Twin branches (two legs): a run b.. that hangs off lane P and has the same shape (length, joint types / axes, body
    // boundaries) as an earlier sequence a.. of the same length below the same parent -- either the lanes P + 1.. that continue
    // P's own run (straight walker) or an earlier run that also hangs off P (19-dof walker).  Their hinges act on disjoint sets
    // of lanes, so the kinematics applies hinge a + i and its twin b + i in ONE rotation step (a lane takes the sine / cosine
    // of whichever of the two is on its chain).
    static constexpr int run_len(int a) { return tab.re[a] - tab.rs[a] + 1; }
    static constexpr bool same_shape(int a, int b, int L) {
        for (int i = 0; i < L; i++) {
            if (l_type(a + i) != 1 || l_type(b + i) != 1 || l_axis(a + i) != l_axis(b + i)) return false;
            if (dof_first(a + i) != dof_first(b + i)) return false;
        }
        return true;
    }
    static constexpr int twin_of_run(int b) {    // b: start of a run (> 0) -> first lane of its twin segment, or -1
        const int P = l_parent(b), L = run_len(b);
        if (P < 0) return -1;
        const int a = P + 1;                     // continuation of the parent's run
        if (a + L - 1 <= tab.re[P] && a + L - 1 < b && same_shape(a, b, L)) return a;
        for (int r = P + 1; r < b; r++)          // an earlier run below the same parent
            if (run_start(r) && l_parent(r) == P && run_len(r) == L && twin_of_run(r) < 0 && same_shape(r, b, L)) return r;
        return -1;
    }
    static constexpr bool is_follower(int d) {   // lane d is applied together with an earlier twin
        const int b = tab.rs[d];
        return b > 0 && run_start(b) && twin_of_run(b) >= 0;
    }
    static constexpr int partner(int d) {        // the lane applied in the same step as lane d, or -1
        for (int b = d + 1; b < NL; b++) {
            if (!run_start(b)) continue;
            const int a = twin_of_run(b);
            if (a >= 0 && d >= a && d < a + run_len(b)) return b + (d - a);
        }
        return -1;
    }
    static constexpr bool slides_first() {       // slide joints: root body only, world-aligned (before any hinge)
        bool hinge = false;
        for (int j = 0; j < TP::NV; j++) { if (TP::dof_type(j) == 1) hinge = true; else if (hinge || TP::dof_body(j) != 1) return false; }
        return true;
    }
    // Elimination order of the leaf-first factorisation (g_chol_rev): children before parents -- lanes by decreasing depth -- so that
    // the factor has no fill-in outside the tree pattern and the two legs are eliminated side by side (independent chains).
    static constexpr int depth(int l) { int d = 0; for (int p = l_parent(l); p >= 0; p = l_parent(p)) d++; return d; }
    struct Order { int at[GL]; };
    static constexpr Order make_order() {
        Order o{};
        int n = 0, maxd = 0;
        for (int l = 0; l < NL; l++) if (depth(l) > maxd) maxd = depth(l);
        for (int d = maxd; d >= 0; d--) for (int l = 0; l < NL; l++) if (depth(l) == d) o.at[n++] = l;
        for (; n < GL; n++) o.at[n] = 0;
        return o;
    }
    static constexpr Order order = make_order();
    static constexpr bool proper_anc(int k, int a) { return a != k && ((tab.anc[k] >> a) & 1u); }      // lane a is a proper ancestor of lane k
    static constexpr bool is_leaf(int k) { for (int l = 0; l < NL; l++) if (proper_anc(l, k)) return false; return true; }
    static constexpr bool rooted() {             // lane 0 is an ancestor of every lane (its subtree is the whole walker)
        for (int j = 0; j < NL; j++) if (!(tab.anc[j] & 1u)) return false;
        return true;
    }
};
// this lane's topology words (registers)
template <typename T> struct GLaneTopo {
    uint32_t anc, bodies;
    int rs;
    bool last;
    T ms[4], ns[4];          // 1/0: lane j - 2^k (j + 2^k) belongs to the same run
    T ancf[GL];              // 1/0: lane a is on the root -> j chain.  Float masks in VGPRs: the boolean form lives in SGPR pairs,
                             // which the step kernel has to spill (two v_readlane per use inside the evaluation)
};
template <typename T, typename TP> __device__ __forceinline__ void g_lane_topo(int j, GLaneTopo<T>& lt) {
    const auto& tb = GTopo<TP>::tab;
    lt.anc = tb.anc[j]; lt.bodies = tb.bodies[j]; lt.rs = tb.rs[j]; lt.last = tb.last[j] != 0;
    const int re = tb.re[j];
#pragma unroll
    for (int k = 0; k < 4; k++) { lt.ms[k] = (j - (1 << k) >= lt.rs) ? T(1) : T(0); lt.ns[k] = (j + (1 << k) <= re) ? T(1) : T(0); }
#pragma unroll
    for (int a = 0; a < GL; a++) { lt.ancf[a] = ((lt.anc >> a) & 1u) ? T(1) : T(0); g_pin(lt.ancf[a]); }
}
// x_j <- sum over the lanes a on the root -> j chain of x_a, for NVAL values at once (one v_fmac_f32_dpp per value
// and scan step)
template <typename T, typename TP, int NVAL> __device__ __forceinline__ void g_chain_sum_n(T (&x)[NVAL], int j, const GLaneTopo<T>& lt) {
    constexpr int MR = GTopo<TP>::max_run();
    g_dpp_ready_n<NVAL>(x);
#pragma unroll
    for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x111>(x[i], lt.ms[0]);
    if constexpr (MR > 2) {
        g_dpp_ready_n<NVAL>(x);
#pragma unroll
        for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x112>(x[i], lt.ms[1]);
    }
    if constexpr (MR > 4) {
        g_dpp_ready_n<NVAL>(x);
#pragma unroll
        for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x114>(x[i], lt.ms[2]);
    }
    if constexpr (MR > 8) {
        g_dpp_ready_n<NVAL>(x);
#pragma unroll
        for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x118>(x[i], lt.ms[3]);
    }
    static_for<GD<TP>::NL>([&](auto ri) {
        constexpr int r = ri.value;
        if constexpr (r > 0 && GTopo<TP>::run_start(r)) {
            constexpr int P = GTopo<TP>::l_parent(r);
            const T f = lt.rs == r ? T(1) : T(0);
            g_dpp_ready_n<NVAL>(x);
#pragma unroll
            for (int i = 0; i < NVAL; i++) fmac_bcast_self<P>(x[i], f);
        }
    });
}
// x_j <- sum over the lanes d of the subtree of j (j on the root -> d chain) of x_d
template <typename T, typename TP, int NVAL> __device__ __forceinline__ void g_subtree_sum_n(T (&x)[NVAL], int j, const GLaneTopo<T>& lt) {
    constexpr int MR = GTopo<TP>::max_run(), NL = GD<TP>::NL;
    g_dpp_ready_n<NVAL>(x);
#pragma unroll
    for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x101>(x[i], lt.ns[0]);
    if constexpr (MR > 2) {
        g_dpp_ready_n<NVAL>(x);
#pragma unroll
        for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x102>(x[i], lt.ns[1]);
    }
    if constexpr (MR > 4) {
        g_dpp_ready_n<NVAL>(x);
#pragma unroll
        for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x104>(x[i], lt.ns[2]);
    }
    if constexpr (MR > 8) {
        g_dpp_ready_n<NVAL>(x);
#pragma unroll
        for (int i = 0; i < NVAL; i++) fmac_dpp_self<0x108>(x[i], lt.ns[3]);
    }
    static_for<NL>([&](auto ri) {
        constexpr int r = NL - 1 - ri.value;               // deepest runs first
        if constexpr (r > 0 && GTopo<TP>::run_start(r)) {
            constexpr int P = GTopo<TP>::l_parent(r);
            constexpr uint32_t ancP = GTopo<TP>::tab.anc[P];
            const T f = ((ancP >> j) & 1u) ? T(1) : T(0);
            g_dpp_ready_n<NVAL>(x);
#pragma unroll
            for (int i = 0; i < NVAL; i++) fmac_bcast_self<r>(x[i], f);
        }
    });
}
// rotate the frame (X,Y,Z) about its own coordinate axis IDX (compile time) by the angle with (s, c)
template <int IDX, typename T> __device__ __forceinline__ void rot_axis_c(V3<T>& X, V3<T>& Y, V3<T>& Z, T s, T c) {
    if constexpr (IDX == 0) { const V3<T> A = c * Y + s * Z, B = c * Z - s * Y; Y = A; Z = B; }
    else if constexpr (IDX == 1) { const V3<T> A = c * Z + s * X, B = c * X - s * Z; Z = A; X = B; }
    else { const V3<T> A = c * X + s * Y, B = c * Y - s * X; X = A; Y = B; }
}

// what a dof lane keeps in registers after the kinematics
template <typename T> struct GKin { V3<T> X, Y, Z, pos, axis; T rootz; };

// [3P] mj_kinematics.  Lane j ends with the frame after its dof (= the frame of its body if that is the body's last
// dof), the origin of its body relative to the root origin, and its joint axis.  Every lane runs the SAME
// straight-line code over all hinges in dof order; hinges that are not on the lane's chain enter as the identity
// (s, c) = (0, 1), so no lane waits for another one and nothing goes through LDS except the body frames that the
// collision stage reads (BFR) and rootz (MISC[0]).  qx: the replicated root translations (only the vertical one
// matters: positions are relative to the root origin).
template <typename T, typename TP, bool PUBLISH = true, bool PIN = GD<TP>::PIN_ALL>      // PUBLISH = false: registers only
__device__ __forceinline__ void g_fk(const GCtx<T, TP>& g, const GLaneTopo<T>& lt, T q, const GX<T, GD<TP>::NX>& qx, GKin<T>& k) {
    static_assert(GTopo<TP>::slides_first(), "slide joints must be world-aligned root joints");
    using Ld = GLds<TP>;
    using TPL = GTopo<TP>;
    constexpr int NL = GD<TP>::NL, NX = GD<TP>::NX;
    DL_LDS T* wb = g.wb;
    const int j = g.j;
    const auto& ln = *g.ln;
    const T dq = (j < NL) ? q - ln.qpos0 : T(0);
    T s = T(0), c = T(1);
    if (j < NL && ln.type == 1) dl_sincos(ln.sign * dq, s, c);
    const T cm1 = c - T(1);
    V3<T> X = mk<T>(1, 0, 0), Y = mk<T>(0, 1, 0), Z = mk<T>(0, 0, 1), pos = mk<T>(0, 0, 0);
    T rootz = g.c->root_z0;
    // body offsets: pinned registers, or (19-dof walker) scalar loads of this evaluation -- the opaque pointer keeps them from
    // being hoisted out of the RK4 loops into registers the kernel does not have
    const DL_CONST GModel<T, TP>* mc = g.m;
    if constexpr (!PIN) DL_SPIN(mc);
    auto body_pos = [&](int b, int kk) -> T { if constexpr (PIN) return g.c->body_pos[b][kk]; else return mc->body_pos[b][kk]; };
    static_for<NX>([&](auto ti) {
        constexpr int t = ti.value;
        if constexpr (TP::dof_axis(t) == 2) rootz += T(TP::dof_sign(t)) * (qx.x[t] - g.c->xs_qpos0[t]);
    });
    static_for<NL>([&](auto ai) {
        constexpr int a = ai.value;
        if constexpr (TPL::is_follower(a)) return;   // applied together with its twin (the other leg)
        constexpr int p = TPL::partner(a);
        const T f = lt.ancf[a];                           // 1 where hinge / body a is on this lane's chain, else the identity
        if constexpr (TPL::dof_first(a) && TPL::l_body(a) != 1) {
            constexpr int b = TPL::l_body(a);
            T bx = f * body_pos(b, 0), by = f * body_pos(b, 1), bz = f * body_pos(b, 2);
            if constexpr (p >= 0) {
                constexpr int bp = TPL::l_body(p);
                const T fp = lt.ancf[p];
                bx += fp * body_pos(bp, 0); by += fp * body_pos(bp, 1); bz += fp * body_pos(bp, 2);
            }
            pos = pos + bx * X + by * Y + bz * Z;
        }
        if constexpr (TPL::l_type(a) == 1) {
            T sa = rbcast<a>(s) * f, ca = T(1) + rbcast<a>(cm1) * f;
            if constexpr (p >= 0) { const T fp = lt.ancf[p]; sa += rbcast<p>(s) * fp; ca += rbcast<p>(cm1) * fp; }
            rot_axis_c<TPL::l_axis(a)>(X, Y, Z, sa, ca);
        } else if constexpr (TPL::l_axis(a) == 2) {
            rootz += T(TP::dof_sign(a + NX)) * rbcast<a>(dq);
        }
    });
    const int idx = ln.axis;
    const V3<T> col = idx == 0 ? X : (idx == 1 ? Y : Z);
    k.X = X; k.Y = Y; k.Z = Z; k.pos = pos; k.axis = ln.sign * col; k.rootz = rootz;
    if constexpr (PUBLISH) {
        if (j < NL && lt.last) {
            DL_LDS T* f = g.bfr + Ld::BFR_W * ln.body;
            st4(f, X.x, X.y, X.z, Y.x); st4(f + 4, Y.y, Y.z, Z.x, Z.y); st4(f + 8, Z.z, pos.x, pos.y, pos.z);
        }
        if (j == 0) *g.rz = rootz;
        g_sync<T>();
    }
}

// height of the lowest foot-sole site above the floor at the configuration last passed to g_fk
// (reset_model's COM-z adjustment, drloco/mujoco/mimic_env.py:547-559); identical in the 16 lanes of the row
template <typename T, typename TP>
__device__ __forceinline__ T g_lowest_site(const GCtx<T, TP>& g) {
    using Ld = GLds<TP>;
    const DL_CONST GModel<T, TP>& m = *g.m;
    DL_LDS T* wb = g.wb;
    const int j = g.j;
    T low = T(1e30);
    if (j < m.nsite) {
        const int b = m.site_body[j];
        DL_LDS T* f = g.bfr + Ld::BFR_W * b;
        const T pz = f[11] + m.site_pos[j][0] * f[2] + m.site_pos[j][1] * f[5] + m.site_pos[j][2] * f[8];
        low = *g.rz + pz;
    }
    low = dl_min(low, dpp_f<0x128>(low));
    low = dl_min(low, dpp_f<0x124>(low));
    low = dl_min(low, dpp_f<0x122>(low));
    low = dl_min(low, dpp_f<0x121>(low));
    return low;
}

// the smooth dynamics of a walker as its lanes hold them
template <typename T, typename TP> struct GSmooth {
    static constexpr int NXA = GD<TP>::NXA;
    T mrow[GL];              // row j of M over the lane dofs (diagonal entry: see mcorr)
    T mdiag, mcorr;
    T smooth;                // qfrc_smooth of the lane's dof
    // replicated root translations (NX > 0): M[j][t] of this lane, M[t][t] (the translations are mutually orthogonal: M[t][t'] = 0), qfrc_smooth[t]
    T mxl[NXA], mxx[NXA], smoothx[NXA];
};

// [3P] mj_kinematics + mj_crb + mj_rne + passive/actuator forces, one dof per lane, everything in registers:
//   spatial quantities in world orientation about the root origin (parent <-> child transforms are the identity),
//   body twist / velocity-product acceleration = chain sums of the joint contributions (segmented scans),
//   spatial inertia + inertial wrench of a body on the lane of its last dof,
//   composite inertia / wrench = subtree sums, bias_j = S_j . W_j, M[j][a] = S_a . (Ic_j S_j) for a on the chain of j.
// The lower triangle of M is mirrored through LDS once so that every lane holds its full row (mrow).
// The replicated root translations are ancestors of every lane: their velocity enters every twist, their composite
// inertia / wrench is the whole walker's (the subtree of lane 0), M[j][t] = S_t . (Ic_j S_j) is one entry per lane.
// Out: sm; kinematics in k; body frames (BFR) and rootz (MISC[0]) in LDS.
// WITH_M = false (the dynamics wave of a split workgroup): the velocity-dependent half only -- bias, passive and applied forces (sm.smooth).  The
// configuration-dependent half, the mass matrix, is the partner wave's (g_mass_rows), computed one evaluation AHEAD: inside RK4 the configuration of the
// next stage depends on this stage's velocity only, which is known before this stage's constraint solve starts.
template <typename T, typename TP, bool PUBLISH = true, bool HAVE_KIN = false, bool WITH_M = true>      // HAVE_KIN: the caller has run g_fk already (k is an input)
__device__ __forceinline__ void g_smooth_dynamics(const GCtx<T, TP>& g, const GLaneTopo<T>& lt, T q, T v, T ctrl_force, const GX<T, GD<TP>::NX>& qx, const GX<T, GD<TP>::NX>& vx,
                                                  GKin<T>& k, GSmooth<T, TP>& sm) {
    using Ld = GLds<TP>;
    constexpr int NL = GD<TP>::NL, NX = GD<TP>::NX;
    DL_LDS T* wb = g.wb;
    const int j = g.j;
    const auto& ln = *g.ln;
    if constexpr (!HAVE_KIN) g_fk<T, TP, PUBLISH>(g, lt, q, qx, k);
    const bool isdof = j < NL;
    // motion subspace of dof j and its joint velocity contribution
    SV<T> S;
    {
        // (selects, not branches: an exec-masked region costs more scalar instructions than these six moves)
        const bool hinge = isdof && ln.type != 0, slide = isdof && ln.type == 0;
        const V3<T> pa = cross(k.pos, k.axis);
        S.w = mk<T>(hinge ? k.axis.x : T(0), hinge ? k.axis.y : T(0), hinge ? k.axis.z : T(0));
        S.v = mk<T>(hinge ? pa.x : (slide ? k.axis.x : T(0)), hinge ? pa.y : (slide ? k.axis.y : T(0)), hinge ? pa.z : (slide ? k.axis.z : T(0)));
    }
    const T qd = isdof ? v : T(0);
    const SV<T> vJ = {qd * S.w, qd * S.v};
    T sv[6] = {vJ.w.x, vJ.w.y, vJ.w.z, vJ.v.x, vJ.v.y, vJ.v.z};
    g_chain_sum_n<T, TP, 6>(sv, j, lt);
    SV<T> vel = {mk<T>(sv[0], sv[1], sv[2]), mk<T>(sv[3], sv[4], sv[5])};
    static_for<NX>([&](auto ti) { constexpr int t = ti.value; vcomp<TP::dof_axis(t)>(vel.v) += T(TP::dof_sign(t)) * vx.x[t]; });
    // velocity-product acceleration: sum over the chain of (twist of the parent) x (joint velocity); the twist of
    // the parent is vel - vJ and vJ x vJ = 0 (a translation of the root contributes nothing: its parent does not rotate)
    const SV<T> cJ = {cross(vel.w, vJ.w), cross(vel.w, vJ.v) + cross(vel.v, vJ.w)};
    T sa[6] = {cJ.w.x, cJ.w.y, cJ.w.z, cJ.v.x, cJ.v.y, cJ.v.z};
    g_chain_sum_n<T, TP, 6>(sa, j, lt);
    const SV<T> acc = {mk<T>(sa[0], sa[1], sa[2]), mk<T>(sa[3], sa[4], sa[5] - g.c->gravity_z)};
    // spatial inertia and inertial wrench of the body whose last dof this is (zero on the other lanes)
    SI<T> I;
    V3<T> com;                // centre of mass of the lane's body (relative to the root origin)
    {
        const T ms = (isdof && lt.last) ? g.wk->mscale : T(0);
        const T mass = ms * ln.mass, i0 = ms * ln.inertia[0], i1 = ms * ln.inertia[1], i2 = ms * ln.inertia[2];
        const V3<T>&X = k.X, &Y = k.Y, &Z = k.Z;
        const V3<T> c = k.pos + ln.ipos[0] * X + ln.ipos[1] * Y + ln.ipos[2] * Z;
        com = c;
        const T cc = dot(c, c);
        I.m = mass; I.h = mass * c;
        I.I.xx = i0 * X.x * X.x + i1 * Y.x * Y.x + i2 * Z.x * Z.x + mass * (cc - c.x * c.x);
        I.I.yy = i0 * X.y * X.y + i1 * Y.y * Y.y + i2 * Z.y * Z.y + mass * (cc - c.y * c.y);
        I.I.zz = i0 * X.z * X.z + i1 * Y.z * Y.z + i2 * Z.z * Z.z + mass * (cc - c.z * c.z);
        I.I.xy = i0 * X.x * X.y + i1 * Y.x * Y.y + i2 * Z.x * Z.y - mass * c.x * c.y;
        I.I.xz = i0 * X.x * X.z + i1 * Y.x * Y.z + i2 * Z.x * Z.z - mass * c.x * c.z;
        I.I.yz = i0 * X.y * X.z + i1 * Y.y * Y.z + i2 * Z.y * Z.z - mass * c.y * c.z;
    }
    SV<T> F;
    {
        const SV<T> Iv = si_mul(I, vel), Ia = si_mul(I, acc);
        F = {Ia.w + cross(vel.w, Iv.w) + cross(vel.v, Iv.v), Ia.v + cross(vel.w, Iv.v)};
    }
    // composite inertia and wrench of the subtree of dof j
    SI<T> Ic;
    SV<T> W;
    if constexpr (WITH_M) {
        T cs16[16] = {I.m, I.h.x, I.h.y, I.h.z, I.I.xx, I.I.xy, I.I.xz, I.I.yy, I.I.yz, I.I.zz, F.w.x, F.w.y, F.w.z, F.v.x, F.v.y, F.v.z};
        g_subtree_sum_n<T, TP, 16>(cs16, j, lt);
        Ic.m = cs16[0]; Ic.h = mk<T>(cs16[1], cs16[2], cs16[3]);
        Ic.I.xx = cs16[4]; Ic.I.xy = cs16[5]; Ic.I.xz = cs16[6]; Ic.I.yy = cs16[7]; Ic.I.yz = cs16[8]; Ic.I.zz = cs16[9];
        W = {mk<T>(cs16[10], cs16[11], cs16[12]), mk<T>(cs16[13], cs16[14], cs16[15])};
    } else {
        T cs6[6] = {F.w.x, F.w.y, F.w.z, F.v.x, F.v.y, F.v.z};
        g_subtree_sum_n<T, TP, 6>(cs6, j, lt);
        W = {mk<T>(cs6[0], cs6[1], cs6[2]), mk<T>(cs6[3], cs6[4], cs6[5])};
    }
    const T bias = sdot(S, W);
    if constexpr (WITH_M) {
    const SV<T> f = si_mul(Ic, S);
    // M[j][a] = S_a . (Ic_j S_j) for the dofs a on the chain of j (incl. j) -- zero elsewhere by the chain mask -- is what the
    // lane can compute.  The lane writes this part of its row as four 16-byte groups, no per-entry predicates; after the
    // exchange, own part + column j of the block (= the parts the descendants computed) is the full row.  The diagonal is
    // then counted twice and lacks the armature: mrow[j] is only ever used in products M x, where mcorr * x_j repairs it
    // (the factorisation takes the diagonal from mdiag).
    T Sr[6] = {S.w.x, S.w.y, S.w.z, S.v.x, S.v.y, S.v.z};
    g_dpp_ready_n<6>(Sr);
    T ml[GL];
#pragma unroll
    for (int a = NL; a < GL; a++) ml[a] = T(0);
    static_for<NL>([&](auto ai) {
        constexpr int a = ai.value;
        T mij = T(0);
        if constexpr (GTopo<TP>::l_type(a) == 1) { fmac_bcast<a, 1>(mij, Sr[0], f.w.x); fmac_bcast<a, 1>(mij, Sr[1], f.w.y); fmac_bcast<a, 1>(mij, Sr[2], f.w.z); }
        fmac_bcast<a, 1>(mij, Sr[3], f.v.x); fmac_bcast<a, 1>(mij, Sr[4], f.v.y); fmac_bcast<a, 1>(mij, Sr[5], f.v.z);
        ml[a] = mij * lt.ancf[a];
    });
    {
        const T mjj = sdot(S, f);
        sm.mdiag = isdof ? mjj + ln.armature : T(1);
        sm.mcorr = isdof ? ln.armature - mjj : T(0);
        DL_LDS T* row = g.mm + j * Ld::MS;
        st4(row, ml[0], ml[1], ml[2], ml[3]); st4(row + 4, ml[4], ml[5], ml[6], ml[7]);
        st4(row + 8, ml[8], ml[9], ml[10], ml[11]); st4(row + 12, ml[12], ml[13], ml[14], ml[15]);
    }
    g_sync<T>();
#pragma unroll
    for (int a = 0; a < GL; a++) sm.mrow[a] = (a < NL) ? ml[a] + g.mm[a * Ld::MS + j] : T(0);
    if constexpr (NX > 0) {
        static_assert(GTopo<TP>::rooted(), "lane 0 must be the root of the lane tree");
        const T mtot = rbcast<0>(Ic.m);
        static_for<NX>([&](auto ti) {
            constexpr int t = ti.value, ax = TP::dof_axis(t);
            sm.mxl[t] = isdof ? T(TP::dof_sign(t)) * vcomp<ax>(f.v) : T(0);
            sm.mxx[t] = mtot + g.c->xs_armature[t];
        });
    }
    }   // WITH_M
    // [3P] xfrc_applied on the torso (body 1): J^T of a world-frame force at its centre of mass
    T push_q = T(0);
    if (g.wk->pushed) {
        constexpr int RL = TP::body_last_dof(1) - NX;
        const V3<T> pc = mk<T>(rbcast<RL>(com.x), rbcast<RL>(com.y), rbcast<RL>(com.z));
        if ((lt.bodies >> 1) & 1u) push_q = dot(S.w, cross(pc, g.wk->push)) + dot(S.v, g.wk->push);
    }
    T damping = ln.damping, cforce = ctrl_force, xsd[GD<TP>::NXA];
    if constexpr (!WITH_M && GSplit<TP>::LANE_FILE) {          // the dynamics wave of the 19-dof walker's split workgroup: from its lane file in LDS (GSplit::LSP)
        const Q4<T> lf = ld4(wb + GSplit<TP>::LSP + 4 * j), xd = ld4(wb + Ld::MISC + 4);
        cforce = lf.a; damping = lf.b;
        const T xdv[3] = {xd.a, xd.b, xd.c};
        static_for<NX>([&](auto ti) { xsd[ti.value] = xdv[ti.value]; });
    } else {
        static_for<NX>([&](auto ti) { xsd[ti.value] = g.c->xs_damping[ti.value]; });
    }
    sm.smooth = isdof ? -damping * v - bias + cforce + push_q : T(0);
    if constexpr (NX > 0) {
        // total wrench of the walker = composite wrench of lane 0 (the total mass: above, with M)
        const V3<T> Wv = mk<T>(rbcast<0>(W.v.x), rbcast<0>(W.v.y), rbcast<0>(W.v.z));
        static_for<NX>([&](auto ti) {
            constexpr int t = ti.value, ax = TP::dof_axis(t);
            const T sg = T(TP::dof_sign(t));
            T fs = -xsd[t] * vx.x[t] - sg * vcomp<ax>(Wv);
            if (g.wk->pushed) fs += sg * vcomp<ax>(g.wk->push);
            sm.smoothx[t] = fs;
        });
    }
}

// The configuration-dependent half of the smooth dynamics for the lane-only walker: [3P] mj_crb -- composite inertias (subtree sums) and
// M[j][a] = S_a . (Ic_j S_j) -- from the kinematics `k` of the configuration.  Leaves the COMPLETE row j of M in g.mm (16-byte groups: M[j][0..15],
// then mdiag, mcorr as g_smooth_dynamics defines them), i.e. performs the mirror exchange itself.  Run by the partner wave of a split workgroup.
template <typename T, typename TP>
__device__ __forceinline__ void g_mass_rows(const GCtx<T, TP>& g, const GLaneTopo<T>& lt, const GKin<T>& k) {
    using Ld = GLds<TP>;
    constexpr int NL = GD<TP>::NL;
    static_assert(Ld::MS >= GL + 2, "two spare words per row of the mirror block");
    constexpr int NX = GD<TP>::NX;
    const int j = g.j;
    const auto& ln = *g.ln;
    const bool isdof = j < NL;
    SV<T> S;
    {
        const bool hinge = isdof && ln.type != 0, slide = isdof && ln.type == 0;
        const V3<T> pa = cross(k.pos, k.axis);
        S.w = mk<T>(hinge ? k.axis.x : T(0), hinge ? k.axis.y : T(0), hinge ? k.axis.z : T(0));
        S.v = mk<T>(hinge ? pa.x : (slide ? k.axis.x : T(0)), hinge ? pa.y : (slide ? k.axis.y : T(0)), hinge ? pa.z : (slide ? k.axis.z : T(0)));
    }
    SI<T> I;
    {
        const T ms = (isdof && lt.last) ? g.wk->mscale : T(0);
        const T mass = ms * ln.mass, i0 = ms * ln.inertia[0], i1 = ms * ln.inertia[1], i2 = ms * ln.inertia[2];
        const V3<T>&X = k.X, &Y = k.Y, &Z = k.Z;
        const V3<T> c = k.pos + ln.ipos[0] * X + ln.ipos[1] * Y + ln.ipos[2] * Z;
        const T cc = dot(c, c);
        I.m = mass; I.h = mass * c;
        I.I.xx = i0 * X.x * X.x + i1 * Y.x * Y.x + i2 * Z.x * Z.x + mass * (cc - c.x * c.x);
        I.I.yy = i0 * X.y * X.y + i1 * Y.y * Y.y + i2 * Z.y * Z.y + mass * (cc - c.y * c.y);
        I.I.zz = i0 * X.z * X.z + i1 * Y.z * Y.z + i2 * Z.z * Z.z + mass * (cc - c.z * c.z);
        I.I.xy = i0 * X.x * X.y + i1 * Y.x * Y.y + i2 * Z.x * Z.y - mass * c.x * c.y;
        I.I.xz = i0 * X.x * X.z + i1 * Y.x * Y.z + i2 * Z.x * Z.z - mass * c.x * c.z;
        I.I.yz = i0 * X.y * X.z + i1 * Y.y * Y.z + i2 * Z.y * Z.z - mass * c.y * c.z;
    }
    T cs10[10] = {I.m, I.h.x, I.h.y, I.h.z, I.I.xx, I.I.xy, I.I.xz, I.I.yy, I.I.yz, I.I.zz};
    g_subtree_sum_n<T, TP, 10>(cs10, j, lt);
    SI<T> Ic;
    Ic.m = cs10[0]; Ic.h = mk<T>(cs10[1], cs10[2], cs10[3]);
    Ic.I.xx = cs10[4]; Ic.I.xy = cs10[5]; Ic.I.xz = cs10[6]; Ic.I.yy = cs10[7]; Ic.I.yz = cs10[8]; Ic.I.zz = cs10[9];
    const SV<T> f = si_mul(Ic, S);
    T Sr[6] = {S.w.x, S.w.y, S.w.z, S.v.x, S.v.y, S.v.z};
    g_dpp_ready_n<6>(Sr);
    T ml[GL];
#pragma unroll
    for (int a = NL; a < GL; a++) ml[a] = T(0);
    static_for<NL>([&](auto ai) {
        constexpr int a = ai.value;
        T mij = T(0);
        if constexpr (GTopo<TP>::l_type(a) == 1) { fmac_bcast<a, 1>(mij, Sr[0], f.w.x); fmac_bcast<a, 1>(mij, Sr[1], f.w.y); fmac_bcast<a, 1>(mij, Sr[2], f.w.z); }
        fmac_bcast<a, 1>(mij, Sr[3], f.v.x); fmac_bcast<a, 1>(mij, Sr[4], f.v.y); fmac_bcast<a, 1>(mij, Sr[5], f.v.z);
        ml[a] = mij * lt.ancf[a];
    });
    const T mjj = sdot(S, f);
    DL_LDS T* row = g.mm + j * Ld::MS;
    st4(row, ml[0], ml[1], ml[2], ml[3]); st4(row + 4, ml[4], ml[5], ml[6], ml[7]);
    st4(row + 8, ml[8], ml[9], ml[10], ml[11]); st4(row + 12, ml[12], ml[13], ml[14], ml[15]);
    g_sync<T>();
    T mr[GL];
#pragma unroll
    for (int a = 0; a < GL; a++) mr[a] = (a < NL) ? ml[a] + g.mm[a * Ld::MS + j] : T(0);
    g_sync<T>();          // every lane has read its column before the rows are overwritten with the completed ones
    st4(row, mr[0], mr[1], mr[2], mr[3]); st4(row + 4, mr[4], mr[5], mr[6], mr[7]);
    st4(row + 8, mr[8], mr[9], mr[10], mr[11]); st4(row + 12, mr[12], mr[13], mr[14], mr[15]);
    st4(row + 16, isdof ? mjj + ln.armature : T(1), isdof ? ln.armature - mjj : T(0), T(0), T(0));
    if constexpr (NX > 0) {
        // replicated root translations (as g_smooth_dynamics<WITH_M> forms them): M[j][t] = S_t . (Ic_j S_j), M[t][t] = the walker's mass + armature
        static_assert(GTopo<TP>::rooted(), "lane 0 must be the root of the lane tree");
        const T mtot = rbcast<0>(Ic.m);
        T mx[4] = {T(0), T(0), T(0), mtot};
        static_for<NX>([&](auto ti) { constexpr int t = ti.value, ax = TP::dof_axis(t); mx[t] = isdof ? T(TP::dof_sign(t)) * vcomp<ax>(f.v) : T(0); });
        st4(g.wb + GSplit<TP>::MXL + 4 * j, mx[0], mx[1], mx[2], mx[3]);
    }
    g_sync<T>();
}

// 1/x: float = v_rcp_f32 + one Newton step; double = exact division
__device__ __forceinline__ float dl_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return r * fmaf(-x, r, 2.0f);
}
__device__ __forceinline__ double dl_rcp(double x) { return 1.0 / x; }
// [3P] solimp sigmoid (getimpedance) with the three reciprocals of the constants taken once (GModel::solimp_inv)
template <typename C, typename T> __device__ __forceinline__ T g_impedance(const C& m, T pos) {
    const T x = dl_abs(pos) * m.solimp_inv[0];
    // (both branches of the power-2 sigmoid and a select: lanes of one wave sit on either side of the midpoint, and an exec-masked
    //  region per side costs more than the three extra multiplies)
    const T ya = x * x * m.solimp_inv[1], yb = T(1) - (T(1) - x) * (T(1) - x) * m.solimp_inv[2];
    const T y = (m.solimp[4] == T(1)) ? x : ((x <= m.solimp[3]) ? ya : yb);
    const T imp = m.solimp[0] + y * (m.solimp[1] - m.solimp[0]);
    return x >= T(1) ? m.solimp[1] : (x <= T(0) ? m.solimp[0] : imp);
}

__device__ __forceinline__ float dl_sqrt_fast(float x) {
#if defined(DL_GROUP_EMU)
    return sqrtf(x);
#else
    return __builtin_amdgcn_sqrtf(x);
#endif
}
__device__ __forceinline__ double dl_sqrt_fast(double x) { return sqrt(x); }
// 1/sqrt(x): float = v_rsq_f32 + one Newton step (<= 1 ulp-ish, no denormal fix-ups); double = exact path
__device__ __forceinline__ float dl_rsqrt(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    return y * fmaf(-0.5f * x * y, y, 1.5f);
}
__device__ __forceinline__ double dl_rsqrt(double x) { return 1.0 / sqrt(x); }

// dense Cholesky of the symmetric N x N matrix whose FULL row j is held in lane j (h[a], a != j; the diagonal separately
// in hd).  Column k of the factor is gathered with row broadcasts (one DPP each).  The trailing update runs over the whole
// row, and lanes j <= k take no part in step k (their l_jk is 0), so afterwards lane j holds
//   lo[a], a < j: L[j][a]                       (row j of the factor; zero for a >= j), and
//   h[a], a > j:  L[a][j] * L[j][j]             (column j of the factor, unscaled: the symmetric trailing matrix at step j),
// invd = 1 / L[j][j].  Having both the row and the column in the lane makes BOTH triangular solves broadcast-type
// (one v_fmac_f32_dpp per step, no cross-lane reduction).  Pivots are floored (mju_cholFactor's mjMINVAL guard).
// pivot reciprocal square root: float = bare v_rsq_f32 (1 ulp; the factor only shapes the Newton direction and its 91
// float32 trailing updates round far more than that), double = exact path
__device__ __forceinline__ float dl_rsqrt_pivot(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double dl_rsqrt_pivot(double x) { return 1.0 / sqrt(x); }

// In: h = full row j of the matrix (h[j] unused), hd = its diagonal.  Out: lo[k] = L[j][k] for k < j and 0 otherwise;
// h[a] for a > j = L[a][j] * L[j][j]; invd.  A lane's pivot is final once the steps k < j are done (l_jk = 0 for k >= j),
// so invd is taken once at the end from the lane's own hd -- the same bits every lane used at step j.
template <typename T, int N> __device__ __forceinline__ void g_chol(T (&h)[GL], T (&lo)[GL], T hd, T& invd, int j, T floor_) {
#if DL_CHOL_SHORT_CHAIN
    // (see g_chol_rev: one-instruction pivot broadcast + floor, lane mask on h[k] instead of on the pivot's reciprocal root; lo[k] leaves
    // as the forward substitution's multiplier -L[j][k] / L[k][k])
    static_for<N>([&](auto kk) {
        constexpr int k = kk.value;
        const T inv = dl_rsqrt_pivot(max_bcast<k>(hd, floor_));
        const T hk = (j > k) ? h[k] : T(0);
        T lik = hk * inv;                                     // lanes j > k: L[j][k]
        lo[k] = -lik * inv;
        hd -= lik * lik;
        g_dpp_ready(lik);
        static_for<N - 1 - k>([&](auto aa) {
            constexpr int a = k + 1 + aa.value;
            fmac_bcast<a, -1>(h[a], lik, lik);                // lanes j > k, all columns a > k: h[a] -= L[a][k] L[j][k]
        });
    });
#else
    static_for<N>([&](auto kk) {
        constexpr int k = kk.value;
        const T inv = dl_rsqrt_pivot(dl_max(rbcast<k>(hd), floor_));
        T lik = h[k] * ((j > k) ? inv : T(0));                // lanes j > k: L[j][k]
        lo[k] = lik;
        hd -= lik * lik;
        g_dpp_ready(lik);
        static_for<N - 1 - k>([&](auto aa) {
            constexpr int a = k + 1 + aa.value;
            fmac_bcast<a, -1>(h[a], lik, lik);                // lanes j > k, all columns a > k: h[a] -= L[a][k] L[j][k]
        });
    });
#endif
    invd = dl_rsqrt_pivot(dl_max(hd, floor_));
}
// solve (L L^T) x = b with the factor as g_chol leaves it; b_j in, x_j out
template <typename T, int N> __device__ __forceinline__ T g_chol_solve(const T (&lo)[GL], const T (&up)[GL], T invd, T b, int j) {
#if DL_CHOL_SHORT_CHAIN
    // both substitutions as one self-referencing DPP multiply-add per step (see g_chol_solve_rev)
    T acc = b;
    static_for<N - 1>([&](auto kk) { constexpr int k = kk.value; fmac_bcast_chain<k>(acc, lo[k]); });
    const T s2 = -invd * invd;
    T u = acc * invd * invd;
    T up2[GL];
    static_for<N - 1>([&](auto kk) { constexpr int k = 1 + kk.value; up2[k] = (j < k) ? s2 * up[k] : T(0); });     // the lanes above row k hold column entries
    static_for<N - 1>([&](auto kk) { constexpr int k = N - 1 - kk.value; fmac_bcast_chain<k>(u, up2[k]); });
    return u;
#else
    // forward: y_k = (b_k - sum_{a<k} L[k][a] y_a) / L[k][k].  lo[k] is zero in the lanes j <= k, so a lane's accumulator
    // is final after step j - 1 and y_j is read off after the loop
    T acc = b;
    static_for<N - 1>([&](auto kk) {
        constexpr int k = kk.value;
        T yloc = acc * invd;                                  // y_k in lane k
        g_dpp_ready(yloc);
        fmac_bcast<k, -1>(acc, yloc, lo[k]);                  // lanes j > k: b_j - sum_{a<=k} L[j][a] y_a
    });
    const T yj = acc * invd;
    // backward: x_k = (y_k - sum_{i>k} L[i][k] x_i) / L[k][k] with L[i][k] = up_k[i] / L[k][k] from lane k's own registers:
    // t_k = sum_{i>k} up_k[i] x_i accumulates as the x_i become final (highest first)
    T x = T(0), t = T(0);
    static_for<N>([&](auto kk) {
        constexpr int k = N - 1 - kk.value;
        T xloc = (yj - invd * t) * invd;                      // final in lane k
        if (j == k) x = xloc;
        if constexpr (k > 0) {
            g_dpp_ready(xloc);
            fmac_bcast<k, 1>(t, xloc, up[k]);                 // lanes j < k (t of the other lanes is no longer read)
        }
    });
    return x;
#endif
}

// Leaf-first variant for tree-sparse matrices (H = M + J^T D J couples two dofs only if one is an ancestor of the other): eliminating
// children before parents (GTopo::order) produces no fill-in -- step k only touches the rows and columns of k's proper ancestors
// (75 trailing updates instead of 91 for the straight walker) -- and the two legs are independent chains until the root, so their pivot
// chains (broadcast -> max -> v_rsq -> multiply -> update, ~50 cycles each, the factorisation's critical path at one wave per SIMD)
// can run side by side (the order alternates between them).  Measured: +1 % on the benchmark line.  H = U U^T with U[a][k] != 0 only for a an ancestor of k.
// In: h = full row j (h[j] unused), hd = diagonal.  Out: lo[k] = U[j][k] for the descendants k of j (0 elsewhere); h[a] for the proper
// ancestors a of j = U[a][j] * U[j][j] (unscaled column j, frozen when j was eliminated); invd = 1 / U[j][j].
template <typename T, typename TP> __device__ __forceinline__ void g_chol_rev(T (&h)[GL], T (&lo)[GL], T hd, T& invd, int j, T floor_) {
    using TPL = GTopo<TP>;
    constexpr int N = GD<TP>::NL;
#if DL_OPT_LITERAL_MASKS == 2 && !defined(DL_GROUP_EMU)
    // the lane predicates (j < k) are formed HERE, per factorisation, from a lane index the compiler cannot trace back to the launch: one v_cmp into VCC per step instead of two
    // v_readlane of a 64-bit mask computed once per launch and spilled (thirteen masks = 26 SGPRs of a kernel that has none to spare)
    DL_VPIN(j);
#endif
#if DL_CHOL_SHORT_CHAIN
    // The pivots are one dependent chain through the whole factorisation (a dependent VALU instruction issues ~7 cycles after its producer
    // where independent ones issue every ~2.6): per step it is  max(bcast(hd), floor) [one DPP instruction] -> v_rsq -> multiply -> hd update;
    // the lane mask is applied to h[k] (ready long before the pivot) instead of to the pivot's reciprocal root.
    // Out (this form): lo[k] = -U[j][k] / U[k][k] -- the forward substitution's multiplier, scaled and negated here, off the chain.
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[ss.value];
        if constexpr (k > 0) {
            const T inv = dl_rsqrt_pivot(max_bcast<k>(hd, floor_));
#if DL_OPT_LITERAL_MASKS == 1
            const T hk = lane_keep<(1u << k) - 1u>(h[k], j);
#else
            const T hk = (j < k) ? h[k] : T(0);               // ancestors of k (unrelated lanes hold an exact zero; descendants their frozen column)
#endif
            T lik = hk * inv;
            lo[k] = -lik * inv;
            hd -= lik * lik;
            g_dpp_ready(lik);
            static_for<k>([&](auto aa) {
                constexpr int a = aa.value;
                if constexpr (TPL::proper_anc(k, a)) fmac_bcast<a, -1>(h[a], lik, lik);
            });
        }
    });
#else
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[ss.value];
        if constexpr (k > 0) {
            const T inv = dl_rsqrt_pivot(dl_max(rbcast<k>(hd), floor_));
            // ancestors have smaller indices than k; unrelated lanes hold an exact zero in h[k]; descendants (j > k) hold their frozen column
            T lik = h[k] * ((j < k) ? inv : T(0));
            lo[k] = lik;
            hd -= lik * lik;
            g_dpp_ready(lik);
            static_for<k>([&](auto aa) {
                constexpr int a = aa.value;
                if constexpr (TPL::proper_anc(k, a)) fmac_bcast<a, -1>(h[a], lik, lik);      // rows of the ancestors: h[a] -= U[a][k] U[j][k]
            });
        }
    });
#endif
    invd = dl_rsqrt_pivot(dl_max(hd, floor_));
}
// solve (U U^T) x = b with the factor as g_chol_rev leaves it; b_j in, x_j out
template <typename T, typename TP> __device__ __forceinline__ T g_chol_solve_rev(const T (&lo)[GL], const T (&up)[GL], T invd, T b, int j) {
    using TPL = GTopo<TP>;
    constexpr int N = GD<TP>::NL;
#if DL_CHOL_SHORT_CHAIN
    // Both substitutions as ONE self-referencing DPP multiply-add per step (destination = broadcast source = the running vector):
    //   U y = b, children first:  acc_j -= (U[j][k] / U[k][k]) acc_k  for the ancestors j of k; acc_k is final by then; y = acc / U[j][j] at the end;
    //   U^T x = y, parents first, on u_j = (y_j - sum over the proper ancestors a of U[a][j] x_a) / U[j][j], which IS x_j once j's ancestors are done:
    //   u_j -= x_k U[k][j] / U[j][j] = x_k * up_j[k] / U[j][j]^2 for the lanes j below k.
    T acc = b;
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[ss.value];
        if constexpr (k > 0) fmac_bcast_chain<k>(acc, lo[k]);
    });
    const T s2 = -invd * invd;
    T u = acc * invd * invd;
    // only the lanes below k take part in step k: the slot up[k] of lane k itself is not a matrix entry and those of the lanes above k are
    // stale rows (lanes of the other branch hold an exact zero) -- masked here, off the chain, so that u is x when the loop ends
    T up2[GL];
#if DL_OPT_LITERAL_MASKS == 1
    static_for<N>([&](auto kk) { constexpr int k = kk.value; if constexpr (!TPL::is_leaf(k)) up2[k] = lane_keep<0xffffu & ~((2u << k) - 1u)>(s2 * up[k], j); });
#else
#if DL_OPT_LITERAL_MASKS == 2 && !defined(DL_GROUP_EMU)
    DL_VPIN(j);
#endif
    static_for<N>([&](auto kk) { constexpr int k = kk.value; if constexpr (!TPL::is_leaf(k)) up2[k] = (j > k) ? s2 * up[k] : T(0); });
#endif
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[N - 1 - ss.value];
        if constexpr (!TPL::is_leaf(k)) fmac_bcast_chain<k>(u, up2[k]);
    });
    return u;
#else
    // U y = b, children first: y_k = (b_k - sum over the descendants d of k of U[k][d] y_d) / U[k][k]; lo[k] is zero in the lanes that are
    // not ancestors of k, so a lane's accumulator is final once its descendants are done
    T acc = b;
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[ss.value];
        if constexpr (k > 0) {
            T yloc = acc * invd;                                  // y_k in lane k
            g_dpp_ready(yloc);
            fmac_bcast<k, -1>(acc, yloc, lo[k]);
        }
    });
    const T yj = acc * invd;
    // U^T x = y, parents first: x_k = (y_k - sum over the proper ancestors a of k of U[a][k] x_a) / U[k][k], U[a][k] = up_k[a] / U[k][k]
    T x = T(0), t = T(0);
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[N - 1 - ss.value];
        T xloc = (yj - invd * t) * invd;                          // final in lane k
        if (j == k) x = xloc;
        if constexpr (!TPL::is_leaf(k)) {                         // leaves have nobody below them to tell
            g_dpp_ready(xloc);
            fmac_bcast<k, 1>(t, xloc, up[k]);                     // lanes below k in the tree (up[k] is zero elsewhere)
        }
    });
    return x;
#endif
}

// The replicated dofs are eliminated FIRST: with the Hessian ordered [replicated | lanes], L = [[Lxx, 0], [Lxl, Lll]].
// Lxx (NX x NX, uniform) and the pivots live in every lane, row j of Lxl in lane j; the trailing update leaves the lane block
// for g_chol.  hxx: lower triangle of the uniform block, hxl: this lane's coupling entries H[j][t].
template <typename T, int NXA> struct GCholX { T lxx[NXA][NXA], lxl[NXA], invd[NXA]; };
template <typename T, int NX, int NL> __device__ __forceinline__ void g_chol_x(T (&hxx)[NX > 0 ? NX : 1][NX > 0 ? NX : 1], T (&hxl)[NX > 0 ? NX : 1], T (&h)[GL], T& hd,
                                                                                  GCholX<T, (NX > 0 ? NX : 1)>& cx, T floor_) {
    static_for<NX>([&](auto kk) {
        constexpr int k = kk.value;
        const T inv = dl_rsqrt_pivot(dl_max(hxx[k][k], floor_));
        cx.invd[k] = inv;
        T lik = hxl[k] * inv;                                 // L[lane j][k]
        cx.lxl[k] = lik;
        static_for<NX - 1 - k>([&](auto tt) { constexpr int t = k + 1 + tt.value; cx.lxx[t][k] = hxx[t][k] * inv; });
        static_for<NX - 1 - k>([&](auto tt) {
            constexpr int t = k + 1 + tt.value;
            static_for<t - k>([&](auto uu) { constexpr int u = k + 1 + uu.value; hxx[t][u] -= cx.lxx[t][k] * cx.lxx[u][k]; });
            hxl[t] -= cx.lxx[t][k] * lik;
        });
        hd -= lik * lik;
        g_dpp_ready(lik);
        static_for<NL>([&](auto aa) { constexpr int a = aa.value; fmac_bcast<a, -1>(h[a], lik, lik); });   // h[a] -= L[a][k] L[j][k]
    });
}
// (L L^T) x = b for the full system: bx / xx are the replicated components (uniform), b / return value the lane's
template <typename T, int NX, int NL> __device__ __forceinline__ T g_chol_solve_x(const GCholX<T, (NX > 0 ? NX : 1)>& cx, const T (&lo)[GL], const T (&up)[GL], T invd, T b,
                                                                                   const T (&bx)[NX > 0 ? NX : 1], T (&xx)[NX > 0 ? NX : 1], int j) {
    if constexpr (NX == 0) return g_chol_solve<T, NL>(lo, up, invd, b, j);
    else {
        T yx[NX];
        static_for<NX>([&](auto kk) {
            constexpr int k = kk.value;
            T acc = bx[k];
            static_for<k>([&](auto tt) { acc -= cx.lxx[k][tt.value] * yx[tt.value]; });
            yx[k] = acc * cx.invd[k];
        });
        T bl = b;
        static_for<NX>([&](auto kk) { bl -= cx.lxl[kk.value] * yx[kk.value]; });
        const T x = g_chol_solve<T, NL>(lo, up, invd, bl, j);
        T s[NX];
        static_for<NX>([&](auto kk) { s[kk.value] = cx.lxl[kk.value] * x; });
        gsum_n<NX>(s);
        static_for<NX>([&](auto kk) {
            constexpr int k = NX - 1 - kk.value;
            T acc = yx[k] - s[k];
            static_for<NX - 1 - k>([&](auto tt) { constexpr int t = k + 1 + tt.value; acc -= cx.lxx[t][k] * xx[t]; });
            xx[k] = acc * cx.invd[k];
        });
        return x;
    }
}

// ---- the replicated dofs eliminated LAST (round 5).  Eliminated first (g_chol_x) they couple every pair of lanes: the lane block fills in and needs the dense
// factorisation -- sixteen pivots in one dependent chain.  Ordered [lanes leaf-first | replicated] the lane block keeps its tree pattern (g_chol_rev: no fill-in, the
// branches' pivot chains side by side) and the replicated dofs become a 3 x 3 Schur complement:
//   H_ll = U U^T (g_chol_rev);  W = U^-1 H_lx (the forward substitution for NX right-hand sides, row j of W in lane j);  S = H_xx - W^T W = Ls Ls^T (uniform).
// Solve: y = U^-1 b_l;  z_x = Ls^-1 (b_x - W^T y);  x_x = Ls^-T z_x;  x_l = U^-T (y - W x_x).
template <typename T, int NXA> struct GCholXL { T w[NXA], ls[NXA][NXA], invs[NXA]; };
template <typename T, typename TP> __device__ __forceinline__ void g_chol_x_last(const T (&lo)[GL], T invd, const T (&hxl)[GD<TP>::NXA], const T (&hxx)[GD<TP>::NXA][GD<TP>::NXA],
                                                                                  GCholXL<T, GD<TP>::NXA>& cx, T floor_) {
    using TPL = GTopo<TP>;
    constexpr int N = GD<TP>::NL, NX = GD<TP>::NX;
    static_assert(DL_CHOL_SHORT_CHAIN, "written for the multiplier form of g_chol_rev (lo[k] = -U[j][k] / U[k][k])");
    T acc[NX];
    static_for<NX>([&](auto ti) { acc[ti.value] = hxl[ti.value]; });
    static_for<N>([&](auto ss) {          // NX independent chains, interleaved
        constexpr int k = TPL::order.at[ss.value];
        if constexpr (k > 0) static_for<NX>([&](auto ti) { fmac_bcast_chain<k>(acc[ti.value], lo[k]); });
    });
    static_for<NX>([&](auto ti) { cx.w[ti.value] = acc[ti.value] * invd; });
    // S = H_xx - W^T W (lower triangle), summed over the lanes of the row
    constexpr int NS = NX * (NX + 1) / 2;
    T sp[NS];
    static_for<NX>([&](auto ti) { constexpr int t = ti.value; static_for<t + 1>([&](auto ui) { constexpr int u = ui.value; sp[t * (t + 1) / 2 + u] = cx.w[t] * cx.w[u]; }); });
    gsum_n<NS>(sp);
    T S[NX][NX];
    static_for<NX>([&](auto ti) { constexpr int t = ti.value; static_for<t + 1>([&](auto ui) { constexpr int u = ui.value; S[t][u] = hxx[t][u] - sp[t * (t + 1) / 2 + u]; }); });
    static_for<NX>([&](auto kk) {
        constexpr int k = kk.value;
        const T inv = dl_rsqrt_pivot(dl_max(S[k][k], floor_));
        cx.invs[k] = inv;
        static_for<NX - 1 - k>([&](auto tt) { constexpr int t = k + 1 + tt.value; cx.ls[t][k] = S[t][k] * inv; });
        static_for<NX - 1 - k>([&](auto tt) {
            constexpr int t = k + 1 + tt.value;
            static_for<t - k>([&](auto uu) { constexpr int u = k + 1 + uu.value; S[t][u] -= cx.ls[t][k] * cx.ls[u][k]; });
        });
    });
}
template <typename T, typename TP> __device__ __forceinline__ T g_chol_solve_rev_x(const T (&lo)[GL], const T (&up)[GL], T invd, const GCholXL<T, GD<TP>::NXA>& cx, T b,
                                                                                    const T (&bx)[GD<TP>::NXA], T (&xx)[GD<TP>::NXA], int j) {
    using TPL = GTopo<TP>;
    constexpr int N = GD<TP>::NL, NX = GD<TP>::NX;
    T acc = b;
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[ss.value];
        if constexpr (k > 0) fmac_bcast_chain<k>(acc, lo[k]);
    });
    const T y = acc * invd;
    T r[NX];
    static_for<NX>([&](auto ti) { r[ti.value] = cx.w[ti.value] * y; });
    gsum_n<NX>(r);
    T zx[NX];
    static_for<NX>([&](auto kk) {
        constexpr int k = kk.value;
        T a = bx[k] - r[k];
        static_for<k>([&](auto tt) { a -= cx.ls[k][tt.value] * zx[tt.value]; });
        zx[k] = a * cx.invs[k];
    });
    static_for<NX>([&](auto kk) {
        constexpr int k = NX - 1 - kk.value;
        T a = zx[k];
        static_for<NX - 1 - k>([&](auto tt) { constexpr int t = k + 1 + tt.value; a -= cx.ls[t][k] * xx[t]; });
        xx[k] = a * cx.invs[k];
    });
    T yl = y;
    static_for<NX>([&](auto ti) { yl -= cx.w[ti.value] * xx[ti.value]; });
    const T s2 = -invd * invd;
    T u = yl * invd;
    T up2[GL];
#if DL_OPT_LITERAL_MASKS == 1
    static_for<N>([&](auto kk) { constexpr int k = kk.value; if constexpr (!TPL::is_leaf(k)) up2[k] = lane_keep<0xffffu & ~((2u << k) - 1u)>(s2 * up[k], j); });
#else
#if DL_OPT_LITERAL_MASKS == 2 && !defined(DL_GROUP_EMU)
    DL_VPIN(j);
#endif
    static_for<N>([&](auto kk) { constexpr int k = kk.value; if constexpr (!TPL::is_leaf(k)) up2[k] = (j > k) ? s2 * up[k] : T(0); });
#endif
    static_for<N>([&](auto ss) {
        constexpr int k = TPL::order.at[N - 1 - ss.value];
        if constexpr (!TPL::is_leaf(k)) fmac_bcast_chain<k>(u, up2[k]);
    });
    return u;
}

// contact-frame Jacobian of the replicated root translations: the columns are the contact frame itself (normal = floor
// normal z, tangent 1 = (tx, ty, 0), tangent 2 = (-ty, tx, 0)), whatever body the contact is on.
// (dn, d1, d2) += J_x xs
template <typename T, typename TP> __device__ __forceinline__ void g_slide_jx(T tx, T ty, const T (&xs)[GD<TP>::NXA], T& dn, T& d1, T& d2) {
    static_for<GD<TP>::NX>([&](auto ti) {
        constexpr int t = ti.value, ax = TP::dof_axis(t);
        const T s = T(TP::dof_sign(t)) * xs[t];
        if constexpr (ax == 0) { d1 += tx * s; d2 -= ty * s; }
        else if constexpr (ax == 1) { d1 += ty * s; d2 += tx * s; }
        else dn += s;
    });
}
// column t of J_x: (jn, j1, j2)
template <typename T, typename TP, int t> __device__ __forceinline__ void g_slide_col(T tx, T ty, T& jn, T& j1, T& j2) {
    constexpr int ax = TP::dof_axis(t);
    const T s = T(TP::dof_sign(t));
    if constexpr (ax == 0) { jn = T(0); j1 = s * tx; j2 = -s * ty; }
    else if constexpr (ax == 1) { jn = T(0); j1 = s * ty; j2 = s * tx; }
    else { jn = s; j1 = T(0); j2 = T(0); }
}

template <typename TP> using GCandMask = std::conditional_t<(GD<TP>::NPASS > 2), uint64_t, uint32_t>;
__device__ __forceinline__ int g_popc(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int g_popc(uint64_t x) { return __popcll(x); }

// lane j's column (normal, tangent 1, tangent 2) of contact c's Jacobian: its own record, or -- packed layout -- the record at the lane's depth if the lane owns it
template <typename T, typename TP> __device__ __forceinline__ Q4<T> g_jc_load(const DL_LDS T* wb, int c, int j, int depth) {
    using Ld = GLds<TP>;
    if constexpr (!GD<TP>::JC_PACKED) return ld4(wb + Ld::JC + (c * GL + j) * 4);
    else {
        const Q4<T> r = ld4(wb + Ld::JC + (c * Ld::JCL + depth) * 4);
        const T m = r.d == T(j) ? T(1) : T(0);
        return {m * r.a, m * r.b, m * r.c, r.d};
    }
}
template <typename T> __device__ __forceinline__ int g_lane_depth(const GLaneTopo<T>& lt) { const int d = __popc(lt.anc) - 1; return d < 0 ? 0 : d; }

// ------------------------------------------------------------------------------------------
// [3P] mj_collision + position part of mj_makeConstraint for one walker (all 16 lanes).
// Returns (nlim, ncon) identical in every lane of the row.  `grp` = row index inside the wave.
// x0 / x0x = B v + a (lane / replicated dofs): the start point of the solver, folded into the rows' J a - aref.
// Second half of the constraint stage: the contact Jacobians and J x0 for the ncon contacts whose records are in LDS (needs the lane's
// kinematics in registers, nothing else of the first half).
template <typename T, typename TP>
__device__ __forceinline__ void g_contact_jacobians(const GCtx<T, TP>& g, const GLaneTopo<T>& lt, const GKin<T>& kin, int ncon, T x0, const T (&x0x)[GD<TP>::NXA]) {
    using Ld = GLds<TP>;
    constexpr int NX = GD<TP>::NX, MAXROW = Ld::MAXROW;
    DL_LDS T* wb = g.wb;
#if !defined(DL_GROUP_EMU)
    if constexpr (!GD<TP>::PIN_ALL) {          // (opaque: the loop's per-lane base addresses are otherwise formed once per control step and spilled -- four scratch reloads per evaluation)
        uint32_t wbo = (uint32_t)(uintptr_t)wb;
        DL_VPIN(wbo);
        wb = (DL_LDS T*)(uintptr_t)wbo;
    }
#endif
    int j = g.j;
    const auto& ln = *g.ln;
    int jdepth = g_lane_depth(lt);
#if !defined(DL_GROUP_EMU)
    if constexpr (!GD<TP>::PIN_ALL) { DL_VPIN(j); DL_VPIN(jdepth); }          // (opaque as well, round 6: the loop's three per-lane LDS bases -- wb + f(j), wb + f(depth) -- were still formed once per launch and reloaded from scratch in every evaluation)
#endif
    // ---- contact-frame Jacobians, dof-lane major: lane a writes its own column (normal, tangent 1, tangent 2) of
    // every contact from its joint axis / anchor in registers (dofs that do not move the contact's body write zeros),
    // two contacts per trip; the same trip adds J x0 to the contacts' rows (six interleaved row sums, expanded to the
    // pyramid rows by lanes 0..7), so the Jacobian is not read back for the start point
#if DL_OPT_JAC_PIPE
    // the records of a pair are requested one trip ahead -- those of the first pair before the contact count is looked at (the slots exist
    // whatever they hold): a trip is one LDS round trip + six row sums of dependent instructions, with nothing else to issue meanwhile
    Q4<T> A0 = ld4(wb + Ld::CON), B0 = ld4(wb + Ld::CON + 4), A1 = ld4(wb + Ld::CON + Ld::CON_W), B1 = ld4(wb + Ld::CON + Ld::CON_W + 4);
    T base = wb[Ld::ROW + Ld::R_JAREF * MAXROW + (j < 8 ? j : 0)];
#endif
    for (int c = 0; c < ncon; c += 2) {
        DL_LDS T* rja = wb + Ld::ROW + Ld::R_JAREF * MAXROW + 4 * c + j;
        const bool writer = j < 4 || (j < 8 && c + 1 < ncon);
#if DL_OPT_JAC_PIPE
        const Q4<T> A0n = A0, B0n = B0, A1n = A1, B1n = B1;
        const T basen = base;
        if (c + 2 < ncon) {
            const DL_LDS T* cn = wb + Ld::CON + Ld::CON_W * (c + 2);
            A0 = ld4(cn); B0 = ld4(cn + 4); A1 = ld4(cn + Ld::CON_W); B1 = ld4(cn + Ld::CON_W + 4);
            base = wb[Ld::ROW + Ld::R_JAREF * MAXROW + 4 * (c + 2) + (j < 8 ? j : 0)];
        }
#else
        const DL_LDS T* cn = wb + Ld::CON + Ld::CON_W * c;
        const Q4<T> A0n = ld4(cn), B0n = ld4(cn + 4), A1n = ld4(cn + Ld::CON_W), B1n = ld4(cn + Ld::CON_W + 4);
        const T basen = writer ? *rja : T(0);
#endif
        V3<T> w0 = ln.type == 0 ? kin.axis : cross(kin.axis, mk<T>(A0n.a, A0n.b, A0n.c) - kin.pos);
        V3<T> w1 = ln.type == 0 ? kin.axis : cross(kin.axis, mk<T>(A1n.a, A1n.b, A1n.c) - kin.pos);
        if (!((lt.bodies >> (int)A0n.d) & 1u)) w0 = mk<T>(0, 0, 0);
        if (!((lt.bodies >> (int)A1n.d) & 1u)) w1 = mk<T>(0, 0, 0);
        const T j0n = w0.z, j0a = B0n.a * w0.x + B0n.b * w0.y, j0b = -B0n.b * w0.x + B0n.a * w0.y;
        const T j1n = w1.z, j1a = B1n.a * w1.x + B1n.b * w1.y, j1b = -B1n.b * w1.x + B1n.a * w1.y;
        if constexpr (GD<TP>::JC_PACKED) {
            // every slot of the two contacts is cleared (owner -1) by the lane of its number, then the lanes on a contact's chain write their records at their depth:
            // LDS operations of a wave are performed in order, the later store stands
            if (j < Ld::JCL) { st4(wb + Ld::JC + (c * Ld::JCL + j) * 4, T(0), T(0), T(0), T(-1)); st4(wb + Ld::JC + ((c + 1) * Ld::JCL + j) * 4, T(0), T(0), T(0), T(-1)); }
            if ((lt.bodies >> (int)A0n.d) & 1u) st4(wb + Ld::JC + (c * Ld::JCL + jdepth) * 4, j0n, j0a, j0b, T(j));
            if ((lt.bodies >> (int)A1n.d) & 1u) st4(wb + Ld::JC + ((c + 1) * Ld::JCL + jdepth) * 4, j1n, j1a, j1b, T(j));
        } else {
        st4(wb + Ld::JC + (c * GL + j) * 4, j0n, j0a, j0b, T(0));
        st4(wb + Ld::JC + ((c + 1) * GL + j) * 4, j1n, j1a, j1b, T(0));
        }
        T r[6] = {j0n * x0, j0a * x0, j0b * x0, j1n * x0, j1a * x0, j1b * x0};
        gsum_n<6>(r);
        if constexpr (NX > 0) { g_slide_jx<T, TP>(B0n.a, B0n.b, x0x, r[0], r[1], r[2]); g_slide_jx<T, TP>(B1n.a, B1n.b, x0x, r[3], r[4], r[5]); }
        if (writer) {
            const bool second = j >= 4;
            const T vn = second ? r[3] : r[0], v1 = second ? r[4] : r[1], v2 = second ? r[5] : r[2], mu = second ? B1n.c : B0n.c;
            *rja = basen + vn + ((j & 1) ? -mu : mu) * ((j & 2) ? v2 : v1);
        }
    }
    g_sync<T>();
}

// ---- first half of the constraint stage in two parts: DETECT (a function of the configuration alone: which limits and candidates are active, where, with
// which impedance / regulariser; results in the lane's registers) and COMMIT (records and rows into LDS; the limit rows take the solver's start point).
// The partner wave of a split workgroup detects one evaluation ahead and commits when the evaluation is requested.
template <typename T, typename TP> struct GDet {
    static constexpr int NPASS = GD<TP>::NPASS;
    bool act[NPASS], lim;
    int cinf[NPASS], slot[NPASS], nlim, ncon, my_lim;
    T cpx[NPASS], cpy[NPASS], cpz[NPASS];
    T cdist[NPASS], ctx[NPASS], cty[NPASS], r_mu[NPASS], r_D[NPASS], r_kd[NPASS], lim_sign, l_D, l_k0;
};
template <typename T, typename TP, bool PIN = GD<TP>::PIN_ALL>
__device__ __forceinline__ void g_detect_constraints(const GCtx<T, TP>& g, const GLaneTopo<T>& lt, int grp, T q, GDet<T, TP>& d) {
    using Ld = GLds<TP>;
    using CM = GCandMask<TP>;
    constexpr int NL = GD<TP>::NL, NPASS = GD<TP>::NPASS;
    const int j = g.j;
    const auto& ln = *g.ln;
    // collision candidates and solimp: pinned registers, or (19-dof walker) loads of this evaluation from the model block
    GLaneCand<T, NPASS> cfetch;
    const DL_CONST GModel<T, TP>* mc = g.m;
    if constexpr (!PIN) { int jj = j; DL_VPIN(jj); DL_SPIN(mc); cfetch = mc->lanes[jj].cand; }
    const GLaneCand<T, NPASS>& cd = PIN ? ln.cand : cfetch;
    struct { T solimp[5], solimp_inv[3]; } simp;
    for (int k = 0; k < 5; k++) simp.solimp[k] = PIN ? g.c->solimp[k] : mc->solimp[k];
    for (int k = 0; k < 3; k++) simp.solimp_inv[k] = PIN ? g.c->solimp_inv[k] : mc->solimp_inv[k];
    const T rootz = *g.rz;
    // ---- joint limits (dof lanes), ranked by dof order through a ballot
    bool lim = false, lim_lo = false;
    T lim_dist = T(0);
    {
        const bool has = j < NL && ln.limited;
        const T dlo = q - ln.range_lo, dhi = ln.range_hi - q;
        lim_lo = has && dlo < T(0);
        lim = has && (dlo < T(0) || dhi < T(0));
        lim_dist = lim ? (lim_lo ? dlo : dhi) : T(0);
    }
    const uint32_t lmask = (uint32_t)((__ballot(lim) >> (GL * grp)) & 0xFFFFull);
    d.nlim = __popc(lmask);
    d.lim = lim; d.lim_sign = lim_lo ? T(1) : T(-1);
    // ---- contact candidates: NPASS passes of 16 (capsule ends and box corners in geom order); a candidate is a
    // constant body-local point (GLane), so the test is one frame transform + the floor distance
    static_for<NPASS>([&](auto pass_) {
        constexpr int pass = pass_.value;
        const int cinfo = cd.cinfo[pass];
        d.cinf[pass] = cinfo;
        const int b = (cinfo >> 5) & 15;
        DL_LDS T* f = g.bfr + Ld::BFR_W * b;
        const Q4<T> f0 = ld4(f), f1 = ld4(f + 4), f2 = ld4(f + 8);
        const V3<T> X = mk<T>(f0.a, f0.b, f0.c), Y = mk<T>(f0.d, f1.a, f1.b), Z = mk<T>(f1.c, f1.d, f2.a), pos = mk<T>(f2.b, f2.c, f2.d);
        const V3<T> pt = pos + cd.cpl[pass][0] * X + cd.cpl[pass][1] * Y + cd.cpl[pass][2] * Z;
        const T relz = cd.crl[pass][0] * X.z + cd.crl[pass][1] * Y.z + cd.crl[pass][2] * Z.z;
        T tx = cd.cal[pass][0] * X.x + cd.cal[pass][1] * Y.x + cd.cal[pass][2] * Z.x;
        T ty = cd.cal[pass][0] * X.y + cd.cal[pass][1] * Y.y + cd.cal[pass][2] * Z.y;
        const T n2 = tx * tx + ty * ty;
        const bool box = (cinfo >> 1) & 1;
        if (n2 < T(1e-30)) { tx = box ? T(0) : T(1); ty = box ? T(1) : T(0); } else { const T inv = dl_rsqrt(n2); tx *= inv; ty *= inv; }
        const T rad = cd.crad[pass];
        const T dist = rootz + pt.z - rad;
        d.act[pass] = (cinfo & 1) && dist < T(0) && !(relz > T(0));
        d.cpx[pass] = pt.x; d.cpy[pass] = pt.y; d.cpz[pass] = pt.z - (rad + T(0.5) * dist);
        d.cdist[pass] = dist; d.ctx[pass] = tx; d.cty[pass] = ty;
    });
    // box rule: only the first four qualifying corners of a box make contacts (mjc_PlaneBox)
    auto cand_mask = [&]() {
        CM m = 0;
        static_for<NPASS>([&](auto pass_) { constexpr int pass = pass_.value; m |= (CM)((__ballot(d.act[pass]) >> (GL * grp)) & 0xFFFFull) << (GL * pass); });
        return m;
    };
    CM cm = cand_mask();
    static_for<NPASS>([&](auto pass_) {
        constexpr int pass = pass_.value;
        const int c = j + GL * pass;
        if (d.act[pass] && ((d.cinf[pass] >> 1) & 1)) {
            const int first = c - ((d.cinf[pass] >> 2) & 7);                             // first corner of this box in the candidate list
            const CM before = cm & (((CM)1 << c) - (CM)1) & ~(((CM)1 << first) - (CM)1);
            if (g_popc(before) >= 4) d.act[pass] = false;
        }
    });
    cm = cand_mask();
    d.ncon = g_popc(cm);
    static_for<NPASS>([&](auto pass_) { constexpr int pass = pass_.value; d.slot[pass] = g_popc((CM)(cm & (((CM)1 << (j + GL * pass)) - (CM)1))); });
    d.my_lim = lim ? 4 * d.ncon + __popc(lmask & ((1u << j) - 1u)) : -1;
    // (the impedance / regulariser chains -- two reciprocals each -- of all passes and of the limit row are computed side by side, for every lane,
    //  and pinned: behind their lanes' predicates they ran one after the other, ~300 cycles of dependent instructions each)
    static_for<NPASS>([&](auto pass_) {
        constexpr int pass = pass_.value;
        const T mu = dl_max(cd.cmu[pass], g.wk->floor_mu), dist = d.cdist[pass];
        const T imp = g_impedance(simp, dist);
        const T diag = cd.cinvw[pass] * (T(1) + mu * mu);
        const T R = T(2) * mu * mu * dl_max(T(1e-15), (T(1) - imp) * diag * dl_rcp(imp));
        d.r_mu[pass] = mu; d.r_D[pass] = dl_rcp(R); d.r_kd[pass] = g.c->solK * imp * dist;
    });
    {
        const T imp = g_impedance(simp, lim_dist);
        const T R = dl_max(T(1e-15), (T(1) - imp) * ln.invw * dl_rcp(imp));
        d.l_D = dl_rcp(R); d.l_k0 = g.c->solK * imp * lim_dist;
    }
    static_for<NPASS>([&](auto pass_) { constexpr int pass = pass_.value; g_pin(d.r_D[pass]); g_pin(d.r_kd[pass]); g_pin(d.ctx[pass]); g_pin(d.cty[pass]); });
    g_pin(d.l_D); g_pin(d.l_k0);
}
// x0: the solver's start point B v + a of the lane's dof, folded into the limit rows' J a - aref
template <typename T, typename TP>
__device__ __forceinline__ void g_commit_constraints(const GCtx<T, TP>& g, const GDet<T, TP>& d, T x0) {
    using Ld = GLds<TP>;
    constexpr int NX = GD<TP>::NX, NPASS = GD<TP>::NPASS, MAXROW = Ld::MAXROW, MAXCON = Ld::MAXCON;
    DL_LDS T* wb = g.wb;
    const int j = g.j;
    // (BFR is aliased with JC, which is first written after the next g_sync: the frame reads of the detection are complete by then;
    //  the rows written below overwrite the mirror block of the mass matrix, whose reads precede them in program order)
    // ---- the lane of a candidate writes the contact record AND the contact's rows (rows 4c..4c+3: D, K imp r, cleared
    // active flags): nothing about a contact waits for another lane
    static_for<NPASS>([&](auto pass_) {
        constexpr int pass = pass_.value;
        if (d.act[pass]) {
            const int slot = d.slot[pass];
            const T mu = d.r_mu[pass], D = d.r_D[pass], kd = d.r_kd[pass];
            // (each value passes through an opaque statement on its way into a 16-byte store: the optimiser otherwise turns the scalar reads of neighbouring
            //  members of `d` into vector loads of the struct itself, which then has to live in private memory -- across the partner wave's wait, as scratch)
            T px = d.cpx[pass], py = d.cpy[pass], pz = d.cpz[pass], tx = d.ctx[pass], ty = d.cty[pass], dist = d.cdist[pass];
            g_pin(px); g_pin(py); g_pin(pz); g_pin(tx); g_pin(ty); g_pin(dist);
            DL_LDS T* cn = wb + Ld::CON + Ld::CON_W * slot;
            st4(cn, px, py, pz, T((d.cinf[pass] >> 5) & 15));
            st4(cn + 4, tx, ty, mu, dist);
            st4(wb + Ld::ROW + Ld::R_D * MAXROW + 4 * slot, D, D, D, D);
            st4(wb + Ld::ROW + Ld::R_JAREF * MAXROW + 4 * slot, kd, kd, kd, kd);
            T zero = T(0);
            g_pin(zero);          // (formed here: the constant quad is otherwise hoisted out of a caller's step loop, four registers that end up in scratch)
            st4(wb + Ld::ROW + Ld::R_TMP * MAXROW + 4 * slot, zero, T(0), T(0), T(0));
        }
    });
    // contacts are processed in pairs: a neutral record (world body: no dof moves it; mu = 0) closes an odd count
    if (j == 0 && d.ncon < MAXCON) {
        DL_LDS T* cn = wb + Ld::CON + Ld::CON_W * d.ncon;
        T one = T(1);
        g_pin(one);               // (a constant quad {1, 0, 0, 0} is hoisted out of every enclosing loop as four registers and then spilled: formed here)
        st4(cn, T(0), T(0), T(0), T(0)); st4(cn + 4, one, T(0), T(0), T(0));
        st4(wb + Ld::FC + Ld::FC_W * d.ncon, T(0), T(0), T(0), T(0));
        if constexpr (NX > 0) st4(wb + Ld::FC + Ld::FC_W * d.ncon + 8, T(0), T(0), T(0), T(0));
    }
    // limit rows follow in dof order; jar = J a - aref at the start point x0 = B v + a:  K imp r + (+-x0_j)
    if (d.lim) {
        const int r = d.my_lim;
        wb[Ld::ROW + Ld::R_D * MAXROW + r] = d.l_D;
        wb[Ld::ROW + Ld::R_JAREF * MAXROW + r] = d.l_k0 + d.lim_sign * x0;
        wb[Ld::ROW + Ld::R_TMP * MAXROW + r] = T(0);
    }
    g_sync<T>();
}

// [3P] mj_collision + position part of mj_makeConstraint for one walker (all 16 lanes): detection, commit, contact Jacobians (one-wave form)
template <typename T, typename TP>
__device__ __forceinline__ void g_make_constraints(const GCtx<T, TP>& g, const GLaneTopo<T>& lt, const GKin<T>& kin, int grp, T q, T x0, const T (&x0x)[GD<TP>::NXA],
                                                   int& nlim_out, int& ncon_out, int& my_lim, T& lim_sign) {
    GDet<T, TP> d;
    g_detect_constraints<T, TP>(g, lt, grp, q, d);
    g_commit_constraints<T, TP>(g, d, x0);
    g_contact_jacobians<T, TP>(g, lt, kin, d.ncon, x0, x0x);
    nlim_out = d.nlim; ncon_out = d.ncon; my_lim = d.my_lim; lim_sign = d.lim_sign;
}

template <typename T> struct GEps;
template <> struct GEps<float> { static constexpr float value = 1.1920929e-7f; };
template <> struct GEps<double> { static constexpr double value = 2.220446049250313e-16; };

// rows JV = J x for the walker and (M x)_j; x_j lives in lane j (xx: the replicated dofs).  M x: row broadcasts of x against
// the lane's row of M.  J x: the limit row of a dof is +-x_j; per contact the three contact-frame components are row sums of the
// lane's own Jacobian column times x_j, expanded to the four pyramid rows by lanes 0..3.
template <typename T, typename TP>
__device__ __forceinline__ T g_apply(const GCtx<T, TP>& g, int ncon, int my_lim, T lim_sign, T x, const T (&xx)[GD<TP>::NXA], const GSmooth<T, TP>& sm, T (&mxx_out)[GD<TP>::NXA], int jdepth) {
    using Ld = GLds<TP>;
    constexpr int N = GD<TP>::NL, NX = GD<TP>::NX, MAXROW = Ld::MAXROW;
    DL_LDS T* wb = g.wb;
    const int j = g.j;
    T mx = sm.mcorr * x, mx1 = T(0), xb = x;
    g_dpp_ready(xb);
    // two accumulators: fourteen dependent multiply-adds are ~100 cycles of latency for a wave that has its SIMD to itself
    static_for<N>([&](auto ai) { constexpr int a = ai.value; if constexpr (a % 2 == 0 || !DL_OPT_MX2) fmac_bcast<a, 1>(mx, xb, sm.mrow[a]); else fmac_bcast<a, 1>(mx1, xb, sm.mrow[a]); });
    mx += mx1;
    if constexpr (NX > 0) {
        T s[NX];
        static_for<NX>([&](auto ti) { constexpr int t = ti.value; mx += sm.mxl[t] * xx[t]; s[t] = sm.mxl[t] * x; });
        gsum_n<NX>(s);
        static_for<NX>([&](auto ti) { constexpr int t = ti.value; mxx_out[t] = sm.mxx[t] * xx[t] + s[t]; });
    }
    if (my_lim >= 0) wb[Ld::ROW + Ld::R_JV * MAXROW + my_lim] = lim_sign * x;
    // two contacts per trip (the Jacobian record after the last contact is zero): six interleaved row sums
    DL_UNROLL(DL_UNROLL_APPLY)
    for (int c = 0; c < ncon; c += 2) {
        const Q4<T> ja = g_jc_load<T, TP>(wb, c, j, jdepth), jb = g_jc_load<T, TP>(wb, c + 1, j, jdepth);
        T mua, mub;
        T r[6] = {ja.a * x, ja.b * x, ja.c * x, jb.a * x, jb.b * x, jb.c * x};
        if constexpr (NX > 0) {
            const Q4<T> B0 = ld4(wb + Ld::CON + Ld::CON_W * c + 4), B1 = ld4(wb + Ld::CON + Ld::CON_W * (c + 1) + 4);
            mua = B0.c; mub = B1.c;
            gsum_n<6>(r);
            g_slide_jx<T, TP>(B0.a, B0.b, xx, r[0], r[1], r[2]); g_slide_jx<T, TP>(B1.a, B1.b, xx, r[3], r[4], r[5]);
        } else {
            mua = wb[Ld::CON + Ld::CON_W * c + Ld::C_MU]; mub = wb[Ld::CON + Ld::CON_W * (c + 1) + Ld::C_MU];
            gsum_n<6>(r);
        }
        if (j < 8) {
            const bool second = j >= 4;
            const T vn = second ? r[3] : r[0], v1 = second ? r[4] : r[1], v2 = second ? r[5] : r[2], mu = second ? mub : mua;
            if (!second || c + 1 < ncon) wb[Ld::ROW + Ld::R_JV * MAXROW + 4 * c + j] = vn + ((j & 1) ? -mu : mu) * ((j & 2) ? v2 : v1);
        }
    }
    return mx;
}

// [3P] mj_forward for one walker spread over 16 lanes.  In: q_j, v_j, force of the motor on dof j, warmstart_j (and the
// replicated root translations qx, vx, warmx).  Out: qacc_j (qaccx).  nefc/niter for diagnostics.
//
// Constraint solver: primal Newton on MuJoCo's convex cost  1/2 (a - a_s)^T M (a - a_s) + sum_i 1/2 D_i min(0, (J a - aref)_i)^2
// (mj_solNewton), started at the warm start.  The minimiser is unique, so the path may differ from MuJoCo's
// (which first compares the warm start with a_s = M^-1 f_smooth): here the gradient  M a - f_smooth - J^T f  never
// needs a_s, one factorisation is spent per iteration and none on bookkeeping.  Per iteration:
//   row forces / cost, J^T f, Hessian rows of constraint rows whose active state flipped (incremental, as MuJoCo),
//   gradient test, H = L L^T, direction, J dir / M dir, exact line search, step, improvement test.
// Line search: the 1-D cost is convex piecewise quadratic; safeguarded Newton on its derivative (bracket [lo, hi],
// Newton candidate, bisection when the candidate leaves the bracket), first trial alpha = 1 (exact when no row
// changes state).  It is written with selects so that the 4 walkers of a wave do not serialise on it.
// TIMED: accumulate shader-clock cycles per section into tacc[8] (diagnostics build only):
// 0 smooth dynamics, 1 constraints, 2 rows/J^T f/Hessian, 3 factor + solve, 4 J dir / M dir, 5 line search + step, 6 #iterations of the wave
// SPLIT: this wave is the dynamics wave of a split workgroup -- it hands (q, B v + a) to its partner, which runs g_fk + g_make_constraints
// on the same LDS regions while this wave does the smooth dynamics, and picks the result up before the solver starts (split_seq: the
// pair's command counter, kept by the caller).
template <typename T, typename TP, bool TIMED = false, bool SPLIT = false>
__device__ __forceinline__ T g_forward(const GCtx<T, TP>& g, const GLaneTopo<T>& lt, int grp, T q, T v, T ctrl_force, T warm,
                                       const GX<T, GD<TP>::NX>& qx, const GX<T, GD<TP>::NX>& vx, const GX<T, GD<TP>::NX>& warmx, GX<T, GD<TP>::NX>& qaccx,
                                       int& ncon_o, int& nefc_o, int& niter_o, long long* tacc = nullptr, int* split_seq = nullptr, T q_next = T(0), T* q_ann = nullptr,
                                       const GX<T, GD<TP>::NX>* qx_next = nullptr, GX<T, GD<TP>::NX>* qx_ann = nullptr) {
    constexpr int N = GD<TP>::NL, NX = GD<TP>::NX, NXA = GD<TP>::NXA;
    using Ld = GLds<TP>;
    constexpr int MAXROW = Ld::MAXROW;
    long long t_last = 0;
    if constexpr (TIMED) t_last = DL_CLOCK();
#ifdef DL_EXP_FINE        // diagnostics: the first phase of the Newton loop in three parts (2 row phase, 3 J^T f + Hessian, 4 reductions + exit test), the rest lumped into 5
    auto tick = [&](int k) {
        if constexpr (TIMED) { const long long t = DL_CLOCK(); const int kk = k < 2 ? k : (k == 2 ? 4 : (k >= 10 ? k - 8 : 5)); tacc[kk] += t - t_last; t_last = t; }
    };
#else
    auto tick = [&](int k) {
        if constexpr (TIMED) { if (k < 10) { const long long t = DL_CLOCK(); tacc[k] += t - t_last; t_last = t; } }
    };
#endif
    DL_LDS T* wb = g.wb;
    const int j = g.j;
    const int jdepth = g_lane_depth(lt);          // (packed contact Jacobians: the slot of this lane's records)
    GKin<T> kin;
    GSmooth<T, TP> sm;
    const GConst<T, TP>& cs = *g.c;
    int nlim, ncon, my_lim;
    T lim_sign;
    if constexpr (SPLIT) {
        using Sp = GSplit<TP>;
        // a pair whose hand-over has failed once (split_seq[3]) takes no further part in the protocol: its walkers are on the exception path
        if (split_seq[3]) { ncon_o = 0; nefc_o = 0; niter_o = 0; return warm; }
        // the kinematics first: body frames and root height go to LDS for the constraint wave, which then needs no kinematics of its own;
        // with them the configuration and the solver's start point; then the rest of the smooth dynamics while the partner works
        // The partner wave works one evaluation AHEAD: with the previous request this wave announced the configuration of this evaluation (inside RK4 it
        // depends on the previous stage's velocity only), and while this wave solved, the partner computed its kinematics, its mass matrix and the
        // configuration half of its constraints.  So an evaluation starts by taking the lane's kinematics (body frame, joint axis) and its row of M from
        // LDS, then sends the request (solver start point, NEXT configuration): the partner commits contacts and rows -- a few stores -- and goes on to
        // the next configuration.  If this evaluation's configuration is not the announced one (first request of a launch, reset, injected state), the
        // request says so and waits for the partner to compute everything now.
        volatile DL_LDS int* fl = (volatile DL_LDS int*)g.mbox0;
        const int seq = ++*split_seq;
        using QBits = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;          // the announced configuration must be THIS one to the last bit of its own type
        bool differs = j < N && __builtin_bit_cast(QBits, q) != __builtin_bit_cast(QBits, *q_ann);
        static_for<NX>([&](auto ti) { constexpr int t = ti.value; differs = differs || __builtin_bit_cast(QBits, qx.x[t]) != __builtin_bit_cast(QBits, qx_ann->x[t]); });
        const bool fast = !__any(differs);
        *q_ann = q_next;
        g.mbox[Sp::MB_Q + j] = q;
        g.mbox[Sp::MB_X0 + j] = (j < N) ? cs.solB * v + warm : T(0);
        g.mbox[Sp::MB_QN + j] = q_next;
        T x0x_s[NXA] = {T(0)};          // the solver's start point of the replicated dofs (uniform over the row)
        if constexpr (NX > 0) {
            static_for<NX>([&](auto ti) { constexpr int t = ti.value; x0x_s[t] = cs.solB * vx.x[t] + warmx.x[t]; });
            *qx_ann = *qx_next;
            if (j == 0) {
                st4(g.mbox + Sp::MB_QX, qx.x[0], qx.x[NX > 1 ? 1 : 0], qx.x[NX > 2 ? 2 : 0], T(0));
                st4(g.mbox + Sp::MB_X0X, x0x_s[0], x0x_s[NX > 1 ? 1 : 0], x0x_s[NX > 2 ? 2 : 0], T(0));
                st4(g.mbox + Sp::MB_QNX, qx_next->x[0], qx_next->x[NX > 1 ? 1 : 0], qx_next->x[NX > 2 ? 2 : 0], T(0));
            }
        }
#ifdef DL_EXP_SPLIT_PROF
        const long long tp0 = DL_CLOCK();
#endif
        // (flags are taken through readfirstlane: wave-uniform values in SGPRs, scalar branches instead of exec-mask regions)
        // what the partner left for this configuration: the lane's kinematics (the frame of its body, its joint axis), the complete row of M (the partner
        // performed the mirror exchange), the lane's limit row and the counts (records and rows follow with the request).  All loads in flight together; they
        // are complete (pinned) before the regions are handed back to the partner with the request.
        auto take = [&]() {
            const DL_LDS T* f = wb + Sp::BFRX + Ld::BFR_W * g.ln->body;
            const Q4<T> f0 = ld4(f), f1 = ld4(f + 4), f2 = ld4(f + 8), ax = ld4(wb + Sp::AXX + 4 * j);
            const DL_LDS T* row = g.mm + j * Ld::MS;
            const Q4<T> m0 = ld4(row), m1 = ld4(row + 4), m2 = ld4(row + 8), m3 = ld4(row + 12), m4 = ld4(row + 16);
            T r_lim = g.mbox[Sp::MB_LIM + j], r_ncon = g.mbox[Sp::MB_NCON], r_nlim = g.mbox[Sp::MB_NLIM];
            lim_sign = g.mbox[Sp::MB_SGN + j];
            g_pin(r_lim); g_pin(r_ncon); g_pin(r_nlim); g_pin(lim_sign);
            my_lim = (int)r_lim; ncon = (int)r_ncon; nlim = (int)r_nlim;
            kin.X = mk<T>(f0.a, f0.b, f0.c); kin.Y = mk<T>(f0.d, f1.a, f1.b); kin.Z = mk<T>(f1.c, f1.d, f2.a); kin.pos = mk<T>(f2.b, f2.c, f2.d);
            kin.axis = mk<T>(ax.a, ax.b, ax.c); kin.rootz = ax.d;
            sm.mrow[0] = m0.a; sm.mrow[1] = m0.b; sm.mrow[2] = m0.c; sm.mrow[3] = m0.d; sm.mrow[4] = m1.a; sm.mrow[5] = m1.b; sm.mrow[6] = m1.c; sm.mrow[7] = m1.d;
            sm.mrow[8] = m2.a; sm.mrow[9] = m2.b; sm.mrow[10] = m2.c; sm.mrow[11] = m2.d; sm.mrow[12] = m3.a; sm.mrow[13] = m3.b; sm.mrow[14] = m3.c; sm.mrow[15] = m3.d;
            sm.mdiag = m4.a; sm.mcorr = m4.b;
            if constexpr (NX > 0) {
                Q4<T> mx = ld4(wb + Sp::MXL + 4 * j);
                g_pin(mx.a); g_pin(mx.d);
                const T mxv[3] = {mx.a, mx.b, mx.c};
                static_for<NX>([&](auto ti) { constexpr int t = ti.value; sm.mxl[t] = mxv[t]; sm.mxx[t] = mx.d + cs.xs_armature[t]; });
            }
#pragma unroll
            for (int a = 0; a < GL; a++) g_pin(sm.mrow[a]);
            g_pin(sm.mdiag); g_pin(sm.mcorr); g_pin(kin.X.x); g_pin(kin.Y.y); g_pin(kin.Z.z); g_pin(kin.pos.x); g_pin(kin.axis.x);
        };
        int answered = 0;
        if (fast) {
            // the look-ahead of the previous request is complete long ago (the partner had a whole solve for it): the flag is read and the data requested in one
            // go -- LDS operations of a wave are performed in order, so data requested after a flag read that returns "posted" is the posted data
            const int pre = fl[Sp::MB_PRE];
            g_sync<T>();          // (compiler fence: the data loads below must not be moved in front of the flag read -- `volatile` orders the flag read against other volatile accesses only)
            take();
            answered = DL_UNIFORM((int)(pre == seq - 1));
            if (!answered) {
                for (int it = 0; !(answered = DL_UNIFORM((int)(fl[Sp::MB_PRE] == seq - 1))) && it < g.spin_limit; it++) DL_SLEEP();
                if (answered) { DL_WG_ACQUIRE(); g_sync<T>(); take(); }
            }
        } else {
            g_sync<T>();
            DL_WG_RELEASE();
            if (grp == 0 && j == 0) { ((volatile DL_LDS int*)g.mbox0)[Sp::MB_CMD] = 2; ((volatile DL_LDS int*)g.mbox0)[Sp::MB_CMDSEQ] = seq; }
            DL_WAKE();
            for (int it = 0; !(answered = DL_UNIFORM((int)(fl[Sp::MB_MOK] == seq))) && it < g.spin_limit; it++) DL_SLEEP();
            if (answered) { DL_WG_ACQUIRE(); g_sync<T>(); take(); }
        }
        if (answered) {
            g_sync<T>();
            DL_WG_RELEASE();
            if (fast) { if (grp == 0 && j == 0) { ((volatile DL_LDS int*)g.mbox0)[Sp::MB_CMD] = 1; ((volatile DL_LDS int*)g.mbox0)[Sp::MB_CMDSEQ] = seq; } }
            else if (grp == 0 && j == 0) fl[Sp::MB_MFREE] = seq;
            DL_WAKE();
#ifdef DL_EXP_SPLIT_PROF
            const long long tp1 = DL_CLOCK();
#endif
            g_smooth_dynamics<T, TP, false, true, false>(g, lt, q, v, ctrl_force, qx, vx, kin, sm);
            tick(0);
#ifdef DL_EXP_SPLIT_PROF
            const long long tp2 = DL_CLOCK();
#endif
            answered = 0;
            for (int it = 0; !(answered = DL_UNIFORM((int)(fl[Sp::MB_DONESEQ] == seq))) && it < g.spin_limit; it++) DL_SLEEP();
#ifdef DL_EXP_SPLIT_PROF          // [1]: waiting for the rows (DL_EXP_SPLIT_PROF = 1 / 3) or taking kinematics + mass matrix incl. any wait (= 2); [2]: the velocity half
            split_seq[1] += (int)(((DL_EXP_SPLIT_PROF == 2) ? (tp1 - tp0) : (DL_CLOCK() - tp2)) >> 4); split_seq[2] += (int)((tp2 - tp1) >> 4);
#endif
        }
        if (!answered) {      // timeout (wave-uniform): never carry on with stale rows -- fault word, exception path for the four walkers
            split_seq[3] = 1;
            if (grp == 0 && j == 0 && g.fault) DL_FAULT_OR(g.fault, DL_FAULT_DYN_TIMEOUT);
            ncon_o = 0; nefc_o = 0; niter_o = 0;
            return warm;
        }
        DL_WG_ACQUIRE();
        g_sync<T>();
#if !DL_JAC_ON_PARTNER
        {   // the contact Jacobians are this wave's part of the constraint stage (its partner is the slower of the two otherwise)
            g_contact_jacobians<T, TP>(g, lt, kin, ncon, (j < N) ? cs.solB * v + warm : T(0), x0x_s);
        }
#endif
    } else {
    g_smooth_dynamics<T, TP>(g, lt, q, v, ctrl_force, qx, vx, kin, sm);
    tick(0);
    // rows: D, jar = J a - aref at a = warm start (K imp r + J (B v + a)), cleared "active" flags (TMP) of the Hessian
    T x0x[NXA];
    static_for<NX>([&](auto ti) { constexpr int t = ti.value; x0x[t] = cs.solB * vx.x[t] + warmx.x[t]; });
    g_make_constraints<T, TP>(g, lt, kin, grp, q, (j < N) ? cs.solB * v + warm : T(0), x0x, nlim, ncon, my_lim, lim_sign);
    }
    const T smooth = sm.smooth;
    const int nefc = nlim + 4 * ncon;
    ncon_o = ncon; nefc_o = nefc;
    DL_LDS T* rD = wb + Ld::ROW + Ld::R_D * MAXROW;
    DL_LDS T* rJA = wb + Ld::ROW + Ld::R_JAREF * MAXROW;
    DL_LDS T* rJV = wb + Ld::ROW + Ld::R_JV * MAXROW;
    DL_LDS T* rTM = wb + Ld::ROW + Ld::R_TMP * MAXROW;
    T qacc = warm, Ma = sm.mcorr * warm;
    static_for<N>([&](auto ai) { constexpr int a = ai.value; Ma += sm.mrow[a] * rbcast<a>(qacc); });
    // replicated dofs: acceleration, M a, and their part of the Hessian (uniform block hxx, this lane's coupling hxl)
    T qax[NXA], Max[NXA], hxx[NXA][NXA], hxl[NXA];
    if constexpr (NX > 0) {
        T s[NX];
        static_for<NX>([&](auto ti) { constexpr int t = ti.value; qax[t] = warmx.x[t]; Ma += sm.mxl[t] * qax[t]; s[t] = sm.mxl[t] * qacc; });
        gsum_n<NX>(s);
        static_for<NX>([&](auto ti) {
            constexpr int t = ti.value;
            Max[t] = sm.mxx[t] * qax[t] + s[t];
            hxl[t] = sm.mxl[t];
            static_for<NX>([&](auto ui) { hxx[t][ui.value] = (ui.value == t) ? sm.mxx[t] : T(0); });
        });
    }
    T h[GL], hd = sm.mdiag;       // row j of H = M + sum_active D row^T row (off-diagonal part) and its diagonal
#pragma unroll
    for (int a = 0; a < GL; a++) h[a] = sm.mrow[a];
    const T nvf = cs.nvf, scale = cs.scale;
    bool alive = true;         // this walker still iterates (identical in the 16 lanes of the row)
    int iter = 0;
    g_sync<T>();
    tick(1);
    // dl_config.strict_solver (one-wave form; wave-uniform): the decisions of [3P] mj_solNewton instead of the product's path: start at the cheaper of the warm start and
    // qacc_smooth, every line search run to its derivative tolerance (no Armijo acceptance of the first trial), no early exit on an unchanged active set, IEEE division /
    // square root in the line search and the gradient test.  The minimiser is the same; the iterates -- and the iteration count the diagnostics report -- follow the reference's.
    bool strict_ = false;
    if constexpr (!SPLIT) strict_ = g.strict != 0;
    // the Newton direction -H^-1 grad for the Hessian as it stands (h, hd, hxx, hxl); used by the loop and, in strict mode, once before it with H = M
    auto newton_dir = [&](T grad, const T (&gradx)[NXA], T& dir, T (&dirx)[NXA]) {
        T up[GL], lo[GL], invd = T(1), hdk = hd;
#pragma unroll
        for (int a = 0; a < GL; a++) { up[a] = h[a]; lo[a] = T(0); }
        GCholX<T, NXA> cx;
        constexpr bool X_LAST = NX > 0 && DL_CHOL_X_LAST && DL_CHOL_LEAF_FIRST && DL_CHOL_SHORT_CHAIN;
        if constexpr (NX > 0 && !X_LAST) {
            T wxx[NXA][NXA], wxl[NXA];
            static_for<NX>([&](auto ti) { constexpr int t = ti.value; wxl[t] = hxl[t]; static_for<NX>([&](auto ui) { wxx[t][ui.value] = hxx[t][ui.value]; }); });
            g_chol_x<T, NX, N>(wxx, wxl, up, hdk, cx, T(1e-10));
        }
        T sx[NXA];
        if constexpr (X_LAST) {
            GCholXL<T, NXA> cl;
            g_chol_rev<T, TP>(up, lo, hdk, invd, j, T(1e-10));
            g_chol_x_last<T, TP>(lo, invd, hxl, hxx, cl, T(1e-10));
            dir = -g_chol_solve_rev_x<T, TP>(lo, up, invd, cl, grad, gradx, sx, j);
        } else
        if constexpr (NX == 0 && DL_CHOL_LEAF_FIRST) {
            g_chol_rev<T, TP>(up, lo, hdk, invd, j, T(1e-10));
            dir = -g_chol_solve_rev<T, TP>(lo, up, invd, grad, j);
        } else {
            g_chol<T, N>(up, lo, hdk, invd, j, T(1e-10));
            dir = -g_chol_solve_x<T, NX, N>(cx, lo, up, invd, grad, gradx, sx, j);
        }
        static_for<NX>([&](auto ti) { dirx[ti.value] = -sx[ti.value]; });
    };
    if constexpr (!SPLIT) {
        if (strict_ && nefc > 0) {
            // [3P] mj_solNewton: "if the cost at qacc_smooth is lower than at the warm start, start there".  H = M before any row is active, so qacc_smooth - warm = -M^-1 (M warm - smooth):
            // the loop's own direction code with the gradient of the unconstrained problem; then J d (rows JV) and M d by the loop's own g_apply
            T g0x[NXA], d0, d0x[NXA], Md0x[NXA];
            static_for<NX>([&](auto ti) { constexpr int t = ti.value; g0x[t] = Max[t] - sm.smoothx[t]; });
            newton_dir(Ma - smooth, g0x, d0, d0x);
            const T Md0 = g_apply<T, TP>(g, ncon, my_lim, lim_sign, d0, d0x, sm, Md0x, jdepth);
            g_sync<T>();
            // cost(warm) = 1/2 d^T M d + sum 1/2 D min(0, jar)^2,  cost(qacc_smooth) = sum 1/2 D min(0, jar + jv)^2
            T cw = T(0), cs0 = T(0);
            for (int r = j; r < nefc; r += GL) {
                const T a_ = rJA[r], b_ = a_ + rJV[r], D = rD[r];
                if (a_ < T(0)) cw += T(0.5) * D * a_ * a_;
                if (b_ < T(0)) cs0 += T(0.5) * D * b_ * b_;
            }
            T r3s[3] = {cw, cs0, (j < N) ? T(0.5) * d0 * Md0 : T(0)};
            gsum_n<3>(r3s);
            static_for<NX>([&](auto ti) { constexpr int t = ti.value; r3s[2] += T(0.5) * d0x[t] * Md0x[t]; });
            if (r3s[1] < r3s[0] + r3s[2]) {          // (identical in the 16 lanes of the row: row sums)
                qacc += d0; Ma += Md0;
                static_for<NX>([&](auto ti) { constexpr int t = ti.value; qax[t] += d0x[t]; Max[t] += Md0x[t]; });
                for (int r = j; r < nefc; r += GL) rJA[r] += rJV[r];
            }
            g_sync<T>();
        }
    }
    for (;;) {
        // ---- rows are processed by their owners: force, cost and active-set flips of a limit row by the lane of its
        // dof, of the four pyramid rows of contact c by lane c, which leaves the contact-frame force and the weights of
        // the rank-3 Hessian update for the dof lanes
        T c = T(0), fcon = T(0);
        if (my_lim >= 0) {
            const T jar = rJA[my_lim], D = rD[my_lim];
            const bool on = jar < T(0), was = rTM[my_lim] != T(0);
            if (on) { c = T(0.5) * D * jar * jar; fcon = -lim_sign * D * jar; }
            if (on != was && alive) hd += on ? D : -D;
            rTM[my_lim] = on ? T(1) : T(0);
        }
        for (int cc = j; cc < ncon; cc += GL) {
            const Q4<T> ja = ld4(rJA + 4 * cc), tm = ld4(rTM + 4 * cc);
            const T D = rD[4 * cc];
            T mu, ctx = T(0), cty = T(0);
            if constexpr (NX > 0) { const Q4<T> B = ld4(wb + Ld::CON + Ld::CON_W * cc + 4); ctx = B.a; cty = B.b; mu = B.c; }
            else mu = wb[Ld::CON + Ld::CON_W * cc + Ld::C_MU];
            const T jar[4] = {ja.a, ja.b, ja.c, ja.d}, was[4] = {tm.a, tm.b, tm.c, tm.d};
            T f4[4], dD[4], onf[4], c4[4];
            bool anyflip = false;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const bool on = jar[s4] < T(0), w = was[s4] != T(0);
                f4[s4] = on ? -D * jar[s4] : T(0);
                c4[s4] = on ? T(0.5) * D * jar[s4] * jar[s4] : T(0);
                dD[s4] = (on == w) ? T(0) : (on ? D : -D);
                anyflip = anyflip || (on != w);
                onf[s4] = on ? T(1) : T(0);
            }
            c += (c4[0] + c4[1]) + (c4[2] + c4[3]);
            // dW = sum_s dD_s d_s d_s^T with d = (1, +-mu, 0) or (1, 0, +-mu)
            DL_LDS T* fc = wb + Ld::FC + Ld::FC_W * cc;
            const T Fn = f4[0] + f4[1] + f4[2] + f4[3], F1 = mu * (f4[0] - f4[1]), F2 = mu * (f4[2] - f4[3]);
            st4(fc, Fn, F1, F2, anyflip ? T(1) : T(0));
            st4(fc + 4, dD[0] + dD[1] + dD[2] + dD[3], mu * (dD[0] - dD[1]), mu * (dD[2] - dD[3]), mu * mu * (dD[0] + dD[1]));
            // world-frame force of the contact: what the replicated root translations see of it
            if constexpr (NX > 0) st4(fc + 8, mu * mu * (dD[2] + dD[3]), ctx * F1 - cty * F2, cty * F1 + ctx * F2, Fn);
            else fc[8] = mu * mu * (dD[2] + dD[3]);
            st4(rTM + 4 * cc, onf[0], onf[1], onf[2], onf[3]);
        }
        g_sync<T>();
        tick(10);
        // ---- J^T f and the Hessian rows (dof lanes; the lane's Jacobian column of contact cc is one 16-byte read)
        T fcx[NXA];
        static_for<NX>([&](auto ti) { fcx[ti.value] = T(0); });
        DL_UNROLL(DL_UNROLL_JTF)
        for (int c2 = 0; c2 < ncon; c2 += 2) {
            const DL_LDS T* fca = wb + Ld::FC + Ld::FC_W * c2;
            const Q4<T> Fa = ld4(fca), Fb = ld4(fca + Ld::FC_W), ja = g_jc_load<T, TP>(wb, c2, j, jdepth), jb = g_jc_load<T, TP>(wb, c2 + 1, j, jdepth);
            fcon += (ja.a * Fa.a + ja.b * Fa.b + ja.c * Fa.c) + (jb.a * Fb.a + jb.b * Fb.b + jb.c * Fb.c);
            if constexpr (NX > 0) {
                const Q4<T> Ga = ld4(fca + 8), Gb = ld4(fca + Ld::FC_W + 8);      // (w22, Fx, Fy, Fz)
                const V3<T> Fw = mk<T>(Ga.b + Gb.b, Ga.c + Gb.c, Ga.d + Gb.d);
                static_for<NX>([&](auto ti) { constexpr int t = ti.value; fcx[t] += T(TP::dof_sign(t)) * vcomp<TP::dof_axis(t)>(Fw); });
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const Q4<T>& F = half ? Fb : Fa;
                const Q4<T>& jt = half ? jb : ja;
                if (F.d != T(0) && alive) {
                    const DL_LDS T* fc = fca + half * Ld::FC_W;
                    const Q4<T> W = ld4(fc + 4);
                    const T w22 = fc[8];
                    const T t0 = W.a * jt.a + W.b * jt.b + W.c * jt.c, t1 = W.b * jt.a + W.d * jt.b, t2 = W.c * jt.a + w22 * jt.c;
                    hd += jt.a * t0 + jt.b * t1 + jt.c * t2;
                    T bj[3] = {jt.a, jt.b, jt.c};
                    g_dpp_ready_n<3>(bj);                         // one hazard wait for the three broadcast sources
                    static_for<N>([&](auto ai) { constexpr int a = ai.value; fmac_bcast<a, 1>(h[a], bj[0], t0); fmac_bcast<a, 1>(h[a], bj[1], t1); fmac_bcast<a, 1>(h[a], bj[2], t2); });
                    if constexpr (NX > 0) {
                        // replicated dofs: H[j][t] += J_x[t] . (W J_j), H[t][u] += J_x[t] . (W J_x[u])
                        const Q4<T> B = ld4(wb + Ld::CON + Ld::CON_W * (c2 + half) + 4);
                        T jn[NX], j1[NX], j2[NX], u0[NX], u1[NX], u2[NX];
                        static_for<NX>([&](auto ti) {
                            constexpr int t = ti.value;
                            g_slide_col<T, TP, t>(B.a, B.b, jn[t], j1[t], j2[t]);
                            hxl[t] += jn[t] * t0 + j1[t] * t1 + j2[t] * t2;
                            u0[t] = W.a * jn[t] + W.b * j1[t] + W.c * j2[t]; u1[t] = W.b * jn[t] + W.d * j1[t]; u2[t] = W.c * jn[t] + w22 * j2[t];
                        });
                        static_for<NX>([&](auto ti) {
                            constexpr int t = ti.value;
                            static_for<t + 1>([&](auto ui) { constexpr int u = ui.value; hxx[t][u] += jn[t] * u0[u] + j1[t] * u1[u] + j2[t] * u2[u]; });
                        });
                    }
                }
            }
        }
        tick(11);
        const T grad = Ma - smooth - fcon;
        T gradx[NXA];
        T r3[3] = {c, grad * grad, dl_abs(Ma) + dl_abs(smooth) + dl_abs(fcon)};
        gsum_n<3>(r3);
        static_for<NX>([&](auto ti) {
            constexpr int t = ti.value;
            gradx[t] = Max[t] - sm.smoothx[t] - fcx[t];
            r3[1] += gradx[t] * gradx[t];
            r3[2] += dl_abs(Max[t]) + dl_abs(sm.smoothx[t]) + dl_abs(fcx[t]);
        });
        const T pc0 = r3[0];
        {
            const T gn = r3[1], gmag = r3[2];
            // float32: the gradient carries rounding noise proportional to the magnitude of its terms.  scale |g| >= bound, compared as squares
            // (both sides are >= 0): the correctly rounded square root was a dozen dependent instructions in front of the loop's exit branch
            const T bound = cs.tolerance + cs.tol_rel * scale * gmag;
#if DL_OPT_GRADSQ
            if (strict_) { if (!(scale * dl_sqrt(gn) >= bound) || iter >= cs.iterations) alive = false; }
            else
            if (!(scale * scale * gn >= bound * bound) || iter >= cs.iterations) alive = false;   // also stops on NaN
#else
            if (!(scale * dl_sqrt(gn) >= bound) || iter >= cs.iterations) alive = false;
#endif
        }
        tick(2);
        if (!__any(alive)) break;
        if constexpr (TIMED) tacc[6] += 1;
        // ---- Newton direction
        T dir, dirx[NXA];
        newton_dir(grad, gradx, dir, dirx);
        tick(3);
        T Mdx[NXA];
        const T Md = g_apply<T, TP>(g, ncon, my_lim, lim_sign, dir, dirx, sm, Mdx, jdepth);      // rows JV = J dir
        g_sync<T>();
        tick(4);
        // ---- the common case: the full Newton step leaves the active set as it is.  The cost is quadratic on that
        // set, so a + dir is its exact minimiser (gradient zero up to rounding): no line search, no further pass over
        // the rows.  Checked per walker; the line search below only runs if some walker of the wave still needs it.
        // The lane's rows (r = j, j + 16, j + 32) are taken into registers once: the check below, every trial of the line
        // search and the final update then run without LDS round trips; rows beyond 48 (rare) stay in LDS.
        constexpr int NS = 3;
        T ja[NS], jv[NS], dd[NS];
        {
            bool flips = !(dir == dir);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int r = j + GL * s;
                const bool ok = r < nefc;
                const int rr = ok ? r : 0;
                const T a_ = rJA[rr], v_ = rJV[rr], d_ = rD[rr], t_ = rTM[rr];
                ja[s] = ok ? a_ : T(1); jv[s] = ok ? v_ : T(0); dd[s] = ok ? d_ : T(0);
                flips = flips || (ok && ((a_ + v_ < T(0)) != (t_ != T(0))));
            }
            for (int r = j + GL * NS; r < nefc; r += GL) flips = flips || ((rJA[r] + rJV[r] < T(0)) != (rTM[r] != T(0)));
            if (alive && !strict_ && !gany(flips)) {
                qacc += dir; iter++; alive = false;
                static_for<NX>([&](auto ti) { qax[ti.value] += dirx[ti.value]; });
            }
        }
        if (!__any(alive)) { tick(5); break; }
        // ---- exact line search along dir
        T r4[4] = {dir * (Ma - smooth), T(0.5) * dir * Md, dir * grad, dir * dir};
        gsum_n<4>(r4);
        static_for<NX>([&](auto ti) {
            constexpr int t = ti.value;
            r4[0] += dirx[t] * (Max[t] - sm.smoothx[t]); r4[1] += T(0.5) * dirx[t] * Mdx[t]; r4[2] += dirx[t] * gradx[t]; r4[3] += dirx[t] * dirx[t];
        });
        const T g1s = r4[0], g2 = r4[1], d0 = r4[2], snorm = (DL_OPT_FSQRT && !strict_) ? dl_sqrt_fast(r4[3]) : dl_sqrt(r4[3]);     // (scales a tolerance: 1 ulp is plenty)
        const T gtol = cs.tolerance * cs.ls_tolerance * snorm * cs.meaninertia * nvf + cs.ls_reltol * dl_abs(d0);
        T alpha = T(1), lo = T(0), hi = T(1e30), best_a = T(0), best_dc = T(0), best_mag = T(0), res_a = T(0), res_dc = T(0), res_mag = T(0);
        bool done = !alive || !(snorm >= T(1e-15));
        const int maxit = cs.ls_iterations;
        int it = 0;
        while (__any(!done)) {
            T pc = T(0), pd1 = T(0), pd2 = T(0);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const T xx = ja[s] + alpha * jv[s];
                const T dx = (xx < T(0)) ? dd[s] * xx : T(0), dj = (xx < T(0)) ? dd[s] * jv[s] : T(0);
                pc += T(0.5) * dx * xx; pd1 += dx * jv[s]; pd2 += dj * jv[s];
            }
            for (int r = j + GL * NS; r < nefc; r += GL) {
                const T jvr = rJV[r], xx = rJA[r] + alpha * jvr;
                if (xx < T(0)) { const T D = rD[r]; pc += T(0.5) * D * xx * xx; pd1 += D * xx * jvr; pd2 += D * jvr * jvr; }
            }
            { T r3b[3] = {pc, pd1, pd2}; gsum_n<3>(r3b); pc = r3b[0]; pd1 = r3b[1]; pd2 = r3b[2]; }
            const T d1 = g1s + T(2) * alpha * g2 + pd1, d2 = T(2) * g2 + pd2;
            const T dc = alpha * g1s + alpha * alpha * g2 + (pc - pc0);
            const T mag = dl_abs(alpha * g1s) + alpha * alpha * g2 + pc + pc0;
            it++;
#if DL_OPT_ARMIJO
            // the full Newton step is taken as it is when it already brings a fair part of the decrease the quadratic model promises (d0 / 2 on
            // an unchanged active set): the exact minimiser along dir is worth two or three more trials only when it does not
            if (!done && it == 1 && !strict_ && dc <= T(DL_OPT_ARMIJO_C) * d0) { res_a = alpha; res_dc = dc; res_mag = mag; done = true; }
#endif
            if (!done) {
                if (d1 < T(0)) lo = alpha; else hi = alpha;
                const T cand = strict_ ? alpha - d1 / d2 : alpha - d1 * dl_rcp(d2);
                const T mid = hi < T(1e29) ? T(0.5) * (lo + hi) : T(2) * alpha;
                const T next = (cand > lo && cand < hi) ? cand : mid;
                // converged: derivative below tolerance, or the iterate no longer moves at working precision
                const bool conv = dl_abs(d1) < gtol || dl_abs(next - alpha) <= T(4) * GEps<T>::value * alpha;
                if (dc < best_dc) { best_dc = dc; best_a = alpha; best_mag = mag; }
                if (conv) { res_a = alpha; res_dc = dc; res_mag = mag; done = true; }
                else if (it >= maxit) { res_a = best_a; res_dc = best_dc; res_mag = best_mag; done = true; }
                alpha = next;
            }
        }
#ifdef DL_EXP_LS_COUNT            // diagnostics: trials and line searches of the wave (tools/diag_sections.py --ls)
        if constexpr (TIMED) tacc[7] += (long long)it * 65536 + 1;
#endif
        // ---- step
        if (alive) {
            if (res_a == T(0)) alive = false;                      // no improvement along a descent direction: converged to working precision
            else {
                qacc += res_a * dir; Ma += res_a * Md;
                static_for<NX>([&](auto ti) { constexpr int t = ti.value; qax[t] += res_a * dirx[t]; Max[t] += res_a * Mdx[t]; });
#pragma unroll
                for (int s = 0; s < NS; s++) { const int r = j + GL * s; if (r < nefc) rJA[r] = ja[s] + res_a * jv[s]; }
                for (int r = j + GL * NS; r < nefc; r += GL) rJA[r] += res_a * rJV[r];
                iter++;
                // mj_solNewton's improvement test (float32: relative to the magnitude of the terms of the cost difference)
                if (!(scale * -res_dc >= cs.tolerance + cs.tol_rel * scale * res_mag) || nefc == 0) alive = false;
            }
        }
        g_sync<T>();
        tick(5);
    }
    niter_o = iter;
    static_for<NX>([&](auto ti) { qaccx.x[ti.value] = qax[ti.value]; });
    return qacc;
}

}  // namespace dl
