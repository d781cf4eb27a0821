/*
 * dl_oracle.h -- CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * A plain-C, float64, scalar restatement of the reference path
 *   MimicEnv.step / reset_model          /root/reference/drloco/mujoco/mimic_env.py:60-126,526-572
 *   StraightWalkingTrajectories.next     /root/reference/drloco/ref_trajecs/straight_walk_trajecs.py:141-159,322-348
 *   mj_step (RK4) of third-party MuJoCo  (call site mimic_env.py:83; algorithm restated from the
 *                                         MuJoCo 2.x documentation/source, see DESIGN.md "oracle")
 *   SB3 1.0 RunningMeanStd / VecNormalize / RolloutBuffer GAE
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY STATUS
 *   env logic (cursor, reward, observation, termination, action mapping, monitor smoothing):
 *     pinned against the golden vectors tests/golden/G1..G11, G13..G15 generated from the reference itself.
 *   dynamics (mj_step): PARITY UNPINNED -- MuJoCo is a third-party binary that is not in
 *     /root/reference and not installable here; pinned only by physics known-answer tests
 *     (tests/test_oracle_physics.py).
 *   SB3 reductions: PARITY UNPINNED (SB3 not installed) -- restated from the published
 *     formulas and checked against closed forms.
 */
#ifndef DL_ORACLE_H
#define DL_ORACLE_H

#include "../include/drloco_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DLO_MAXCON 32
#define DLO_MAXEFC (4 * DLO_MAXCON + 2 * DL_MAX_DOF)

typedef struct dlo_env_s dlo_env; /* N walkers */

/* fills body_invweight0 / dof_invweight0 / meaninertia of `m` (mj_setConst at qpos0) */
void dlo_set_const(dl_model_desc* m);

dlo_env* dlo_create(const dl_model_desc* model, const dl_refs_desc* refs, const dl_config* cfg,
                    int32_t n_envs);
void dlo_destroy(dlo_env* e);

/* all pointers are HOST pointers; layouts as in drloco_hip.h but state is always double */
void dlo_reset(dlo_env* e, const uint8_t* mask, const int32_t* init_step, const int32_t* init_pos,
               double* obs_out);
void dlo_step(dlo_env* e, const double* actions, double* obs, double* rew, uint8_t* done,
              double* term_obs, double* rew_terms);
void dlo_get_state(dlo_env* e, double* qpos, double* qvel, double* qacc_warm, int32_t* cursor,
                   double* walked);
void dlo_set_state(dlo_env* e, const double* qpos, const double* qvel, const double* qacc_warm,
                   const int32_t* cursor, const double* walked);
/* quirk Q4: the COM-z offsets the steps of every walker's data set carry (adjust_COM_Z_pos), double[n_steps, N] */
void dlo_get_ref_offsets(dlo_env* e, double* z);
void dlo_set_ref_offsets(dlo_env* e, const double* z);
void dlo_forward(dlo_env* e, const double* ctrl, double* qacc, int32_t* ncon, int32_t* nefc,
                 int32_t* niter);
/* MimicEnv.activate_evaluation for all walkers */
void dlo_set_eval(dlo_env* e, int32_t on);
/* inject a "MujocoException" for walker i at its next step (mimic_env.py:86-91) */
void dlo_inject_exception(dlo_env* e, int32_t i);
/* all following resets of walker i use (step,pos) as the RSI draw; step < 0 clears */
void dlo_inject_rsi(dlo_env* e, int32_t i, int32_t step, int32_t pos);
/* replace the dynamics of the next step of walker i by a given end state (golden G4 traces) */
void dlo_inject_state(dlo_env* e, int32_t i, const double* qpos, const double* qvel);
int dlo_stats_snapshot(dlo_env* e, const char* name, double* out);
/* observation / imitation reward at the current cursor and state, without stepping */
void dlo_observe(dlo_env* e, double* obs, double* imit, double* terms);
/* torques applied in the last step: double[N, nu] */
void dlo_last_ctrl(dlo_env* e, double* out);
/* feed one Monitor.step record of walker i (golden G7) */
void dlo_monitor_feed(dlo_env* e, int32_t i, double rew, int32_t done, double pos, double vel,
                      double com, double tor, double walked);
/* reference lookup at the current cursor: q_ref[nv], v_ref[nv] of walker i */
void dlo_ref_lookup(dlo_env* e, int32_t i, double* qref, double* vref);
/* build-defined stress test (BASELINE config 5): per-walker body mass/inertia scale, floor friction, push force on the
 * torso (world frame, double[N,3]); NULL leaves a field unchanged */
void dlo_set_randomization(dlo_env* e, const double* mass_scale, const double* floor_friction, const double* xfrc);
/* do_terminate_early (mimic_env.py:652-702), unused by step(): flags[4] */
void dlo_terminate_early(dlo_env* e, int32_t i, int32_t* flags);

/* single-walker physics probes for the known-answer tests (state given explicitly) */
typedef struct dlo_probe {
    double M[DL_MAX_DOF * DL_MAX_DOF];
    double qfrc_bias[DL_MAX_DOF];
    double qfrc_smooth[DL_MAX_DOF];
    double qacc_smooth[DL_MAX_DOF];
    double qacc[DL_MAX_DOF];
    double qfrc_constraint[DL_MAX_DOF];
    double xpos[DL_MAX_BODY * 3];
    double xmat[DL_MAX_BODY * 9];
    double xipos[DL_MAX_BODY * 3];
    double site_xpos[DL_MAX_SITE * 3];
    double energy[2]; /* potential, kinetic */
    int32_t ncon, nefc, niter;
    double con_pos[DLO_MAXCON * 3];
    double con_dist[DLO_MAXCON];
    double con_frame[DLO_MAXCON * 9];
    int32_t con_geom[DLO_MAXCON];
    double efc_J[DLO_MAXEFC * DL_MAX_DOF];
    double efc_pos[DLO_MAXEFC], efc_D[DLO_MAXEFC], efc_aref[DLO_MAXEFC], efc_force[DLO_MAXEFC];
    double solver_cost;
} dlo_probe;
/* flags: bit0 disable contacts, bit1 disable limits, bit2 disable damping, bit3 disable gravity,
 * bit4 disable actuation */
void dlo_probe_forward(const dl_model_desc* m, const double* qpos, const double* qvel,
                       const double* ctrl, const double* warm, int flags, dlo_probe* out);
/* n RK4 mj_steps of one walker with options; returns 0 or the step at which it diverged */
int dlo_probe_steps(const dl_model_desc* m, double* qpos, double* qvel, const double* ctrl,
                    double* warm, double dt, int n, int flags);

/* warm-start schedule of the RK4 stages: 0 = per evaluation (the device kernels), 1 = per mj_step as in MuJoCo 2.x
 * (qacc_warmstart saved once per step by mj_advance); process-global, see dl_oracle.c */
void dlo_set_warmstart_schedule(int schedule);

/* SB3 reductions (host, double / float as in SB3) */
void dlo_moments_update(double* mean, double* var, double* count, const double* x, int32_t B,
                        int32_t D);
void dlo_gae(const float* rew, const float* val, const uint8_t* ep_start, const float* last_val,
             const uint8_t* last_done, float gamma, float lam, int32_t T, int32_t N, float* adv,
             float* ret);

/* RSI random stream shared bit-exactly with the device kernels */
void dlo_rsi_draw(uint64_t seed, uint32_t global_env, uint32_t episode, int32_t n_steps,
                  const int32_t* step_off, int32_t* i_step, int32_t* pos);

#ifdef __cplusplus
}
#endif
#endif
