/*
 * drloco_hip.h -- C-ABI of the MI355X-native DRLoco hot path.
 *
 * The library (drloco_amd/csrc/libdrloco_hip.so) replaces, for N walkers at once,
 * the reference's per-process environment stack
 *
 *     VecNormalize( SubprocVecEnv( Monitor( MimicEnv ) ) )      drloco/common/utils.py:97-134
 *
 * and SB3's RolloutBuffer return/advantage scan.  Every entry point cites the reference
 * interface it stands in for.  Conventions:
 *
 *   - plain C, no exceptions/longjmp across the boundary; every function returns 0 on
 *     success or a negative DL_E_* code, and dl_last_error() returns the text of the last
 *     failure on the calling thread;
 *   - unless stated otherwise every array pointer is a DEVICE pointer owned by the caller
 *     (e.g. torch tensors); work is enqueued on `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream) and nothing synchronises with the host;
 *   - per-walker state lives inside the handle as structure-of-arrays [field][N];
 *   - a handle is not thread-safe; one host thread per GPU process.
 *
 * There is no CPU fallback behind these symbols: dl_create() fails with DL_E_NODEVICE when
 * no HIP device is present.
 */
#ifndef DRLOCO_HIP_H
#define DRLOCO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL_ABI_VERSION 7   /* 2: dl_rollout_policy, dl_vecnorm_state, dl_profile_steps; dl_profile takes a sampling stride.  3: dl_adv_stats takes a caller-owned workspace; dl_vecnormalize_step flag 16.  4: dl_vecnormalize_steps, dl_set_split.  5: DL_E_FAULT, dl_fault_check / dl_fault_clear, dl_collect_rollouts, dl_vecnormalize_step flag 32, dl_policy_pack / dl_policy_forward_packed, dl_vn_local_sums / dl_vn_merge_sums (flag 64).  6: dl_policy_forward_pair, DL_ROLLOUT_WORKGROUP_TILES, DL_ROLLOUT_DETERMINISTIC, dl_stats_snapshot first_ep_*.  7: dl_config.intended_semantics (quirk Q4 reproduced by default; switches for Q2-Q4) and dl_config.strict_solver, dl_get_ref_offsets / dl_set_ref_offsets, dl_dpp_probe */

/* static capacities of the POD descriptors */
#define DL_MAX_BODY 12
#define DL_MAX_DOF 20
#define DL_MAX_GEOM 12
#define DL_MAX_SITE 8
#define DL_MAX_ACT 16

/* error codes */
#define DL_OK 0
#define DL_E_INVAL (-1)     /* bad argument / unsupported model topology */
#define DL_E_NODEVICE (-2)  /* no HIP device */
#define DL_E_HIP (-3)       /* a HIP runtime call failed */
#define DL_E_NOMEM (-4)
#define DL_E_FAULT (-5)     /* a kernel of this handle reported a fault (dl_fault_check): results since then are not a valid rollout */
/* bits of the fault word (dl_fault_check) */
#define DL_FAULT_DYN_TIMEOUT 1      /* a dynamics wave of a split workgroup gave up waiting for its constraint wave */
#define DL_FAULT_SRV_TIMEOUT 2      /* a constraint wave of a split workgroup gave up waiting for a request */
#define DL_FAULT_GRID_TIMEOUT 4     /* a workgroup of the persistent rollout kernel gave up waiting for the grid-wide exchange */

/* environment kinds (the reference's env_map, drloco/mujoco/config.py:9-10) */
#define DL_ENV_STRAIGHT 0
#define DL_ENV_LOCO3D 1

/* joint / geom type codes (subset of MJCF used by the MJCF files under drloco/mujoco/xml) */
#define DL_JNT_SLIDE 0
#define DL_JNT_HINGE 1
#define DL_GEOM_CAPSULE 0
#define DL_GEOM_BOX 1

/* words of the per-walker cursor/episode record (dl_get_state / dl_set_state), int32 each.
 * They restate the python attributes of StraightWalkingTrajectories / MimicEnv:
 *   drloco/ref_trajecs/straight_walk_trajecs.py:98-126,141-170  and  drloco/mujoco/mimic_env.py:38-45 */
#define DL_CUR_I_STEP 0     /* refs._i_step */
#define DL_CUR_POS 1        /* refs._pos */
#define DL_CUR_RSI_STEP 2   /* index of refs._step (bound at reset only, quirk Q1) */
#define DL_CUR_COUNT 3      /* refs.count_steps_same_vel (never reset, quirk Q2) */
#define DL_CUR_EP_DUR 4     /* env.ep_dur */
#define DL_CUR_HAS_DIST 5   /* 1 once the cursor rolled into a later step (COM-x offset active) */
#define DL_CUR_EPISODE 6    /* number of resets so far (RSI random-stream counter) */
#define DL_CUR_READ_STEP 7  /* step whose table the cursor reads (differs from I_STEP only after an eval init, quirk Q3) */
#define DL_CUR_EVAL_K 8     /* refs.n_deterministic_inits: next step index of the deterministic (evaluation) init */
#define DL_CUR_WORDS 9

/* compiled model: the subset of an MJCF model the path needs.  One degree of freedom per
 * joint (slide/hinge only, nq == nv) as in walker3d_flat_feet.xml / walker_165cm_65kg.xml. */
typedef struct dl_model_desc {
    int32_t nbody;      /* including the world body 0 */
    int32_t nv;         /* = nq */
    int32_t nu;
    int32_t ngeom;      /* collision geoms attached to bodies (the floor plane is implicit: z = 0) */
    int32_t nsite;
    int32_t frame_skip; /* sim_freq / CTRL_FREQ, mimic_env.py:194-207 */
    double timestep;    /* <option timestep>, integrator is RK4 */
    double gravity[3];
    double solref[2];   /* MuJoCo defaults (0.02, 1) */
    double solimp[5];   /* (0.9, 0.95, 0.001, 0.5, 2) */
    double tolerance;   /* solver tolerance, 1e-8 */
    double ls_tolerance; /* 0.01 */
    int32_t iterations;  /* 100 */
    int32_t ls_iterations; /* 50 */
    /* bodies */
    int32_t body_parent[DL_MAX_BODY];
    double body_pos[DL_MAX_BODY][3];
    double body_mass[DL_MAX_BODY];
    double body_ipos[DL_MAX_BODY][3];
    double body_inertia[DL_MAX_BODY][3]; /* diagonal, aligned with the body frame */
    /* joints == dofs, in qpos order; a body's joints are contiguous */
    int32_t jnt_type[DL_MAX_DOF];
    int32_t jnt_body[DL_MAX_DOF];
    double jnt_axis[DL_MAX_DOF][3];  /* body-local unit axis */
    double jnt_pos[DL_MAX_DOF][3];   /* body-local anchor */
    double jnt_qpos0[DL_MAX_DOF];    /* `ref` */
    int32_t jnt_limited[DL_MAX_DOF];
    double jnt_range[DL_MAX_DOF][2];
    double jnt_damping[DL_MAX_DOF];
    double jnt_armature[DL_MAX_DOF];
    /* geoms */
    int32_t geom_type[DL_MAX_GEOM];
    int32_t geom_body[DL_MAX_GEOM];
    double geom_pos[DL_MAX_GEOM][3];   /* body-local centre */
    double geom_mat[DL_MAX_GEOM][9];   /* body-local orientation, row-major; capsule axis = column 2 */
    double geom_size[DL_MAX_GEOM][3];  /* capsule: radius, half-length; box: half extents */
    double geom_friction[DL_MAX_GEOM]; /* sliding friction */
    double floor_friction;
    /* sites (foot corners used by reset_model, mimic_env.py:547-559) */
    int32_t site_body[DL_MAX_SITE];
    double site_pos[DL_MAX_SITE][3];
    /* motors */
    int32_t act_dof[DL_MAX_ACT];
    double act_gear[DL_MAX_ACT];
    double act_ctrlrange[DL_MAX_ACT][2];
    double act_forcerange[DL_MAX_ACT][2];
    /* compile-time constants MuJoCo derives at qpos0 (mj_setConst) */
    double body_invweight0[DL_MAX_BODY][2];
    double dof_invweight0[DL_MAX_DOF];
    double meaninertia;
} dl_model_desc;

/* Mocap table: the used rows of the step-segmented .mat, flattened.
 * drloco/ref_trajecs/straight_walk_trajecs.py:304-320 (loading), mimic_walker3d.py:11-23 (rows).
 * HOST pointers; copied at dl_create. */
typedef struct dl_refs_desc {
    int32_t n_steps;         /* 30 for Trajecs_Constant_Speed_400Hz.mat */
    int32_t n_rows;          /* 2*nv: qpos rows then qvel rows, in model order */
    int32_t total_len;       /* sum of step lengths */
    int32_t stride;          /* sample_freq / control_freq (2) */
    const double* table;     /* [n_rows][total_len] */
    const int32_t* step_off; /* [n_steps+1] */
    const int32_t* step_is_left; /* [n_steps], straight_walk_trajecs.py:221-234 */
    const double* step_vel;  /* [n_steps] smoothed mean COM-x velocity, :393-415 */
} dl_refs_desc;

/* Constants the reference keeps in drloco/config/{config,hypers}.py */
typedef struct dl_config {
    double rew_weights[3];  /* pose, vel, com: hypers.py:48 */
    double rew_scale;       /* hypers.py:51 */
    double alive_bonus;     /* hypers.py:55 */
    double com_z_min;       /* 0.5, mimic_env.py:120 */
    double ctrl_freq;       /* 200, config.py:20 */
    int32_t ep_dur_max;     /* 3000, hypers.py:58 */
    int32_t mirror_policy;  /* is_mod(MOD_MIRR_POLICY), hypers.py:23 */
    int32_t precision;      /* 32 (default) or 64: arithmetic type of the dynamics kernels */
    int32_t env_index_base; /* global index of walker 0 of this shard (multi-GPU) */
    uint64_t seed;          /* RSI random stream seed */
    int32_t env_kind;       /* DL_ENV_STRAIGHT: MimicWalker3dEnv + StraightWalkingTrajectories;
                               DL_ENV_LOCO3D: MimicWalker165cm65kgEnv + Loco3dReferenceTrajectories
                               (drloco/mujoco/config.py:9-10) */
    int32_t lanes_per_walker; /* launch geometry of the dynamics kernels: 1 = one walker per lane, 16 = one walker per
                               16-lane DPP row (models with <= 16 dofs), 0 = auto (16 where supported, else 1) */
    int32_t intended_semantics; /* 0 (default) = strict_reference_quirks: the reference's behaviour incl. its quirks Q1-Q4.  Bits select
                               the obviously intended behaviour instead: DL_INTENDED_* below.  (Q1, the COM-x offset of the step bound at reset, has no switch.) */
    int32_t strict_solver;    /* float32 16-lane kernels (one-wave form): 1 = the Newton solver takes the reference's ([3P] MuJoCo mj_solNewton) decisions -- start at the cheaper
                               of warm start and qacc_smooth, every line search run to its derivative tolerance (no Armijo acceptance of the first trial), no early exit
                               on an unchanged active set, IEEE division and square roots -- at the cost of speed: the mode real MuJoCo vectors (golden G12) are compared under.
                               0 (default): the product's iteration path; the minimiser is the same.  Refused (DL_E_INVAL) for split workgroups. */
} dl_config;
/* bits of dl_config.intended_semantics */
#define DL_INTENDED_COUNT_PER_EPISODE 2  /* Q2 off: count_steps_same_vel restarts at 1 with every reset (the reference never resets it, straight_walk_trajecs.py:124,336) */
#define DL_INTENDED_EVAL_OWN_STEP 4      /* Q3 off: an evaluation init reads the kinematics of ITS step k (the reference reads step 0's table, straight_walk_trajecs.py:237-265) */
#define DL_INTENDED_COMZ_PER_EPISODE 8   /* Q4 off: the COM-z re-anchoring of reset_model applies to the reset step for the current episode only (rounds 1-5 of this library);
                                            default: adjust_COM_Z_pos mutates the walker's copy of the data set in place (base_ref_trajecs.py:126-127, mimic_env.py:555-557):
                                            every step keeps the offset of the last reset that landed on it, and a cursor rolling into that step reads it */

typedef struct dl_env_s* dl_handle;

const char* dl_last_error(void);
int dl_abi_version(void);

/* Hardware facts this library's hand-written wait states rest on, checked ON the device (no reference counterpart: the reference has no device code).
 *   dl_dpp_wait_states   how many wait states this build keeps between a VALU write of a VGPR and its read through DPP in its hand-written statements:
 *                        2 = libdrloco_hip.so, the default -- the gfx9 / CDNA ISA manual's number, as `s_nop 0` twice (no s_wakeup of another wave can shorten it);
 *                        1 = libdrloco_hip_dpp1.so (-DDL_DPP_WAIT=1) -- what gfx950 was measured to need (+2.5 % on the benchmark line).  dl_create of THAT build runs
 *                        dl_hw_probe once per device and process and refuses (DL_E_HIP, text names the other library) unless one state is proven enough.
 *   dl_hw_probe          ~0.1 s of micro kernels on `device` (-1: the current one), iters <= 0: 64.  out[8]:
 *                        [0..2] stale DPP reads with 0 / 1 / 2 wait states behind the write, summed over 7 producers x 6 DPP forms, alone and beside a wave that loops over
 *                               s_wakeup ([0] > 0 shows that the test can fail; the one-state build needs [1] == 0).  [1] and [2] also take six VALU-writes-an-SGPR ->
 *                               VALU-reads-it pairs (v_cmp -> v_cndmask, v_readlane -> v_writelane ...) beside s_wakeup: gfx940+ asks for two states there, hipcc pads with ONE
 *                               `s_nop 1`, worth one state beside an s_wakeup (gfx950 measured: no stale read even without a wait, profiles/r06_sgpr_wait.txt);
 *                        [3..5] stale rows of a v_mfma_f32_4x4x1 result read 8 wait states later beside an s_wakeup loop, the wait written as ONE `s_nop 7` ([3] > 0: another
 *                               wave's s_wakeup ends an s_nop after one state), as 8 x v_nop ([4] == 0: what the policy kernels wait with), as two s_nop ([5] == 0);
 *                        [6] lane-reads per cell, [7] cells (84 DPP + 6 SGPR).
 * The Python loader (drloco_amd/lib.py) loads the default build, asks it for the probe, and switches to the one-state build only on that evidence. */
int dl_dpp_wait_states(void);
int dl_hw_probe(int32_t device, int32_t iters, uint64_t* out);

/* MimicWalker3dEnv.__init__ x N  (mimic_walker3d.py:33-40, mimic_env.py:19-57).
 * device: HIP device ordinal. */
int dl_create(const dl_model_desc* model, const dl_refs_desc* refs, const dl_config* cfg,
              int32_t n_envs, int32_t device, dl_handle* out);
int dl_destroy(dl_handle h);
int32_t dl_num_envs(dl_handle h);
int32_t dl_obs_dim(dl_handle h);
int32_t dl_act_dim(dl_handle h);
int32_t dl_real_size(dl_handle h); /* 4 or 8: element size of state arrays */

/* MujocoEnv.reset -> MimicEnv.reset_model (mimic_env.py:526-572) for the walkers whose
 * mask byte is non-zero (NULL = all).  init_step/init_pos (int32[N], NULL = draw from the
 * counter-based RSI stream) inject the two random draws of get_random_init_state
 * (straight_walk_trajecs.py:460-474).  obs_out: float[N, obs_dim], rows of unmasked walkers
 * are left untouched. */
int dl_reset(dl_handle h, const uint8_t* mask, const int32_t* init_step, const int32_t* init_pos,
             float* obs_out, void* stream);

/* MimicEnv.activate_evaluation (mimic_env.py:245-249) for every walker of the handle: later resets
 * use _get_deterministic_init_state (straight_walk_trajecs.py:237-265, incl. quirk Q3: the state is
 * read from step 0's table at 75 % of step k's length, k cycling 0..19) instead of RSI. */
int dl_set_eval(dl_handle h, int32_t on);

/* One control step of every walker: MimicEnv.step (mimic_env.py:60-126) followed by the
 * vec-env auto-reset (SubprocVecEnv worker: terminal_observation + reset).
 *   actions   float[N, nu]   policy outputs (unclipped)
 *   obs       float[N, obs]  next observation (post-reset row for finished walkers)
 *   rew       float[N]
 *   done      uint8[N]
 *   term_obs  float[N, obs]  or NULL: info['terminal_observation'] rows for finished walkers
 *   rew_terms float[N, 3]    or NULL: pos_rew, vel_rew, com_rew (mimic_env.py:645) */
int dl_step(dl_handle h, const float* actions, float* obs, float* rew, uint8_t* done,
            float* term_obs, float* rew_terms, void* stream);

/* T control steps with pre-generated actions (synthetic fixed-length rollout; no policy):
 *   actions float[T, N, nu]; obs float[T, N, obs]; rew float[T, N]; done uint8[T, N].
 * Same results as T calls of dl_step.  Because no action depends on an observation, the 16-lane kernels take up to 512
 * control steps per launch (walker state stays in registers in between; a launch lasts as long as the wave with the
 * largest sum over its steps rather than paying the slowest wave of every step). */
int dl_rollout_fixed(dl_handle h, int32_t T, const float* actions, float* obs, float* rew,
                     uint8_t* done, void* stream);

/* Parity hooks: SoA state in the handle's real type (dl_real_size):
 * qpos[nv,N], qvel[nv,N], qacc_warmstart[nv,N], cursor int32[DL_CUR_WORDS,N].  NULL = skip. */
int dl_get_state(dl_handle h, void* qpos, void* qvel, void* qacc_warm, int32_t* cursor,
                 double* walked, void* stream);
int dl_set_state(dl_handle h, const void* qpos, const void* qvel, const void* qacc_warm,
                 const int32_t* cursor, const double* walked, void* stream);

/* Quirk Q4's per-walker record: the COM-z offset each reference step carries from the last reset that landed on it (adjust_COM_Z_pos, base_ref_trajecs.py:126-127;
 * every SubprocVecEnv worker owns a copy of the data set, so the record is per walker): real[n_steps, N] in the handle's real type, zero at dl_create.  The get / set pair
 * completes dl_get_state / dl_set_state for parity tests and for carrying an environment over (a walker's current offset is re-read from the record by dl_set_ref_offsets). */
int dl_get_ref_offsets(dl_handle h, void* z_offsets, void* stream);
int dl_set_ref_offsets(dl_handle h, const void* z_offsets, void* stream);

/* Forward dynamics only (mj_forward) at the current state with the given ctrl (real[nu,N],
 * already in torque units): writes qacc real[nv,N]; ncon/nefc/niter int32[N] may be NULL. */
int dl_forward(dl_handle h, const void* ctrl, void* qacc, int32_t* ncon, int32_t* nefc,
               int32_t* niter, void* stream);

/* MimicEnv.dynamics_randomization (mimic_env.py:492-524): a stub in the reference that only lists the intended fields.
 * Built here for the fields BASELINE config 5 names: one scale per walker for all body masses and inertias
 * (body_mass, body_inertia) and the sliding friction of the floor (geom_friction; a contact uses the larger of the
 * two geoms' coefficients).  float[N] device arrays, NULL = leave unchanged.  16-lane kernels only. */
int dl_set_randomization(dl_handle h, const float* mass_scale, const float* floor_friction, void* stream);
/* [3P] xfrc_applied on the torso: world-frame force float[N, 3] (device) acting at the torso's centre of mass during
 * every following mj_step until changed; NULL = no force.  16-lane kernels only. */
int dl_set_push(dl_handle h, const float* force, void* stream);
/* The same push as a per-walker SCHEDULE kept on the device, so that launches covering many control steps
 * (dl_rollout_fixed) need no host round trip per step: force float[N, 3], phase int32[N] (device); during the k-th control
 * step after this call walker w is pushed iff (k + phase[w]) % period < duration.  force == NULL switches the schedule off
 * and clears the push.  (BASELINE config 5's "50 N push perturbations"; the reference's hook is a stub, mimic_env.py:492-524.) */
int dl_set_push_schedule(dl_handle h, const float* force, const int32_t* phase, int32_t period, int32_t duration,
                         void* stream);

/* Launch form of the step kernel in float32 (16 lanes per walker; both walkers since ABI 6): on = 1 launches workgroups of eight waves for
 * sixteen walkers -- four dynamics waves and four partner waves that work one forward evaluation AHEAD of them (kinematics, mass matrix and the
 * configuration half of the constraint stage of the next RK4 stage while the dynamics wave solves this one; two waves per SIMD, registers capped
 * at 256, 153 / 157 KB of LDS per workgroup) -- instead of one wave per four walkers.  36.1 against 30.1 M env-steps/s at 4096 straight walkers,
 * 16.3 against 13.1 M for the 19-dof walker; nothing else fits on the GPU next to it, so callers that overlap other kernels with the step (several
 * handles with a policy in the loop) keep the default, 0.  Same algorithm; the float32 results of the two forms may differ in the last bit (two
 * instantiations).  Returns DL_E_INVAL for float64 / one lane per walker. */
int dl_set_split(dl_handle h, int32_t on);

/* Device faults.  The reference turns a diverging simulation into an ended episode (MujocoException -> reward 0, done, reset:
 * mimic_env.py:86-91) and lets every other error raise; the kernels do the same.  The one failure a launch can detect but not repair is the
 * hand-over between the two waves of a split workgroup running out of its (bounded, ~30 ms) poll budget: the wave then ors a reason into the
 * handle's fault word (host-pinned memory: 1 = a dynamics wave gave up on its constraint wave, 2 = a constraint wave gave up waiting for a
 * request), takes no further part in the protocol, and its four walkers take the exception path on every remaining step of the launch --
 * never a silent continuation on stale constraint rows.  The word is sticky: dl_step, dl_rollout_fixed, dl_rollout_policy, dl_get_state and
 * dl_profile_read return DL_E_FAULT (text in dl_last_error) as soon as a completed launch has left it set.  dl_fault_check reads it (after the
 * caller has synchronised with the stream it sees every fault of the enqueued work); *code may be NULL.  dl_fault_clear resets it (then
 * dl_reset the walkers). */
int dl_fault_check(dl_handle h, int32_t* code);
int dl_fault_clear(dl_handle h);

/* MimicEnv.do_terminate_early (mimic_env.py:652-702; the reference defines it but its call in step() is commented
 * out, :113-118) at the current state of every walker: flags int32[N, 4] device =
 * {terminate, COM height too low (< 0.75), trunk angle exceeded, |COM y| > 0.2}.  Straight walker only. */
int dl_terminate_early(dl_handle h, int32_t* flags, void* stream);

/* Monitor attributes (drloco/mujoco/monitor_wrapper.py:88-133) kept per walker on device.
 * name: one of ep_len_smoothed, ep_ret_smoothed, mean_reward_smoothed, moved_distance,
 * mean_ep_pos_rew_smoothed, mean_ep_vel_rew_smoothed, mean_ep_com_rew_smoothed,
 * mean_abs_ep_torque_smoothed; diverged_steps: control steps that took the exception path (MujocoException -> reward 0, done, double reset; mimic_env.py:86-91) since dl_create;
 * first_ep_len, first_ep_moved, first_ep_ret: the FIRST episode a walker finished since dl_create or the last dl_reset of all
 * walkers (mask NULL), measured as TrainingMonitor.eval_walking measures an episode (drloco/common/callback.py:300-317: length incl. the terminal step;
 * walked distance and reward sum without it; 0 = no episode finished yet).  out: double[N] device. */
int dl_stats_snapshot(dl_handle h, const char* name, double* out, void* stream);

/* Measurement hooks (no reference counterpart): enable = k > 0 brackets every k-th launch of the
 * dominant kernel (the fused env-step kernel) by HIP events on its launch stream (k = 1: every
 * launch; event packets between kernels widen the launch gap by a few microseconds each, so a
 * throughput run samples); 0 switches it off.  dl_profile_read waits for the events and returns
 * the summed duration and the number of bracketed launches. */
int dl_profile(dl_handle h, int32_t enable);
int dl_profile_read(dl_handle h, double* total_ms, int32_t* launches);
/* control steps covered by the launches the last dl_profile_read reported (dl_rollout_fixed takes up to 512 control steps per
 * launch of the 16-lane kernel, dl_step one) */
int dl_profile_steps(dl_handle h);
/* launch geometry of the last bracketed launch: out[3] = {threads of the grid, threads per workgroup, dynamic LDS bytes} -- what rocprofv3's kernel trace calls Grid_Size,
 * Workgroup_Size, LDS_Block_Size.  A committed counter pass (profiles/traffic_*.json) records them; bench.py replays its figures only while the device code AND this geometry
 * are the ones that pass measured. */
int dl_profile_launch_config(dl_handle h, int32_t* out);

/* ---- reductions of the SB3 layer (stable-baselines3==1.0, docs/conda_env.yml:30) ---- */

/* RunningMeanStd.update (VecNormalize): Chan parallel update of (mean[D], var[D], count[1])
 * (double, device) with the batch x float[B, D]. */
int dl_moments_update(double* mean, double* var, double* count, const float* x, int32_t B,
                      int32_t D, void* stream);
/* VecNormalize.normalize_obs in place: clip((x - mean)/sqrt(var + eps), +-clip). */
int dl_normalize_obs(float* x, const double* mean, const double* var, int32_t B, int32_t D,
                     double eps, double clip, void* stream);
/* VecNormalize.step_wait reward branch: ret = ret*gamma + r; ret_rms.update(ret);
 * r = clip(r/sqrt(var+eps), +-clip); ret[done] = 0.  ret: double[B] device. */
int dl_normalize_reward(float* rew, double* ret, const uint8_t* done, double* ret_mean,
                        double* ret_var, double* ret_count, int32_t B, double gamma, double eps,
                        double clip, void* stream);
/* VecNormalize.step_wait for one batch in two launches (the fused form of the three calls above):
 *   obs_rms.update(obs); obs_out = clip((obs - mean)/sqrt(var + eps), +-clip_obs);
 *   ret = ret*gamma + rew; ret_rms.update(ret); rew_out = clip(rew/sqrt(ret_var + eps), +-clip_rew); ret[done] = 0.
 * flags: 1 update the observation moments (training), 2 normalise observations, 4 advance ret and update its
 * moments (training), 8 normalise rewards, 16 reduce with 32 blocks instead of one workgroup (for launches on a side stream
 * under other kernels), 32 reduce in the "blocked" order -- sums over blocks of 16 rows, blocks over <= 8 groups, groups in order: the order
 * dl_collect_rollouts' persistent kernel follows, so that the two agree bit for bit (all forms are deterministic, their summation orders differ), 64 the
 * moments were already advanced for this batch (dl_vn_local_sums / dl_vn_merge_sums below): only normalise and reset ret[done].  obs/rew are not modified (get_original_obs / get_original_reward);
 * obs_out/rew_out may be rollout-buffer slots.  workspace: device memory, DL_VN_WORKSPACE_BYTES(D) bytes,
 * zero-initialised once by the caller and owned by this call sequence. */
#define DL_VN_WORKSPACE_BYTES(D) (8 * (2 * 32 * ((D) + 1) + 2))   /* used by the multi-block reduction (flags bit 16) only */
int dl_vecnormalize_step(const float* obs, const float* rew, const uint8_t* done, double* obs_mean,
                         double* obs_var, double* obs_count, double* ret, double* ret_mean, double* ret_var,
                         double* ret_count, int32_t B, int32_t D, double gamma, double eps, double clip_obs,
                         double clip_rew, int32_t flags, float* obs_out, float* rew_out, void* workspace,
                         void* stream);

/* VecNormalize.step_wait's moment update with the batch spread over several ranks (one process per GPU), EXACT per control step -- what SB3's single
 * process computes from all N walkers, here from N / world walkers per rank and one all-reduce of 2 (D + 1) doubles per control step:
 *   dl_vn_local_sums     this rank's shifted sums [2 (D + 1)] (column k: sum of (x - mean_k), sum of squares; column D: the discounted returns, which
 *                        are advanced here) in the blocked summation order; the running moments are only read -- they are equal on all ranks;
 *   (caller)             all-reduce (sum) of `sums` over the ranks (RCCL / gloo);
 *   dl_vn_merge_sums     RunningMeanStd.update_from_moments with the GLOBAL batch size; advances the counts;
 *   dl_vecnormalize_step with flags | 64 ("moments already merged") normalises observations / rewards and resets ret[done].
 * HipVecNormalize(sync='per_step') does this; the default (sync='per_rollout') advances the moments per rank and merges them exactly between
 * rollouts (HipVecNormalize.sync_moments), which needs no collective inside a rollout. */
int dl_vn_local_sums(const float* obs, const float* rew, const double* obs_mean, double* ret, const double* ret_mean,
                     int32_t B, int32_t D, double gamma, int32_t flags, double* sums, void* stream);
int dl_vn_merge_sums(const double* sums, int64_t B_global, double* obs_mean, double* obs_var, double* obs_count,
                     double* ret_mean, double* ret_var, double* ret_count, int32_t D, int32_t flags, void* stream);

/* RolloutBuffer.compute_returns_and_advantage: arrays are [T, N] time-major float;
 * ep_start[t] = "obs_t starts an episode"; last_val float[N]; last_done uint8[N]. */
int dl_gae(const float* rew, const float* val, const uint8_t* ep_start, const float* last_val,
           const uint8_t* last_done, float gamma, float lam, int32_t T, int32_t N, float* adv,
           float* ret, void* stream);
/* sums for PPO advantage normalisation: out[0] = sum(a), out[1] = sum(a^2), out[2] = n
 * (double[3] device; all-reduced over ranks by the caller before dl_adv_normalize).  Deterministic (fixed summation
 * order, no floating-point atomics).  workspace: device memory, DL_ADV_WORKSPACE_BYTES, zero-initialised once by
 * the caller and owned by it (one per stream that may run this call concurrently). */
#define DL_ADV_WORKSPACE_BYTES ((2 * 512 + 2) * 8)
int dl_adv_stats(const float* adv, int64_t n, double* out3, void* workspace, void* stream);
/* a = (a - mean)/(std_unbiased + 1e-8) from the (all-reduced) sums. */
int dl_adv_normalize(float* adv, int64_t n, const double* sums3, void* stream);

/* ---- policy forward pass of the rollout loop ----
 * Parameters of the reference's CustomActorCriticPolicy (drloco/custom/policies.py:13-51) in torch's nn.Linear
 * layout [out][in], float32, DEVICE pointers: shared trunk obs_dim -> hidden -> hidden (tanh), action_net
 * hidden -> act_dim, value_net hidden -> 1, log_std[act_dim].  hidden: multiple of 64, <= 512; obs_dim <= 48; act_dim <= 15. */
typedef struct dl_policy_params {
    const float* w1; const float* b1;   /* [hidden, obs_dim], [hidden] */
    const float* w2; const float* b2;   /* [hidden, hidden], [hidden] */
    const float* wa; const float* ba;   /* [act_dim, hidden], [act_dim] */
    const float* wv; const float* bv;   /* [1, hidden], [1] */
    const float* log_std;               /* [act_dim] */
    int32_t obs_dim, hidden, act_dim;
} dl_policy_params;
/* SB3 1.0 ActorCriticPolicy.forward for n observations (already normalised): actions float[n, act_dim]
 * (unclipped, as the rollout buffer stores them), values float[n], log_probs float[n].
 * eps: float[n, act_dim] standard-normal draws, or NULL = counter-based stream keyed by (seed, counter, index_base +
 * row, action index); deterministic != 0 returns the mean action (evaluation). */
int dl_policy_forward(const dl_policy_params* params, const float* obs, int32_t n, const float* eps,
                      uint64_t seed, uint64_t counter, int32_t index_base, int32_t deterministic,
                      float* actions, float* values, float* log_probs, void* stream);

/* The same forward pass reading the weights from a k-chunk-major copy (hidden = 512, 256 or 128): in torch's [out][in] layout the 16 lanes of a quarter-wave
 * read 16 bytes from 16 different rows (64 cache lines per wave instruction), packed they read one contiguous 256-byte run -- 29 instead of 40 us for
 * 4096 rows, bit-identical results.  dl_policy_pack writes the copy (DL_POLICY_PACKED_FLOATS(512) floats, device memory; call it again whenever
 * the weights change -- dl_collect_rollouts / dl_rollout_policy do so themselves, once per call); packed == NULL = dl_policy_forward. */
#define DL_POLICY_PACKED_FLOATS(hidden) ((size_t)(hidden) * (hidden) + (size_t)48 * (hidden) + (size_t)16 * (hidden))
int dl_policy_pack(const dl_policy_params* params, float* packed, void* stream);
int dl_policy_forward_packed(const dl_policy_params* params, const float* packed, const float* obs, int32_t n,
                             const float* eps, uint64_t seed, uint64_t counter, int32_t index_base,
                             int32_t deterministic, float* actions, float* values, float* log_probs, void* stream);

/* The same forward pass in its four-rows-per-wave-pair form (packed weights, hidden = 512, 256 or 128): a workgroup of two waves per four rows on
 * v_mfma_f32_4x4x1_16B_f32, every sum in the order of dl_policy_forward -- bit-identical outputs (tests/test_gpu_persistent.py::test_policy_pair_form_is_the_forward_pass_bit_for_bit).  It is the building
 * block that lets a wave pair of the persistent rollout kernel evaluate the policy of its own four walkers; stand-alone it is a reference for that. */
int dl_policy_forward_pair(const dl_policy_params* params, const float* packed, const float* obs, int32_t n,
                           const float* eps, uint64_t seed, uint64_t counter, int32_t index_base,
                           int32_t deterministic, float* actions, float* values, float* log_probs, void* stream);

/* VecNormalize state as dl_vecnormalize_step takes it, bundled for dl_rollout_policy (all DEVICE pointers). */
typedef struct dl_vecnorm_state {
    double* obs_mean; double* obs_var; double* obs_count;     /* [D], [D], [1] */
    double* ret; double* ret_mean; double* ret_var; double* ret_count;   /* [N], [1], [1], [1] */
    void* workspace;                                          /* DL_VN_WORKSPACE_BYTES(D), zero-initialised once */
    double gamma, eps, clip_obs, clip_rew;
    int32_t flags;                                            /* as dl_vecnormalize_step */
} dl_vecnorm_state;

/* K consecutive dl_vecnormalize_step calls (SB3 1.0 VecNormalize.step_wait, K control steps of a fixed-action rollout) in six small launches
 * instead of 2 K: obs float[K, B, D], rew float[K, B], done uint8[K, B] (time-major, contiguous, DEVICE); obs_out / rew_out: DEVICE arrays of K
 * pointers, the destination of every step (rollout-buffer slots).  The moments advance exactly as K single steps would advance them, up
 * to rounding: the shifted sums of all K batches use the moments at the start of the run as their shift (the single steps use the moments
 * of the step before), the merges are the same and run in step order; vn->flags bit 16 is irrelevant here.  workspace: caller-owned
 * device memory of DL_VN_STEPS_WORKSPACE_BYTES(K, B, D), no initialisation needed. */
#define DL_VN_STEPS_WORKSPACE_BYTES(K, B, D) (8 * ((size_t)(K) * (B) + (size_t)(K) * 32 * ((D) + 1) * 2 + (size_t)(K) * ((D) + 1) * 2))
int dl_vecnormalize_steps(const dl_vecnorm_state* vn, int32_t K, const float* obs, const float* rew, const uint8_t* done,
                          int32_t B, int32_t D, float* const* obs_out, float* const* rew_out, void* workspace,
                          void* stream);

/* SB3 1.0 OnPolicyAlgorithm.collect_rollouts for T steps in ONE call (constructed at
 * drloco/train.py:110-118, the loop SB3 runs between two PPO updates): for t = 0..T-1
 *     actions[t], values[t], log_probs[t] = policy.forward(observations[t])        (dl_policy_forward, counter0 + t)
 *     raw obs, raw reward, done           = env.step(actions[t])                   (dl_step incl. auto-reset)
 *     observations[t+1], rewards[t]       = VecNormalize.step_wait(...)            (dl_vecnormalize_step)
 *     episode_starts[t+1]                 = done
 * with every producer writing straight into the rollout-buffer arrays (time-major, the layout dl_gae reads):
 *   observations float[T, N, obs] (row 0 = the observation that opens the rollout, filled by the caller),
 *   actions float[T, N, nu], values/log_probs/rewards float[T, N], episode_starts uint8[T, N] (row 0 by the caller).
 * The observation and done flags after the last step go to next_obs float[N, obs] / next_done uint8[N] (they open
 * the next rollout; SB3's _last_obs / _last_episode_starts).  raw_obs float[N, obs] / raw_rew float[N]: scratch that
 * holds the un-normalised outputs of the last step (VecNormalize.get_original_obs / get_original_reward).
 * 4 T launches are enqueued on `stream`; nothing returns to the host in between. */
int dl_rollout_policy(dl_handle h, const dl_policy_params* policy, uint64_t seed, uint64_t counter0,
                      int32_t index_base, const dl_vecnorm_state* vn, int32_t T, float* observations,
                      float* actions, float* values, float* log_probs, float* rewards,
                      uint8_t* episode_starts, float* next_obs, uint8_t* next_done, float* raw_obs,
                      float* raw_rew, void* stream);

/* dl_rollout_policy with a choice of execution form (`mode`):
 *   0                                   T x 3 launches (= dl_rollout_policy);
 *   DL_ROLLOUT_PERSISTENT               ONE launch for the whole rollout: a persistent workgroup of eight waves per sixteen walkers runs, per
 *       control step, the policy forward of its own rows (matrix cores), MimicEnv.step of its walkers (the split-workgroup step kernel's code)
 *       and VecNormalize's moment update through one grid-wide exchange.  Exact SB3 semantics (every step normalises with the moments of all
 *       walkers up to that step); bit-identical to mode 0 when `vn->flags` selects the blocked reduction order (bit 32) and dl_set_split is on.
 *       Needs: float32, 16 lanes per walker, hidden = 512, 256 or 128 (eight waves x 4 / 2 / 1 tiles; the reference's sizes are a config, drloco/config/hypers.py:98-99), at most 128 walkers per CU for the straight walker (32768 on an MI355X; above 16 per CU a workgroup takes several blocks of sixteen walkers per control step) -- query with
 *       dl_rollout_persistent_ok (1 / 0); DL_E_INVAL otherwise.  The 19-dof walker (ABI 6): one block per workgroup, i.e. at most 16 walkers per CU (4096).  EXCLUSIVE GPU: the grid-wide exchange needs every workgroup of the launch resident at
 *       the same time (one per CU), so no other process, stream or handle may hold CUs while it runs -- dl_rollout_persistent_ok only checks the
 *       walker count, it cannot see other users of the device.  A grid exchange that does not complete within its (bounded, ~2 s) poll budget raises
 *       the handle's fault word (DL_FAULT_GRID_TIMEOUT = 4) and the workgroup stops: the call has returned DL_OK by then, so the caller must
 *       synchronise with the stream and call dl_fault_check BEFORE it reads the buffers (drloco_amd.rollout.HipRolloutBuffer.collect_rollouts does,
 *       and in its automatic mode clears the fault, resets the walkers and redoes the rollout with mode 0).
 *   DL_ROLLOUT_PERSISTENT | DL_ROLLOUT_MOMENTS_PER_ROLLOUT   opt-in relaxation: observations and rewards of the whole rollout are normalised
 *       with the moments at its START, workgroups exchange nothing during the rollout (they run free; a rollout lasts as long as its slowest
 *       workgroup's sum of steps), and ONE exact Chan merge of all T x N samples follows -- RunningMeanStd.update fed the rollout as one batch.
 *       Not SB3's per-step update; the same relaxation the cross-rank merge C3 applies.
 *       Kernel: every wave PAIR takes its own four walkers through the rollout (k_rollout_pairs: the policy on v_mfma_f32_4x4x1 chains, no meeting of the
 *       workgroup's pairs per step).  A pair whose two waves fail to meet within the poll budget raises DL_FAULT_SRV_TIMEOUT and stops; its rows of the
 *       buffers stay incomplete -- same rule as above: fault check before reading.  (ONE poll budget serves both persistent kernels' waits -- the grid exchange of the
 *       exact form and this pair meeting: ~seconds by default; the test hook dl_debug_set_grid_spin sets both.  A wave decides "my partner did not come" from a last
 *       read of the partner's counter after the budget has run out, never from the budget alone.)
 *   ... | DL_ROLLOUT_WORKGROUP_TILES   the same relaxation on the exact form's kernel (k_rollout_persistent: sixteen-row policy tiles on v_mfma_f32_16x16x4, the
 *       workgroup's four pairs meet at every control step): the selectable fallback for the pair kernel, about 4 % slower; same buffers, the same moments after
 *       the merge up to the grouping of the float64 sums. */
#define DL_ROLLOUT_PERSISTENT 1
#define DL_ROLLOUT_MOMENTS_PER_ROLLOUT 2
#define DL_ROLLOUT_WORKGROUP_TILES 4
#define DL_ROLLOUT_DETERMINISTIC 8   /* any form: the policy returns the mean action (predict(deterministic=True), drloco/common/callback.py:297): evaluation as one call */
int dl_rollout_persistent_ok(dl_handle h, const dl_policy_params* policy);
int dl_collect_rollouts(dl_handle h, const dl_policy_params* policy, uint64_t seed, uint64_t counter0,
                        int32_t index_base, const dl_vecnorm_state* vn, int32_t T, float* observations,
                        float* actions, float* values, float* log_probs, float* rewards,
                        uint8_t* episode_starts, float* next_obs, uint8_t* next_done, float* raw_obs,
                        float* raw_rew, int32_t mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DRLOCO_HIP_H */
