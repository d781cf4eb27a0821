#!/usr/bin/env python3
"""bench.py -- throughput of the DRLoco hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... -- or plainly as above:
   without WORLD_SIZE in the environment the process starts its N ranks itself as a child torch.distributed.run and forwards rank 0's line)

Workload (BASELINE.json configs[1], per GPU; configs[2] is the same per GPU at N = 8):
  straight_walking 3D walker, 4096 parallel walkers per GPU, fixed 512-step synthetic rollout.
One "step" of this benchmark = ONE 512-step rollout of all walkers of a rank:
  per control step   MimicEnv.step (5 RK4 mj_steps + mocap cursor + imitation reward + observation + termination +
                     Monitor statistics + auto-reset/RSI) -> raw obs/reward, done flags into the next episode_starts
                     slot of the rollout buffer; VecNormalize.step_wait (moments update + observation/reward
                     normalisation) -> the rollout buffer's observations[t+1] / rewards[t] slots;
  per rollout        GAE(lambda) return/advantage scan, advantage statistics + normalisation
                     (RCCL all-reduce of 3 doubles when N > 1 -- the only collective), VecNormalize moment merge across ranks.
The actions are pre-generated, so nothing waits for an observation: the env steps go through dl_rollout_fixed, whose
16-lane kernel takes a whole run of control steps per launch (walker state in registers; a launch lasts as long as the
wave with the largest SUM over its steps, not the sum of every step's slowest wave) -- ONE launch of 512 steps per rollout --
and the rollout's normalisations follow as one dl_vecnormalize_steps call (six small launches; raw outputs in a ring, one event pair per run).  --no-overlap keeps one launch per control step on one stream; --policy puts
the fused policy into the loop (dl_collect_rollouts: the whole rollout as ONE persistent launch -- policy forward, env step and VecNormalize's
moment exchange per control step inside the kernel -- or, --rollout-form launches, three launches per control step; --moments per_rollout is
the opt-in relaxation of the persistent form).  Same results in all forms (tests/test_gpu_bench_shapes.py, tests/test_gpu_persistent.py).
After the timed region the run checks its own output (fault word, finite buffers, reward range, episode ends: `self_check` in the JSON line).
Actions are pre-generated a_t = clip(0.5*N(0,1), -1, 1) and values synthetic, both from a counter-based generator keyed by the global walker
index (tape_normal: every rank draws exactly its own columns), RSI comes from the counter-based stream keyed by the global walker index.  Inputs are resident in
HBM before the timed region.  value = walkers * 512 * K * N / time  [env-steps/s, whole job].
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 664          # 336 B read + 328 B written per walker and control step
ALGO_BYTES_PER_ENV_STEP_LOCO3D = 916   # the same count for the 19-dof walker: 456 B read (3 x 19 state + 13 action + 6 cursor + 38 reference words) + 460 B written (57 state + 6 cursor + 47 obs + 5)
HBM_PEAK_GBS = 8000.0                  # /opt/skills/guides/MI355X_MICROARCH.md


def kernel_code_sha16():
    """Identity of the kernel build a profile was taken with: sha256 over the device code object embedded in the built library
    (drloco_amd.lib.device_code_sha16).  Round 4 hashed the SOURCES, comments included: a comment-only commit forced a re-stamp of every
    PMC pass; the code object only moves when a kernel does."""
    from drloco_amd import lib
    return lib.device_code_sha16()


def _shr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)          # logical shift on int64 tensors


def _mix64(x):
    """splitmix64's finaliser on int64 tensors (torch integer arithmetic wraps)."""
    x = x + (-7046029254386353131)                    # 0x9E3779B97F4A7C15
    x = (x ^ _shr(x, 30)) * (-4658895280553007687)    # 0xBF58476D1CE4E5B9
    x = (x ^ _shr(x, 27)) * (-7723592293110705685)    # 0x94D049BB133111EB
    return x ^ _shr(x, 31)


def tape_normal(seed, stream, T, idx0, n, width, device):
    """Standard-normal tape [T, n, width] (or [T, n] for width 0) for the walkers idx0 .. idx0 + n - 1 from a counter-based generator keyed by
    (seed, stream, step, GLOBAL walker index, component): a rank generates exactly its own columns, and a walker sees the same numbers
    whatever the number of GPUs -- as it does for its RSI draws (the round-2 form drew the whole global tensor on every rank)."""
    import math
    import torch
    w = max(width, 1)
    t = torch.arange(T, dtype=torch.int64, device=device).view(T, 1, 1)
    i = torch.arange(idx0, idx0 + n, dtype=torch.int64, device=device).view(1, n, 1)
    a = torch.arange(w, dtype=torch.int64, device=device).view(1, 1, w)
    key = _mix64(torch.tensor(int(seed) * 1000003 + int(stream), dtype=torch.int64, device=device))
    r = _mix64(key ^ _mix64(((t << 32) | i) * 64 + a))
    u1 = (_shr(r, 40) + 1).to(torch.float32) * (1.0 / 16777216.0)        # (0, 1]
    u2 = (r & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)            # [0, 1)
    z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)
    return z if width else z[..., 0]


def cpu_baseline(n_envs, n_steps, seed=4321):
    """The oracle (a scalar float64 port of the same path) timed on one host core."""
    import numpy as np
    from drloco_amd import abi, mocap, models
    from oracle import oracle as O
    model, refs = models.make_model(), mocap.RefTable.load()
    env = O.OracleEnv(model, refs, abi.default_config(seed=1234), n_envs)
    env.reset()
    rng = np.random.default_rng(seed)
    acts = np.clip(0.5 * rng.standard_normal((n_steps, n_envs, 8)), -1, 1)
    t0 = time.perf_counter()
    for t in range(n_steps):
        env.step(acts[t])
    dt = time.perf_counter() - t0
    return dict(value=n_envs * n_steps / dt, unit='env-steps/s', cores=1, kind='port',
                sample=f'{n_envs} walkers x {n_steps} control steps, same action distribution, oracle/dl_oracle.c on 1 host core '
                       f'({os.cpu_count()} cores present), {dt:.1f} s')


def _subproc_worker(conn, seed):
    import numpy as np
    from drloco_amd import abi, mocap, models
    from oracle import oracle as O
    env = O.OracleEnv(models.make_model(), mocap.RefTable.load(), abi.default_config(seed=seed), 1)
    conn.send(env.reset())
    while True:
        a = conn.recv()
        if a is None:
            break
        conn.send(env.step(a)[:3])
    conn.close()


def cpu_baseline_subproc(n_envs=4, n_steps=2048, seed=4321, rollouts=3, warmup=1):
    """BASELINE configs[0] in the reference's own process structure (SB3 SubprocVecEnv, drloco/common/utils.py:97-134): one
    worker process per env stepping the oracle, pipes to a learner process that only distributes actions.  BASELINE.md 2.4: one warm-up
    rollout, then >= 3 timed 2048-step rollouts (the walkers keep running from rollout to rollout, as in training)."""
    import multiprocessing as mp
    import numpy as np
    ctx = mp.get_context('fork')
    pipes, procs = [], []
    for i in range(n_envs):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_subproc_worker, args=(b, 1234 + i), daemon=True)
        p.start(); pipes.append(a); procs.append(p)
    for c in pipes:
        c.recv()
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    # learner side as in the reference: VecNormalize (running moments + normalisation) every step, GAE at the end of the rollout
    om, ov, oc = np.zeros(29), np.ones(29), 1e-4
    rm, rv, rc = np.zeros(1), np.ones(1), 1e-4
    ret = np.zeros(n_envs)
    times = []
    for rollout in range(warmup + rollouts):
        acts = np.clip(0.5 * rng.standard_normal((n_steps, n_envs, 1, 8)), -1, 1)
        vals = rng.standard_normal((n_steps + 1, n_envs)).astype(np.float32)
        rews = np.zeros((n_steps, n_envs), np.float32); starts = np.zeros((n_steps + 1, n_envs), np.uint8); starts[0] = 1
        t0 = time.perf_counter()
        for t in range(n_steps):
            for i, c in enumerate(pipes):
                c.send(acts[t, i])
            res = [c.recv() for c in pipes]
            obs = np.concatenate([r[0] for r in res]); rew = np.concatenate([r[1] for r in res]); done = np.concatenate([r[2] for r in res])
            oc = O.moments_update(om, ov, oc, obs)
            obs_n = np.clip((obs - om) / np.sqrt(ov + 1e-8), -10, 10)
            ret = ret * 0.99 + rew
            rc = O.moments_update(rm, rv, rc, ret[:, None])
            rews[t] = np.clip(rew / np.sqrt(rv[0] + 1e-8), -10, 10)
            ret[done != 0] = 0
            starts[t + 1] = done != 0
        O.gae(rews, vals[:n_steps], starts[:n_steps], vals[n_steps], starts[n_steps], 0.995, 0.95)
        if rollout >= warmup:
            times.append(time.perf_counter() - t0)
        del obs_n
    for c in pipes:
        c.send(None)
    for p in procs:
        p.join(timeout=5)
    dt = sum(times)
    return dict(value=n_envs * n_steps * rollouts / dt, unit='env-steps/s', cores=n_envs + 1,
                sample=f'{n_envs} worker processes x 1 env (oracle) + 1 learner process (VecNormalize every step, GAE at the end) over pipes; {warmup} warm-up + {rollouts} timed '
                       f'{n_steps}-step rollouts, ' + ' / '.join(f'{x:.2f}' for x in times) + ' s')


def _allcores_worker(barrier, q, seed, n_envs, n_steps):
    import numpy as np
    from drloco_amd import abi, mocap, models
    from oracle import oracle as O
    env = O.OracleEnv(models.make_model(), mocap.RefTable.load(), abi.default_config(seed=seed), n_envs)
    env.reset()
    acts = np.clip(0.5 * np.random.default_rng(seed).standard_normal((n_steps, n_envs, 8)), -1, 1)
    barrier.wait()
    t0 = time.time()
    for t in range(n_steps):
        env.step(acts[t])
    q.put((t0, time.time()))


def cpu_baseline_all_cores(n_envs=64, n_steps=256, max_procs=32):
    """The same scalar oracle on many host cores at once (one process per core, up to 32, independent walker shards, no learner): what
    the CPU path reaches on this box when the environments are the only cost.  (The GPU boxes report 256 cores but schedule a job on
    about a dozen: 128 processes gave 120 k env-steps/s, 11 x one core.)"""
    import multiprocessing as mp
    ctx = mp.get_context('fork')
    n_procs = max(1, min(os.cpu_count() or 1, max_procs))
    barrier, q = ctx.Barrier(n_procs), ctx.Queue()
    procs = [ctx.Process(target=_allcores_worker, args=(barrier, q, 1234 + i, n_envs, n_steps), daemon=True) for i in range(n_procs)]
    for p in procs:
        p.start()
    spans = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=5)
    dt = max(e for _, e in spans) - min(b for b, _ in spans)
    return dict(value=n_procs * n_envs * n_steps / dt, unit='env-steps/s', cores=n_procs,
                sample=f'{n_procs} processes x {n_envs} walkers x {n_steps} control steps (oracle, environments only), {dt:.1f} s')


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher around it: this process -- which has not imported torch, let alone touched a GPU -- starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <the same arguments>` as a CHILD (never an exec), passes the
    ranks' output through (rank 0 prints the one JSON line) and returns the child's exit code.  The reference starts its N environment
    processes with one call, too (drloco/common/utils.py:121-125)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_ranks), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--envs-per-gpu', type=int, default=4096)
    ap.add_argument('--rollout-len', type=int, default=512)
    ap.add_argument('--lanes', type=int, default=0, help='lanes per walker of the dynamics kernels: 0 auto (16), 1, 16')
    ap.add_argument('--walker', choices=['straight', 'loco3d'], default='straight', help='loco3d: BASELINE configs[3] (19-dof walker, synthetic loco3d table); not the benchmark configuration')
    ap.add_argument('--policy', action='store_true', help='not the benchmark configuration: put the fused device policy (dl_policy_forward, 29-512-512-{8,1}) into the loop instead of pre-generated actions/values')
    ap.add_argument('--hidden', type=int, default=512, choices=[64, 128, 256, 512], help='with --policy: the hidden size of the 2-layer trunk (the reference default is 512, drloco/config/hypers.py:98-99; 64 has the launch form only)')
    ap.add_argument('--randomize', action='store_true', help='not the benchmark configuration: BASELINE config 5 stress test -- per-walker mass scale U[0.8,1.2], floor friction U[0.5,1.1], 50 N horizontal pushes on the torso for 0.1 s every 2 s at a random phase (keyed by the global walker index)')
    ap.add_argument('--profile-every', type=int, default=1, help='bracket every k-th launch of the env-step kernel with HIP events (roofline.avg_launch_us); events between kernels cost launch gap, so the default samples')
    ap.add_argument('--runs', type=str, default='', help='control steps per dl_rollout_fixed call = per launch of the 16-lane kernel (each <= 512) in the policy-free configuration, e.g. 448,64; default: ONE launch for the rollout (the 16-lane kernels; tools/prof_step.py issues the same schedule for the PMC passes)')
    ap.add_argument('--handles', type=int, default=1, help='with --policy: split the walkers of a rank over this many env handles, each driving its policy -> step -> normalise chain on its own stream (drloco_amd/group.py); balanced single-step launches need >= 8192 walkers per GPU')
    ap.add_argument('--no-overlap', action='store_true', help='run dl_vecnormalize_step on the main stream after every dl_step instead of on a side stream under the next step')
    ap.add_argument('--vn-single-steps', action='store_true', help='normalise the steps of a fixed-action run one dl_vecnormalize_step at a time instead of with dl_vecnormalize_steps (six small launches per run)')
    ap.add_argument('--no-split', action='store_true', help='keep the one-wave-per-four-walkers launch form of the step kernel (dl_set_split 0); the default switches the split workgroup on where it exists (straight walker, float32, 16 lanes, one handle)')
    ap.add_argument('--rollout-form', choices=['auto', 'launches', 'persistent'], default='auto', help='with --policy: dl_collect_rollouts as three launches per control step or as ONE persistent launch per rollout (auto: persistent where it exists -- straight walker, float32, <= 128 walkers per CU)')
    ap.add_argument('--moments', choices=['per_step', 'per_rollout'], default='per_step', help="with --policy and the persistent form: 'per_rollout' is the opt-in relaxation (the rollout is normalised with its start-of-rollout moments, one exact merge at its end); not SB3's semantics")
    ap.add_argument('--checkpoint-moments', choices=['seat', 'free'], default='seat', help="with --checkpoint: 'seat' (default) = every rollout starts from the checkpoint's VecNormalize moments and advances them by its own samples; 'free' = the statistics run on from rollout to rollout as in training -- under a FIXED policy they drift and the walkers stop walking after ~100 rollouts (EXPERIMENTS.md round 6)")
    ap.add_argument('--checkpoint', type=str, default='', help="with --policy: a TRAINED policy instead of the random-init one -- 'walking' = the packaged drloco_amd/data/walking_policy.npz (examples/train_ppo.py, 8 M steps: the walkers reach the 3000-step episode limit), or the path of such a file (tools/pack_walking_ckpt.py).  Its VecNormalize moments are loaded and the walkers get a training env's step counter (quirk Q2), so the rollouts are what training looks like once the walkers WALK: feet on the ground, contact-rich (not the benchmark configuration)")
    ap.add_argument('--deterministic', action='store_true', help='with --policy: the mean action instead of a sample (DL_ROLLOUT_DETERMINISTIC; evaluation-style rollouts)')
    ap.add_argument('--solver-stats', action='store_true', help="switch the step kernel's solver diagnostics on (dl_debug_counters: a few stores per walker and control step inside the timed region) and report Newton iterations / constraint rows per forward evaluation in self_check")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--vn-sync', choices=['per_rollout', 'per_step'], default='per_rollout', help="with --policy on several ranks: 'per_step' = VecNormalize's moments advance with the batch of ALL ranks every control step (SB3's semantics across ranks: one all-reduce of 2 (obs_dim + 1) doubles per control step, host loop); default: per rank, merged exactly between rollouts")
    ap.add_argument('--dump', type=str, default='', help='after the run every rank saves what its LAST rollout produced (episode starts, action tape, raw step outputs, moments) to <path>.rank<r>.npz (tests/test_gpu_distributed.py compares ranks with a single-process run)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))       # plain `python bench.py --gpus N`: start the N ranks (nothing here has touched the GPU)

    import torch
    import torch.distributed as dist
    from drloco_amd import lib
    from drloco_amd.rollout import HipRolloutBuffer
    from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
    import ctypes as C

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and --gpus disagree (plain `python bench.py --gpus N` starts its own N ranks)')
    # the CPU baselines run BEFORE this process touches the GPU (the worker processes of the second one are forked)
    cpu_base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(256, 512)                  # ~12 s of CPU work on one host core
        try:        # configs[0] in the reference's process structure (reported next to it, not the baseline value)
            cpu_base['subproc_vec_env_4'] = cpu_baseline_subproc(4, 2048)
        except Exception as e:      # a box that cannot fork workers still gets its benchmark line
            cpu_base['subproc_vec_env_4'] = {'error': repr(e)}
        try:        # and the oracle on up to 32 host cores at once
            cpu_base['all_host_cores'] = cpu_baseline_all_cores()
        except Exception as e:
            cpu_base['all_host_cores'] = {'error': repr(e)}
    if local_rank >= torch.cuda.device_count():      # test rigs with fewer GPUs than ranks (DL_BENCH_BACKEND=gloo): share the last device
        if os.environ.get('DL_BENCH_BACKEND', 'nccl') == 'nccl':
            raise SystemExit(f'--gpus {args.gpus} on a box with {torch.cuda.device_count()} GPU(s): RCCL needs one GPU per rank (DL_BENCH_BACKEND=gloo shares a device, for tests)')
        local_rank = torch.cuda.device_count() - 1
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or 'RANK' in os.environ          # under torch.distributed.run the RCCL path runs even with one rank
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = os.environ.get('DL_BENCH_BACKEND', 'nccl')          # nccl = RCCL; gloo only to exercise the multi-rank code path on a box with one GPU
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device('cuda', local_rank)

    n, T = args.envs_per_gpu, args.rollout_len
    # the split workgroups fill the GPU: good for the policy-free rollout at any size and for a policy in the loop at the benchmark size;
    # with a policy and more walkers the one-wave form leaves room for the policy kernel next to the env steps (measured: 18.7 vs 20.0 M at 32 768)
    split = (not args.no_split) and args.lanes in (0, 16) and not (args.policy and (args.handles > 1 or n > 4096))
    # launch schedule of the policy-free rollout: ONE launch covers the rollout (<= 512 steps) and its normalisations follow as one
    # dl_vecnormalize_steps call.  (Round 2 ran 448 + 64 steps with the normalisations of the long launch "on the side stream under the short
    # one": the split workgroups -- two waves x 256 registers -- and the 19-dof kernel -- 508 registers -- hold every SIMD's whole register
    # file, so nothing ever ran next to them and the side stream's kernels only trickled in as workgroups retired; measured, one launch
    # against 448 + 64: 29.06 vs 28.77 M split, 27.44 vs 27.28 M one-wave straight walker, 12.82 vs 12.78 M 19-dof walker.)
    if args.runs:
        runs = [int(x) for x in args.runs.split(',')]
    elif args.lanes in (0, 16):
        runs = [min(T, 512)]
    else:
        runs = [T - 64, 64] if T >= 128 else [T]
    run_starts, t0 = [], 0
    for r in runs:
        r = min(r, 512, T - t0)
        if r > 0:
            run_starts.append((t0, r)); t0 += r
    if args.walker == 'loco3d':
        # BASELINE configs[3] (not the benchmark configuration): MimicWalker165cm65kg (19 dofs: 16 lane dofs + 3 replicated root translations)
        # on the synthetic stand-in for the missing loco3d_guoping.mat (8 clips), mixed-clip reference-state init
        from drloco_amd import mocap, models
        ang, vel = mocap.synthetic_loco3d(L=60000, seed=0)
        venv = HipVecEnv(models.WALKER_165CM, num_envs=n, device=local_rank, seed=1234, env_index_base=rank * n, refs=mocap.loco3d_table(ang, vel), lanes_per_walker=args.lanes)
    else:
        venv = HipVecEnv(num_envs=n, device=local_rank, seed=1234, env_index_base=rank * n, lanes_per_walker=args.lanes)
    if split:
        venv.set_split(True)          # dynamics waves + constraint waves (include/drloco_hip.h: dl_set_split)
    vn = HipVecNormalize(venv, sync=args.vn_sync if args.policy else 'per_rollout')
    if vn.sync == 'per_step':
        vn.blocked_reduce = True
    buf = HipRolloutBuffer(T, n, venv.obs_dim, venv.nu, dev, gamma=0.995, gae_lambda=0.95)
    # what the policy would have produced lives where it would have written it: in the rollout buffer.  The tapes are keyed by the GLOBAL
    # walker index (tape_normal: a counter-based generator, every rank draws exactly its own columns), so a walker sees the same actions /
    # values whatever the number of GPUs -- as it does for its RSI draws
    buf.actions.copy_(torch.clamp(0.5 * tape_normal(4321, 0, T, rank * n, n, venv.nu, dev), -1, 1))
    buf.values.copy_(tape_normal(4321, 1, T, rank * n, n, 0, dev))
    last_values = tape_normal(4321, 2, 1, rank * n, n, 0, dev)[0].contiguous()
    vn.reset()
    if not args.policy and not args.no_overlap:
        vn.enable_overlap(chunk=max(r for _, r in run_starts))      # pre-generated actions: VecNormalize of step t runs on a side stream under the simulation of step t + 1
        vn.batched_steps = not args.vn_single_steps                 # ... as one dl_vecnormalize_steps call per run
    last_obs = vn.norm_obs_t                                    # observation / episode-start flags that open the next rollout
    last_done = buf.next_starts                                 # row T of the episode-start array: the flags after the last step
    last_done.fill_(1)
    if args.randomize:
        import numpy as np
        gidx = np.arange(rank * n, (rank + 1) * n)
        u = lambda salt: np.array([np.random.default_rng((int(i), salt)).random() for i in gidx])     # seed = global walker index
        venv.set_randomization(0.8 + 0.4 * u(1), 0.5 + 0.6 * u(2))
        ang = 2 * np.pi * u(3)
        # 0.1 s = 20 control steps of push every 2 s = 400 control steps, as a schedule kept on the device (dl_set_push_schedule): the long
        # launches of dl_rollout_fixed need no host round trip per control step
        venv.set_push_schedule(np.stack([50 * np.cos(ang), 50 * np.sin(ang), 0 * ang], 1), (400 * u(4)).astype(np.int32), period=400, duration=20)
    policy = None
    ckpt_meta = None
    if args.checkpoint and not (args.policy and args.walker == 'straight' and args.handles == 1):
        raise SystemExit('--checkpoint needs --policy, the straight walker and one handle')
    if args.policy and args.checkpoint:
        from drloco_amd import checkpoint
        policy, ckpt_meta = checkpoint.load_walking_policy(None if args.checkpoint == 'walking' else args.checkpoint, vec_normalize=vn, seed=99, index_base=rank * n)
        args.hidden = policy.hidden
        vn.norm_obs_t.copy_(venv.obs)          # the opening observation again, under the loaded moments
        vn._normalize_obs_inplace(vn.norm_obs_t)
        # a fixed policy under free-running VecNormalize statistics drifts out of its input distribution (checkpoint.moment_seat): every rollout starts from the checkpoint's moments
        ckpt_restore = checkpoint.moment_seat(vn)
    elif args.policy:
        from drloco_amd.policy import HipPolicy
        policy = HipPolicy(obs_dim=venv.obs_dim, act_dim=venv.nu, hidden=args.hidden, seed=99, index_base=rank * n)

    group = None
    if args.policy and args.handles > 1:
        from drloco_amd.group import HipEnvGroup
        venv.close()
        group = HipEnvGroup(T, num_envs=n, handles=args.handles, device=local_rank, seed=1234, index_base=rank * n)
        venv = group.venvs[0]                      # the handle whose step-kernel launches are bracketed by events

    def rollout_group():
        group.collect_rollouts(policy)
        group.compute_returns_and_advantage(policy)
        group.sync_moments()

    def rollout():
        # RolloutBuffer.add without copies: every producer writes straight into the buffer slot of its result
        # (dl_step -> episode_starts[t+1]; dl_vecnormalize_step -> observations[t+1], rewards[t])
        if policy is None and not args.no_overlap:
            # pre-generated actions: dl_rollout_fixed in runs of --steps-per-launch control steps (one launch of the 16-lane kernel
            # each), their normalisations on the side stream
            buf.reset()
            buf.observations[0].copy_(last_obs)
            buf.episode_starts[0].copy_(last_done)
            for t0, r in run_starts:
                ts = range(t0, t0 + r)
                vn.steps_fixed(buf.actions[t0:ts[-1] + 1], [buf.observations[t + 1] if t + 1 < T else last_obs for t in ts], [buf.rewards[t] for t in ts],
                               buf._starts[t0 + 1:ts[-1] + 2])
            T_loop = 0
        elif policy is not None:
            if ckpt_meta is not None and args.checkpoint_moments == 'seat':
                ckpt_restore()
            buf.collect_rollouts(vn, policy, last_obs, last_done, persistent={'auto': None, 'launches': False, 'persistent': True}[args.rollout_form],
                                 moments=args.moments, deterministic=args.deterministic)     # dl_collect_rollouts: the whole loop in one C-ABI call (one launch in the persistent form)
            T_loop = 0
        else:
            buf.reset()
            buf.observations[0].copy_(last_obs)
            buf.episode_starts[0].copy_(last_done)
            T_loop = T
        for t in range(T_loop):
            nxt = t + 1 < T
            if policy is not None:      # collect_rollouts: actions, values, log_probs = policy.forward(obs) -> straight into the buffer
                policy.forward(buf.observations[t], actions_out=buf.actions[t], values_out=buf.values[t], log_probs_out=buf.log_probs[t])
            vn.step_tensors(buf.actions[t], obs_out=buf.observations[t + 1] if nxt else last_obs, rew_out=buf.rewards[t],
                            done_out=buf.episode_starts[t + 1] if nxt else last_done)
        vn.flush()
        buf.compute_returns_and_advantage(last_values, last_done)
        buf.normalize_advantages()                   # all-reduce of [sum, sum^2, n] when world > 1
        if use_dist:
            vn.sync_moments()                        # one all-reduce of the moment increments per rollout: every rank normalises alike

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if group is not None:
        rollout = rollout_group
    for _ in range(args.warmup):
        rollout()
    if args.solver_stats:
        venv.debug_counters()          # enables (and clears) the per-walker solver diagnostics
    lib.check(venv._lib.dl_profile(venv._h, args.profile_every))    # HIP events around every k-th launch of the step kernel
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rollout()
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, launches = C.c_double(), C.c_int32()
    lib.check(venv._lib.dl_profile_read(venv._h, C.byref(tot_ms), C.byref(launches)))
    lib.check(venv._lib.dl_profile(venv._h, 0))
    # ---- self-check of what the timed region produced (outside the timed region): a broken kernel must not print a number
    checks = {}
    if group is None:
        lib.check(venv._lib.dl_fault_check(venv._h, None))          # raises if a split-workgroup hand-over timed out (DL_E_FAULT)
        starts = buf._starts[1:T + 1]
        n_done = int(starts.sum().item())
        fin = bool(torch.isfinite(buf.observations).all().item() and torch.isfinite(buf.rewards).all().item() and torch.isfinite(buf.advantages).all().item()
                   and torch.isfinite(buf.returns).all().item())
        obs_absmax = float(buf.observations.abs().max().item())
        if vn._ov is not None:           # raw (un-normalised) step outputs of the last rollout: the ring the long launches wrote
            raw_rew, raw_obs = vn._ov['raw_rew'], vn._ov['raw_obs']
        else:
            raw_rew, raw_obs = vn.old_rew, vn.old_obs
        rmin, rmax = float(raw_rew.min().item()), float(raw_rew.max().item())
        fin = fin and bool(torch.isfinite(raw_obs).all().item() and torch.isfinite(raw_rew).all().item())
        # (random torques of +-300 N m make a walker's trunk spin up to 1e3 .. 1e7 rad/s now and then before it falls -- the float64 oracle shows the same
        #  events --, and one such observation stays in VecNormalize's never-forgetting variance: DESIGN.md 7.  Reported, not asserted.)
        # what tells a physics blow-up from a kernel defect: walker-steps on the reference's exception path (MujocoException -> reward 0, done, double reset; the device
        # counts them per walker since dl_create: warm-up included) and raw observations far outside any walking state
        div = torch.zeros(n, dtype=torch.float64, device=dev)
        lib.check(venv._lib.dl_stats_snapshot(venv._h, b'diverged_steps', C.c_void_p(div.data_ptr()), None))
        torch.cuda.synchronize()
        raw_abs = raw_obs.abs()
        checks = {'finite': fin, 'raw_reward_min': rmin, 'raw_reward_max': rmax, 'episodes_ended_last_rollout': n_done, 'normalised_obs_absmax': obs_absmax,
                  'obs_rms_var_max': float(vn.obs_rms.var.max()),
                  'exception_path_steps': int(div.sum().item()), 'walker_steps_run': int(n * T * (args.warmup + args.steps)),
                  'raw_obs_beyond_1e3': int((raw_abs > 1e3).sum().item()), 'raw_obs_beyond_1e10': int((raw_abs > 1e10).sum().item()),
                  'raw_obs_sample': ('the last rollout: %d walker-steps' % (raw_obs.shape[0] * raw_obs.shape[1])) if raw_obs.dim() == 3 else 'the last control step only (a policy in the loop keeps no raw ring)',
                  'note': 'obs_rms_var_max >> 1 is the reference\'s VecNormalize fed with random +-300 N m torques: a trunk spinning up before a fall leaves one huge sample in the never-forgetting '
                          'variance of one or two velocity columns (the float64 oracle shows the same events, DESIGN.md 7), which are then normalised to ~0 for the rest of the run -- '
                          'normalised_obs_absmax within the clip is therefore not a sign of healthy statistics; exception_path_steps / raw_obs_beyond_* count those events'}
        if args.solver_stats:
            cnt = venv.debug_counters(clear=False).astype('float64')          # [4, n]: sum of Newton iterations, max of the last step, sum of constraint rows, diverged steps -- over the timed rollouts
            evals = T * args.steps * 4 * venv.model.frame_skip
            it_last, rows_last = venv.debug_eval_iters(rows=True)               # [evals per step, n]: every evaluation of the last control step
            import numpy as np
            checks['solver'] = {'newton_iterations_per_evaluation': float(cnt[0].mean() / evals), 'p99_walker': float(np.quantile(cnt[0] / evals, 0.99)),
                                'constraint_rows_per_evaluation': float(cnt[2].mean() / evals), 'p99_walker_rows': float(np.quantile(cnt[2] / evals, 0.99)),
                                'last_step_iterations_p50_p99_max': [float(np.quantile(it_last, 0.5)), float(np.quantile(it_last, 0.99)), int(it_last.max())],
                                'last_step_rows_p50_p99_max': [float(np.quantile(rows_last, 0.5)), float(np.quantile(rows_last, 0.99)), int(rows_last.max())]}
        if ckpt_meta is not None:
            ep = lambda name: float(torch.tensor(venv.get_attr(name)).double().mean().item())
            checks['walking'] = {'ep_len_smoothed_mean': ep('ep_len_smoothed'), 'moved_distance_mean_m': ep('moved_distance'), 'mean_step_reward_smoothed': ep('mean_reward_smoothed'),
                                 'checkpoint': args.checkpoint, 'actions': 'mean (deterministic)' if args.deterministic else 'sampled (as in training)',
                                 'vecnormalize': "every rollout starts from the checkpoint's moments and advances them by its own samples (a fixed policy under free-running statistics drifts out of its input distribution)" if args.checkpoint_moments == 'seat' else 'free-running statistics under a fixed policy (--checkpoint-moments free): the workload drifts'}
        # MimicEnv.step: reward = 0 on done, else imitation (<= 1) + 0.2 alive bonus (mimic_env.py:142-168); VecNormalize clips at 10
        assert fin, f'bench self-check: non-finite values in the rollout buffer {checks}'
        assert 0.0 <= rmin and rmax <= 1.2 + 1e-5, f'bench self-check: raw rewards outside [0, 1.2] {checks}'
        assert n_done < n * T and (n_done > 0 or T < 256 or ckpt_meta is not None), f'bench self-check: implausible number of episode ends {checks}'      # (random torques: walkers start falling after ~100 control steps; a trained policy may keep every walker up for a whole rollout)
        assert obs_absmax <= 10.0 + 1e-5, f'bench self-check: normalised observations beyond the clip {checks}'
    if group is not None:          # several handles (HipEnvGroup): the same refusal to print a number over broken output, on every handle's buffer
        fin, n_done, obs_absmax = True, 0, 0.0
        for v_, b_ in zip(group.venvs, group.bufs):
            lib.check(v_._lib.dl_fault_check(v_._h, None))
            fin = fin and all(bool(torch.isfinite(getattr(b_, k_)).all().item()) for k_ in ('observations', 'rewards', 'advantages', 'returns', 'values', 'log_probs'))
            n_done += int(b_._starts[1:T + 1].sum().item())
            obs_absmax = max(obs_absmax, float(b_.observations.abs().max().item()))
        checks = {'finite': fin, 'episodes_ended_last_rollout': n_done, 'normalised_obs_absmax': obs_absmax, 'handles': len(group.bufs),
                  'normalised_reward_absmax': max(float(b_.rewards.abs().max().item()) for b_ in group.bufs)}
        assert fin, f'bench self-check: non-finite values in a handle\'s rollout buffer {checks}'
        assert 0 < n_done < n * T, f'bench self-check: implausible number of episode ends {checks}'
        assert obs_absmax <= 10.0 + 1e-5 and checks['normalised_reward_absmax'] <= 10.0 + 1e-5, f'bench self-check: normalised values beyond the clip {checks}'
    if args.dump and group is None:
        import numpy as np
        ring = {}
        if vn._ov is not None:
            ring = dict(raw_obs=vn._ov['raw_obs'].cpu().numpy(), raw_rew=vn._ov['raw_rew'].cpu().numpy())
        np.savez(f'{args.dump}.rank{rank}.npz', starts=buf._starts.cpu().numpy(), actions=buf.actions.cpu().numpy(), values=buf.values.cpu().numpy(), observations=buf.observations.cpu().numpy(),
                 rewards=buf.rewards.cpu().numpy(), advantages=buf.advantages.cpu().numpy(), obs_mean=vn.obs_rms.mean, obs_var=vn.obs_rms.var, obs_count=vn.obs_rms.count,
                 ret_mean=vn.ret_rms.mean, ret_var=vn.ret_rms.var, ret_count=vn.ret_rms.count, cursor=venv.get_state()['cursor'], qpos=venv.get_state()['qpos'], **ring)
    ranks_seen = 1
    per_rank = None
    if use_dist:
        # one all-gather: every rank's wall time of the timed region and the summed duration of its bracketed step-kernel launches -- a SCALE line can then tell
        # imbalance between ranks (kernel time min / max) from the cost of the collectives (wall time - kernel time)
        mine = torch.tensor([dt * 1e3, tot_ms.value, float(launches.value)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ar = torch.stack(allr).cpu().numpy()
        per_rank = {'wall_ms': [round(float(x), 3) for x in ar[:, 0]], 'step_kernel_ms': [round(float(x), 3) for x in ar[:, 1]], 'bracketed_launches': [int(x) for x in ar[:, 2]],
                    'step_kernel_ms_min_max': [round(float(ar[:, 1].min()), 3), round(float(ar[:, 1].max()), 3)],
                    'outside_step_kernel_ms_max': round(float((ar[:, 0] - ar[:, 1]).max()), 3)}
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        seen = torch.ones(1, dtype=torch.int64, device=dev)          # every rank adds 1: what the collective library saw, not what the environment claims
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ranks_seen = int(seen.item())

    if rank == 0:
        env_steps = n * T * args.steps * world
        value = env_steps / dt
        avg_launch_s = tot_ms.value / max(1, launches.value) / 1e3
        steps_per_launch = venv._lib.dl_profile_steps(venv._h) / max(1, launches.value)      # 8 with dl_rollout_fixed, 1 with dl_step
        n_prof = venv.num_envs                       # walkers of the handle whose launches were bracketed (all of the rank's unless --handles)
        algo_bytes = ALGO_BYTES_PER_ENV_STEP_LOCO3D if args.walker == 'loco3d' else ALGO_BYTES_PER_ENV_STEP
        achieved = algo_bytes * n_prof * steps_per_launch / avg_launch_s / 1e9
        # counter-derived figures come from a committed rocprofv3 PMC pass (profiles/traffic_env_step.json, tools/gpu_round_profile.sh): they are
        # reported only for the configuration that pass measured AND only while the kernel sources are the ones it measured
        traffic = valu_busy = mfma_busy = prof_origin = valu_mix = None
        base_cfg = args.walker == 'straight' and args.lanes in (0, 16) and not args.randomize and not args.no_overlap and n == 4096 and T == 512 and not args.runs and not args.no_split and not args.vn_single_steps
        persistent_line = args.policy and group is None and getattr(buf, 'last_form', '') == 'persistent' and args.handles == 1 and args.hidden == 512
        tname = None          # which committed pass belongs to this command line (profiles/, written by tools/summarize_profile.py)
        if base_cfg and not args.policy:
            tname = 'traffic_env_step.json'
        elif base_cfg and persistent_line and args.checkpoint == 'walking' and not args.deterministic:
            tname = 'traffic_env_step_policy_walking_per_rollout.json' if args.moments == 'per_rollout' else 'traffic_env_step_policy_walking.json'
        elif base_cfg and persistent_line and not args.checkpoint and not args.deterministic:
            tname = 'traffic_env_step_policy_per_rollout.json' if args.moments == 'per_rollout' else 'traffic_env_step_policy.json'
        elif args.walker == 'loco3d' and args.lanes in (0, 16) and not args.policy and not args.randomize and not args.no_overlap and n == 4096 and T == 512 and not args.runs:
            tname = 'traffic_env_step_loco3d.json'
        tfile = os.path.join(ROOT, 'profiles', tname) if tname else None
        if tfile and os.path.exists(tfile):
            try:
                pj = json.load(open(tfile))
                cfg_now = (C.c_int32 * 3)()
                lib.check(venv._lib.dl_profile_launch_config(venv._h, cfg_now))
                launch_now = {'grid': int(cfg_now[0]), 'workgroup': int(cfg_now[1]), 'lds_bytes': int(cfg_now[2])}
                ml = pj.get('launch')          # (passes stamped before round 6 carry no geometry; rocprofv3's counter csv reports LDS_Block_Size 0 for dynamic LDS: compared only when it is there)
                same_launch = ml is None or (ml.get('grid') == launch_now['grid'] and ml.get('workgroup') == launch_now['workgroup'] and ml.get('lds_bytes') in (0, None, launch_now['lds_bytes']))
                if pj.get('kernel_code_sha16') == kernel_code_sha16() and same_launch:
                    two_waves = split or persistent_line          # two waves per SIMD of which one mostly sleeps: busy cycles against the SIMDs' time, not the waves'
                    traffic, valu_busy = pj.get('hbm_bytes_per_launch'), (pj.get('valu_busy_frac_simd') if two_waves else pj.get('valu_busy_frac'))
                    mfma_busy = pj.get('mfma_busy_frac_simd')
                    valu_mix = pj.get('valu_mix')
                    prof_origin = {'file': 'profiles/' + tname, 'tag': pj.get('tag'), 'kernel_code_sha16': pj.get('kernel_code_sha16')}
                else:
                    prof_origin = {'file': 'profiles/' + tname, 'stale': True, 'measured_sha16': pj.get('kernel_code_sha16') or pj.get('kernel_sources_sha16'), 'built_sha16': kernel_code_sha16(),
                                   'measured_launch': pj.get('launch'), 'launch': launch_now}
            except Exception:
                traffic = None
        out = {
            'metric': 'env-steps/s (whole node) 3D straight-walk walker', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('loco3d 19-dof walker (synthetic mocap table), ' if args.walker == 'loco3d' else 'straight_walking 3D walker, ') + f'{n} parallel envs per GPU, fixed {T}-step synthetic rollout '
                                   '(env step + VecNormalize + rollout store + GAE + adv-norm)',
                       'envs_per_gpu': n, 'rollout_len': T, 'frame_skip': 10 if args.walker == 'loco3d' else 5, 'integrator': 'RK4', 'sharding': f'env-index ranges x{world}', 'actions': (f'device policy (dl_policy_forward, {venv.obs_dim}-{args.hidden}-{args.hidden}-{{{venv.nu},1}})' + (f', TRAINED weights ({args.checkpoint}): the walkers walk' if args.checkpoint else '') + (', mean actions' if args.deterministic else '') + (f', {args.handles} handles on {args.handles} streams' if group is not None else '')) if args.policy else 'pre-generated',
                       'vecnormalize': 'main stream' if (args.policy or args.no_overlap) else ('side stream, one dl_vecnormalize_steps call per run' if not args.vn_single_steps else 'side stream, under the following run of env steps'),
                       'step_kernel_form': 'split workgroups: 4 dynamics + 4 constraint waves per 16 walkers (dl_set_split 1)' if split else 'one wave per 4 walkers',
                       'env_launches': (('ONE persistent launch per rollout (k_rollout_pairs: every wave pair takes its four walkers through policy + env step, moments per rollout (relaxation))' if args.moments == 'per_rollout' else 'ONE persistent launch per rollout (k_rollout_persistent: policy + env step + moment exchange per control step), exact per-step moments')
                                        if (args.policy and group is None and getattr(buf, 'last_form', '') == 'persistent') else 'one per control step') if (args.policy or args.no_overlap) else
                                       'dl_rollout_fixed, runs of ' + ' + '.join(str(r) for _, r in run_starts) + ' control steps per launch',
                       'dynamics': 'per-walker mass/friction randomisation + 50 N pushes on a device-resident schedule (config 5 stress test)' if args.randomize else 'nominal'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'kernel': ('k_env_step<float,TopoWalker165,32>' if args.walker == 'loco3d' else 'k_env_step<float,TopoStraight,64>') if args.lanes == 1 else
                                   ((('k_rollout_pairs<%s>' if args.moments == 'per_rollout' else 'k_rollout_persistent<%s>') % ('TopoWalker165' if args.walker == 'loco3d' else 'TopoStraight')) if (args.policy and group is None and getattr(buf, 'last_form', '') == 'persistent') else (('k_env_step_g16_split<float,TopoWalker165>' if split else 'k_env_step_g16<float,TopoWalker165>') if args.walker == 'loco3d' else ('k_env_step_g16_split<float,TopoStraight>' if split else 'k_env_step_g16<float,TopoStraight>'))), 'avg_launch_us': avg_launch_s * 1e6,
                         'launches': launches.value, 'sampled_every': args.profile_every, 'control_steps_per_launch': steps_per_launch, 'algorithmic_bytes_per_launch': algo_bytes * n_prof * steps_per_launch,
                         'valu_busy_frac': valu_busy, 'mfma_busy_frac': mfma_busy, 'valu_mix': valu_mix, 'from_profile': prof_origin,
                         'note': 'the fused dynamics kernel is FP32-VALU issue / latency bound (SURVEY.md 8d): valu_busy_frac = SQ_ACTIVE_INST_VALU / ' + ('the SIMD cycles of the launch (1024 SIMDs x GRBM_GUI_ACTIVE / 32; two waves per SIMD)' if split else 'SQ_WAVE_CYCLES') + ' of the committed rocprofv3 PMC pass (profiles/) is the fraction of ISSUE slots used -- not of arithmetic: valu_mix (the same pass: SQ_INSTS_VALU_ADD/MUL/FMA/TRANS_F32, SQ_THREAD_CYCLES_VALU) says how many of the issued instructions are FP32 arithmetic and what that is of the 157.3 TFLOP/s FP32 vector peak (fp32_frac_of_157_3_tf_vector_peak); the HBM fraction is reported as the contract asks'},
        }
        out['code_object'] = {k: v for k, v in (lib.SELECTED or {}).items() if k != 'probe'}          # which of the two builds ran (drloco_amd/lib.py: two wait states in front of a DPP read unless this device proved that one is enough)
        out['distributed'] = {'world_size': dist.get_world_size() if use_dist else 1, 'ranks_seen': ranks_seen, 'backend': dist.get_backend() if use_dist else None, 'vn_sync': vn.sync,
                              'collectives_per_control_step': 'all-reduce of 2 x (obs_dim + 1) doubles (VecNormalize batch sums: exact per-step moments over all ranks)' if (use_dist and vn.sync == 'per_step') else None,
                              'collectives_per_rollout': 'all-reduce of 3 doubles (adv-norm sums) + all-reduce of 2 x (obs_dim + 1) + 2 doubles (VecNormalize moment increments)' if use_dist else None,
                              'per_rank': per_rank}
        out['self_check'] = checks
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_base
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
