"""The hand-over protocol of the split workgroups (a dynamics wave and a partner wave that works one forward evaluation AHEAD, exchanging requests, rows and
look-ahead geometry through mailbox words in LDS) on the CPU: the SAME kernel source (g_wave_env_step<SPLIT>, g_constraint_server) with each wave emulated as 64 fibers and
the two waves' rounds interleaved by a scheduler (tests/host_emu/dl_group_emu.hpp run_pair: seeded random, random bursts, "dynamics wave first", "partner first"; every
cross-lane operation is a possible switch point, every flag post -- DL_WAKE -- and every poll -- DL_SLEEP -- an explicit one).
  * the product protocol is schedule-independent: every schedule gives the bits of every other one and of the one-wave form (the split form moves work between waves, the
    arithmetic per lane is the same), multi-step launches with resets inside (configurations that were not announced: the command-2 path), both walkers, no fault word;
  * the round-4 defect -- the partner reading the announced configuration AFTER posting the rows, when the word may already carry the dynamics wave's next request: about one
    launch in a hundred differed from itself on the GPU and was found by luck -- re-introduced behind -DDL_EXP_R4_LATE_READ in a separate build, is caught DETERMINISTICALLY:
    the "dynamics wave first" schedule produces other results than the "partner first" schedule.
The ownership rule this guards: a mailbox word belongs to the dynamics wave again the moment the rows' flag is posted."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_emu'))

SCHEDULES = [(0, 1), (0, 2), (0, 3), (1, 0), (2, 0), (3, 5), (3, 6)]          # (policy, seed)


@pytest.fixture(scope='module')
def emu():
    import emu as E
    E.lib()
    return E


def _rollout(E, model, refs, cfg, n, acts, precision, schedule):
    e = E.EmuEnv(model, refs, cfg, n, precision)
    e.reset()
    if schedule is None:
        o, r, d, _, _ = e.gstep(acts)
        fw = 0
    else:
        o, r, d, fw = e.gstep_split(acts, policy=schedule[0], seed=schedule[1])
    st = e.get_state()
    e.close()
    return o, r, d, st['qpos'], st['qvel'], st['cursor'], fw


@pytest.mark.timeout(900)
def test_split_protocol_is_schedule_independent(emu, model, refs):
    from drloco_amd import abi
    cfg = abi.default_config(seed=5, ep_dur_max=4)
    #                       # every walker times out and resets inside the launch: the first evaluation after a reset was never announced
    n, T = 8, 7
    rng = np.random.default_rng(0)
    acts = np.clip(0.5 * rng.standard_normal((T, n, 8)), -1, 1).astype(np.float32)
    ref = _rollout(emu, model, refs, cfg, n, acts, 32, None)
    assert ref[2].sum() >= n                 # episodes ended inside the launch
    for sch in SCHEDULES:
        got = _rollout(emu, model, refs, cfg, n, acts, 32, sch)
        assert got[6] == 0, (sch, got[6])    # no wave gave up on its partner
        for a, b, name in zip(ref[:6], got[:6], ('obs', 'rew', 'done', 'qpos', 'qvel', 'cursor')):
            assert np.array_equal(a, b), (sch, name, float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()))


@pytest.mark.timeout(900)
def test_split_protocol_19dof_walker(emu):
    """The same for the walker with replicated root translations (round 5: three more words per request, the lanes' M[j][t] block, the lane file)."""
    from drloco_amd import abi, mocap, models
    ang, vel = mocap.synthetic_loco3d(L=4000, seed=1)
    table = mocap.loco3d_table(ang, vel)
    m = models.make_model(models.WALKER_165CM)
    cfg = abi.loco3d_config(seed=3)
    n, T = 4, 2
    rng = np.random.default_rng(1)
    acts = np.clip(0.5 * rng.standard_normal((T, n, 13)), -1, 1).astype(np.float32)
    ref = _rollout(emu, m, table, cfg, n, acts, 32, None)
    for sch in ((1, 0), (2, 0), (0, 4)):
        got = _rollout(emu, m, table, cfg, n, acts, 32, sch)
        assert got[6] == 0
        for a, b, name in zip(ref[:6], got[:6], ('obs', 'rew', 'done', 'qpos', 'qvel', 'cursor')):
            assert np.array_equal(a, b), (sch, name)


_CHILD = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import emu as E
from drloco_amd import abi, mocap, models
model, refs = models.make_model(), mocap.RefTable.load()
cfg = abi.default_config(seed=5)
n, T = 8, 6
acts = np.clip(0.5 * np.random.default_rng(0).standard_normal((T, n, 8)), -1, 1).astype(np.float32)
out = []
for policy in (1, 2):
    e = E.EmuEnv(model, refs, cfg, n, 32); e.reset()
    o, r, d, fw = e.gstep_split(acts, policy=policy, seed=0)
    out.append(o)
print('DIFFERS' if not np.array_equal(out[0], out[1]) else 'SAME', float(np.abs(out[0] - out[1]).max()))
'''


@pytest.mark.timeout(900)
def test_the_round4_defect_is_caught_deterministically():
    env = dict(os.environ, DL_EMU_R4BUG='1')
    p = subprocess.run([sys.executable, '-c', _CHILD % (ROOT, os.path.join(ROOT, 'tests', 'host_emu'))], env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert p.stdout.strip().startswith('DIFFERS'), p.stdout          # the late read of the announced configuration makes the result depend on the schedule
    env.pop('DL_EMU_R4BUG')
    p = subprocess.run([sys.executable, '-c', _CHILD % (ROOT, os.path.join(ROOT, 'tests', 'host_emu'))], env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0 and p.stdout.strip().startswith('SAME'), p.stdout + p.stderr[-1000:]          # the product build, same schedules: identical
