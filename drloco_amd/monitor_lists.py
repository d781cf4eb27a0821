"""Monitor's per-episode lists on the host (drloco/mujoco/monitor_wrapper.py:57-63,91-93,105-107,122-123,131-132):
`rsi_positions`, `et_positions`, `difficult_rsi_phases` and `median_abs_torque_smoothed`.

The device keeps everything the training loop reads (the smoothed statistics, `dl_stats_snapshot`); these four are lists that
grow by one entry per episode (and a median over an episode's steps) -- nothing on the path reads them (the callback's readers
are commented out, drloco/common/callback.py:71-72,213-215,232), so they are kept on the host, and only on request
(`HipVecEnv.track_monitor_lists()`), from four small device words per walker the step kernel maintains: the reference-cursor
position at the first step and at the end of an episode, the step's mean absolute torque and the "shorter than 0.75 x the smoothed
episode length" flag.  `update()` is called once per control step by the numpy VecEnv surface."""
import numpy as np


class MonitorLists:
    def __init__(self, n_envs, ep_dur_max=3000):
        self.n = int(n_envs)
        self.rsi_positions = [[] for _ in range(self.n)]
        self.et_positions = [[] for _ in range(self.n)]
        self.difficult_rsi_phases = [[] for _ in range(self.n)]
        self.median_abs_torque_smoothed = np.zeros(self.n)
        self._has_median = np.zeros(self.n, bool)
        self._tor = np.zeros((int(ep_dur_max) + 2, self.n), np.float64)        # the running episode's per-step torques
        self._len = np.zeros(self.n, np.int64)

    def update(self, done, init_pos, et_pos, last_abs_torque, difficult):
        """One control step of all walkers: done bool[N]; the four device words after the step (float64[N] each; init_pos / et_pos /
        difficult are only read where they are current)."""
        done = np.asarray(done, bool)
        cap = self._tor.shape[0]
        idx = np.minimum(self._len, cap - 1)
        self._tor[idx, np.arange(self.n)] = last_abs_torque
        first = self._len == 0
        self._len += 1
        for i in np.nonzero(first)[0]:
            self.rsi_positions[i].append(int(init_pos[i]))
        for i in np.nonzero(done)[0]:
            self.et_positions[i].append(int(et_pos[i]))
            if difficult[i] != 0:
                self.difficult_rsi_phases[i].append(int(init_pos[i]))
            med = float(np.median(self._tor[:min(self._len[i], cap), i]))
            if self._has_median[i]:
                self.median_abs_torque_smoothed[i] = 0.75 * med + 0.25 * self.median_abs_torque_smoothed[i]
            else:
                self.median_abs_torque_smoothed[i] = med
                self._has_median[i] = True
            self._len[i] = 0

    def get(self, name, i):
        v = getattr(self, name)
        return float(v[i]) if name == 'median_abs_torque_smoothed' else list(v[i])


NAMES = ('rsi_positions', 'et_positions', 'difficult_rsi_phases', 'median_abs_torque_smoothed')
