import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def model():
    from drloco_amd.models import make_model
    return make_model()


@pytest.fixture(scope='session')
def refs():
    from drloco_amd.mocap import RefTable
    return RefTable.load()


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope='session')
def ramp_refs(tmp_path_factory):
    """The reference's DEFAULT file layout (40 rows per step, GRF rows 35-36, trunk Euler rows 37-39: straight_walk_trajecs.py:22-27,85-91) on the
    synthetic 250-step file golden G14 was generated from (same generator, same seed), through the product's converter: (RefTable, golden dict)."""
    import numpy as np
    from drloco_amd import mocap
    with np.load(os.path.join(GOLDEN, 'G14_ramp_layout.npz')) as z:
        g = {k: z[k] for k in z.files}
    path = str(tmp_path_factory.mktemp('ramp') / 'Trajecs_Ramp_Slow_400Hz_EulerTrunkAdded.mat')
    mocap.write_straight_walk_mat(path, mocap.synthetic_straight_walk(n_steps=int(g['n_steps']), seed=int(g['seed']), n_rows=40), nested=True)
    return mocap.convert_straight_walk_mat(path), g
