"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/drloco_hip.h declares, agrees with the ctypes mirror on struct sizes, and refuses
to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from drloco_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    return lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'drloco_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dl_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f'{n} declared in drloco_hip.h but not exported'
        assert n in lib._SIGNATURES, f'{n} has no ctypes signature in drloco_amd/lib.py'
    assert sorted(lib._SIGNATURES) == names


def test_struct_sizes_and_version(L):
    assert L.dl_abi_version() == abi.DL_ABI_VERSION
    for which, st in enumerate((abi.ModelDesc, abi.RefsDesc, abi.Config, abi.PolicyParams, abi.VecNormState)):
        assert L.dl_abi_sizeof(which) == C.sizeof(st)


def test_no_cpu_fallback(L, model, refs):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a device is present')
    h = C.c_void_p()
    desc = refs.as_desc()
    cfg = abi.default_config()
    rc = L.dl_create(C.byref(model), C.byref(desc), C.byref(cfg), 4, 0, C.byref(h))
    assert rc == abi.DL_E_NODEVICE and not h
    assert b'no HIP device' in L.dl_last_error()
    from drloco_amd.vec_env import HipVecEnv
    with pytest.raises(lib.DrlocoError):
        HipVecEnv(num_envs=4)


@pytest.mark.parametrize('var', ['CUDA_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES'])
def test_nodevice_text_names_the_hidden_gpus(model, refs, var):
    """The reference's train.py hides the GPUs from its own process (use_cpu(): CUDA_VISIBLE_DEVICES = "", drloco/train.py:35,80, whenever config.py:10 USE_CPU is True --
    the default) before it builds the environments.  Where that hides the devices from HIP, dl_create's DL_E_NODEVICE says which variable does it and what to change.
    MEASURED on the MI355X boxes (ROCm 7.2, tools/diag_visible_devices.py): an EMPTY CUDA_VISIBLE_DEVICES does NOT hide the GPU from HIP (torch.cuda.device_count() stays 1,
    dl_create succeeds) -- unlike CUDA --, an empty HIP_VISIBLE_DEVICES does.  So on a GPU box the CUDA_ variable may leave the device visible (accepted); whenever the device
    is gone, the text must name the variable.  (A child process: the variable has to be in place before the HIP runtime initialises.)"""
    import subprocess
    import sys
    code = (
        "import os, ctypes as C\n"
        f"os.environ['{var}'] = ''          # drloco/train.py:35 sets CUDA_VISIBLE_DEVICES\n"
        "from drloco_amd import abi, lib, mocap, models\n"
        "L = lib.load(); h = C.c_void_p(); m = models.make_model(); r = mocap.RefTable.load(); d = r.as_desc(); c = abi.default_config()\n"
        "rc = L.dl_create(C.byref(m), C.byref(d), C.byref(c), 4, 0, C.byref(h))\n"
        "print(rc, '|', L.dl_last_error().decode())\n")
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert p.returncode == 0, p.stderr
    rc, _, text = p.stdout.strip().partition('|')
    rc, text = rc.strip(), text.strip()
    if int(rc) == 0:
        assert var == 'CUDA_VISIBLE_DEVICES', f'{var} = "" left a device visible: {p.stdout}'          # (measured on ROCm 7.2: only the CUDA_ spelling is ignored when empty)
        return
    assert int(rc) == abi.DL_E_NODEVICE, p.stdout
    assert f'{var} is set to the empty string' in text and 'USE_CPU' in text and 'train.py:35' in text, text


def test_null_handle_is_rejected(L):
    assert L.dl_step(None, None, None, None, None, None, None, None) == abi.DL_E_INVAL
    assert L.dl_num_envs(None) == 0


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'drloco_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('the oracle', '').replace("oracle's", '').replace('CPU oracle', '').lower() or f in ('dl_core.hpp',), (dirpath, f)


def test_the_boundary_is_usable_from_plain_c(L, tmp_path):
    """examples/c_abi_smoke.c: the header compiles as C99, the library loads with dlopen, versions and struct sizes agree, and dl_create refuses
    to run without a device (DL_E_NODEVICE) or with an empty model descriptor (DL_E_INVAL)."""
    import shutil
    import subprocess
    if not shutil.which('gcc'):
        pytest.skip('no gcc')
    exe = str(tmp_path / 'c_abi_smoke')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_abi_smoke.c'), '-o', exe, '-ldl'])
    p = subprocess.run([exe, lib.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.startswith('ok: ABI %d' % abi.DL_ABI_VERSION), p.stdout + p.stderr
