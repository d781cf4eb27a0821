"""ctypes binding of drloco_amd/csrc/libdrloco_hip.so (the C-ABI in include/drloco_hip.h).

There is no CPU fallback: importing works anywhere (the library links only against
libamdhip64), but creating an environment without a HIP device raises."""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('DL_LIB_PATH') or os.path.join(CSRC, 'libdrloco_hip.so')      # DL_LIB_PATH: experiment builds (build_variants/), never the product
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include')
_SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.hpp', '.h')))

_lib = None
# compiler flags of the product build beyond -O3 (experiments: DL_EXTRA_FLAGS in the environment).  Both are measured choices for the
# step kernels, which run one wave per SIMD and are bound by the latency of their dependent instructions: the SLP vectoriser
# packs scalar float arithmetic into v_pk_* instructions at the price of register shuffles that cost more issue slots than they save
# (-fno-slp-vectorize: +1.6 % straight walker, +4.7 % 19-dof walker), and the ILP-driven machine scheduler orders the long straight-line
# blocks better than the default occupancy-driven one, which has no occupancy to gain here (another +1.2 % / +3.3 %).
EXTRA_FLAGS = ['-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-ilp']


class DrlocoError(RuntimeError):
    pass


class DrlocoFault(DrlocoError):
    """DL_E_FAULT: a kernel of the handle reported a fault (include/drloco_hip.h: dl_fault_check)."""


LISTING_DIR = os.path.join(os.path.dirname(_HERE), 'build_dbg', 'listing')
LISTING = os.path.join(LISTING_DIR, 'dl_kernels-hip-amdgcn-amd-amdhsa-gfx950.s')


def _sources():
    return [os.path.join(CSRC, s) for s in _SOURCES] + [os.path.join(INCLUDE, 'drloco_hip.h')]


def build(force=False, verbose=False, listing=False):
    """hipcc --offload-arch=gfx950 of the kernels + C-ABI into an in-tree shared library.  listing=True keeps the device assembly of the
    same compilation (-save-temps, build_dbg/listing/) for tools/check_dpp_hazards.py -- the hand-written DPP statements carry their own
    wait states, which the compiler's hazard recogniser does not check."""
    srcs = _sources()
    fresh = lambda path: os.path.exists(path) and all(os.path.getmtime(s) <= os.path.getmtime(path) for s in srcs)
    if not force and fresh(LIB_PATH) and (not listing or fresh(LISTING)):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I' + INCLUDE, '-I' + CSRC] + EXTRA_FLAGS + \
          os.environ.get('DL_EXTRA_FLAGS', '').split() + (['-save-temps'] if listing else []) + [os.path.join(CSRC, 'dl_kernels.hip'), '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd))
    if listing:
        os.makedirs(LISTING_DIR, exist_ok=True)
    subprocess.check_call(cmd, cwd=LISTING_DIR if listing else None)
    return LIB_PATH


def device_code_sha16(path=None):
    """Identity of the DEVICE code of a built library: sha256 over its `.hip_fatbin` section (the gfx950 code object hipcc embedded), first 16
    hex digits.  Comments, host code and anything else that leaves the compiled kernels alone do not move it; profiles/*.json carry it so that
    bench.py replays counter-derived figures only for the kernels they were measured on."""
    import hashlib
    import struct
    with open(path or LIB_PATH, 'rb') as f:
        data = f.read()
    if data[:4] != b'\x7fELF' or data[4] != 2:
        raise DrlocoError(f'{path or LIB_PATH}: not a 64-bit ELF file')
    shoff, = struct.unpack_from('<Q', data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', data, 0x3A)
    sec = lambda i: struct.unpack_from('<IIQQQQIIQQ', data, shoff + i * shentsize)          # name, type, flags, addr, offset, size, ...
    str_off = sec(shstrndx)[4]
    for i in range(shnum):
        name_off, _, _, _, off, size = sec(i)[:6]
        name = data[str_off + name_off:data.index(b'\0', str_off + name_off)]
        if name == b'.hip_fatbin':
            return hashlib.sha256(data[off:off + size]).hexdigest()[:16]
    raise DrlocoError(f'{path or LIB_PATH}: no .hip_fatbin section (not a hipcc build?)')


def check_dpp_hazards():
    """Run tools/check_dpp_hazards.py over the listing of the product build (building it if needed); raises on a violation."""
    build(listing=True)
    tool = os.path.join(os.path.dirname(_HERE), 'tools', 'check_dpp_hazards.py')
    p = subprocess.run([os.environ.get('PYTHON', 'python3'), tool, LISTING], capture_output=True, text=True)
    if p.returncode != 0:
        raise DrlocoError('DPP read-after-write hazard in the device code:\n' + p.stdout[-4000:] + p.stderr[-2000:])
    return p.stdout.strip().splitlines()[-1]


def check_mfma_overlap():
    """Run tools/check_mfma_overlap.py over the listing of the product build (building it if needed); raises on a violation: an MFMA whose
    destination differs from its source C and lies over an A / B operand, a relocated 4x4x1 accumulator, or a missing hand-written wait state."""
    build(listing=True)
    tool = os.path.join(os.path.dirname(_HERE), 'tools', 'check_mfma_overlap.py')
    p = subprocess.run([os.environ.get('PYTHON', 'python3'), tool, LISTING], capture_output=True, text=True)
    if p.returncode != 0:
        raise DrlocoError('MFMA operand overlap / wait-state violation in the device code:\n' + p.stdout[-4000:] + p.stderr[-2000:])
    return p.stdout.strip().splitlines()[-1]


_V, _I, _P = C.c_void_p, C.c_int32, C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)       -- every symbol include/drloco_hip.h declares
    'dl_last_error': (C.c_char_p, []),
    'dl_abi_version': (C.c_int, []),
    'dl_create': (C.c_int, [C.POINTER(abi.ModelDesc), C.POINTER(abi.RefsDesc), C.POINTER(abi.Config), _I, _I, C.POINTER(_V)]),
    'dl_destroy': (C.c_int, [_V]),
    'dl_num_envs': (_I, [_V]),
    'dl_obs_dim': (_I, [_V]),
    'dl_act_dim': (_I, [_V]),
    'dl_real_size': (_I, [_V]),
    'dl_reset': (C.c_int, [_V, _P, _P, _P, _P, _P]),
    'dl_set_eval': (C.c_int, [_V, _I]),
    'dl_step': (C.c_int, [_V, _P, _P, _P, _P, _P, _P, _P]),
    'dl_rollout_fixed': (C.c_int, [_V, _I, _P, _P, _P, _P, _P]),
    'dl_get_state': (C.c_int, [_V, _P, _P, _P, _P, _P, _P]),
    'dl_set_state': (C.c_int, [_V, _P, _P, _P, _P, _P, _P]),
    'dl_forward': (C.c_int, [_V, _P, _P, _P, _P, _P, _P]),
    'dl_set_randomization': (C.c_int, [_V, _P, _P, _P]),
    'dl_set_push': (C.c_int, [_V, _P, _P]),
    'dl_set_push_schedule': (C.c_int, [_V, _P, _P, _I, _I, _P]),
    'dl_set_split': (C.c_int, [_V, _I]),
    'dl_fault_check': (C.c_int, [_V, C.POINTER(_I)]),
    'dl_fault_clear': (C.c_int, [_V]),
    'dl_terminate_early': (C.c_int, [_V, _P, _P]),
    'dl_stats_snapshot': (C.c_int, [_V, C.c_char_p, _P, _P]),
    'dl_profile': (C.c_int, [_V, _I]),
    'dl_profile_read': (C.c_int, [_V, C.POINTER(C.c_double), C.POINTER(_I)]),
    'dl_profile_steps': (C.c_int, [_V]),
    'dl_moments_update': (C.c_int, [_P, _P, _P, _P, _I, _I, _P]),
    'dl_normalize_obs': (C.c_int, [_P, _P, _P, _I, _I, C.c_double, C.c_double, _P]),
    'dl_normalize_reward': (C.c_int, [_P, _P, _P, _P, _P, _P, _I, C.c_double, C.c_double, C.c_double, _P]),
    'dl_vecnormalize_step': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, C.c_double, C.c_double, C.c_double, C.c_double, _I, _P, _P, _P, _P]),
    'dl_vn_local_sums': (C.c_int, [_P, _P, _P, _P, _P, _I, _I, C.c_double, _I, _P, _P]),
    'dl_vn_merge_sums': (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'dl_vecnormalize_steps': (C.c_int, [C.POINTER(abi.VecNormState), _I, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    'dl_policy_forward': (C.c_int, [C.POINTER(abi.PolicyParams), _P, _I, _P, C.c_uint64, C.c_uint64, _I, _I, _P, _P, _P, _P]),
    'dl_policy_pack': (C.c_int, [C.POINTER(abi.PolicyParams), _P, _P]),
    'dl_policy_forward_packed': (C.c_int, [C.POINTER(abi.PolicyParams), _P, _P, _I, _P, C.c_uint64, C.c_uint64, _I, _I, _P, _P, _P, _P]),
    'dl_policy_forward_pair': (C.c_int, [C.POINTER(abi.PolicyParams), _P, _P, _I, _P, C.c_uint64, C.c_uint64, _I, _I, _P, _P, _P, _P]),
    'dl_rollout_policy': (C.c_int, [_V, C.POINTER(abi.PolicyParams), C.c_uint64, C.c_uint64, _I, C.POINTER(abi.VecNormState), _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'dl_rollout_persistent_ok': (C.c_int, [_V, C.POINTER(abi.PolicyParams)]),
    'dl_collect_rollouts': (C.c_int, [_V, C.POINTER(abi.PolicyParams), C.c_uint64, C.c_uint64, _I, C.POINTER(abi.VecNormState), _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'dl_gae': (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_float, _I, _I, _P, _P, _P]),
    'dl_adv_stats': (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    'dl_adv_normalize': (C.c_int, [_P, C.c_int64, _P, _P]),
}
_EXTRA = {
    'dl_abi_sizeof': (C.c_int, [C.c_int]),
    'dl_debug_inject': (C.c_int, [_V, _P, _P, _P, _P, _P]),
    'dl_debug_counters': (C.c_int, [_V, _P, _I, _P]),
    'dl_debug_set_spin_limit': (C.c_int, [_V, _I, _I]),
    'dl_debug_rollout_prof': (C.c_int, [_V, _P, _P]),
    'dl_debug_set_grid_spin': (C.c_int, [_V, _I]),
    'dl_debug_capstate': (C.c_int, [_V, _P, _P]),
    'dl_debug_eval_iters': (C.c_int, [_V, _P, _P]),
    'dl_debug_last_ctrl': (C.c_int, [_V, _P, _P]),
    'dl_debug_selftest': (C.c_int, [_P, _P, _P]),
    'dl_debug_forward_timed': (C.c_int, [_V, _P, _P, _P, _P]),
    'dl_debug_step_timed': (C.c_int, [_V, _P, _P, _P, _P, _P, _P]),
}


def load():
    """Load the HIP library; raises DrlocoError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DrlocoError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in {**_SIGNATURES, **_EXTRA}.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.dl_abi_version() != abi.DL_ABI_VERSION:
        raise DrlocoError('ABI version mismatch between drloco_amd/abi.py and libdrloco_hip.so')
    for which, struct in enumerate((abi.ModelDesc, abi.RefsDesc, abi.Config)):
        if lib.dl_abi_sizeof(which) != C.sizeof(struct):
            raise DrlocoError(f'struct size mismatch for {struct.__name__}')
    _lib = lib
    return lib


def check(rc):
    if rc == abi.DL_E_FAULT:
        raise DrlocoFault(f'drloco_hip fault: {load().dl_last_error().decode()}')
    if rc != 0:
        raise DrlocoError(f'drloco_hip error {rc}: {load().dl_last_error().decode()}')
