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
# The two code objects of the product (dl_group.hpp, DL_DPP_WAIT): 'w2' = libdrloco_hip.so, the DEFAULT -- the ISA manual's two wait states between a VALU write and its
# DPP read in the hand-written statements; 'w1' = libdrloco_hip_dpp1.so, one state: what gfx950 was measured to need (+2.5 % headline), used only on a device that has
# proven it in this process (load() below).  DL_LIB_PATH (experiment builds) bypasses the choice.
VARIANTS = {
    'w2': dict(path=os.path.join(CSRC, 'libdrloco_hip.so'), flags=[], need=2, listing=LISTING),
    'w1': dict(path=os.path.join(CSRC, 'libdrloco_hip_dpp1.so'), flags=['-DDL_DPP_WAIT=1'], need=1,
               listing=os.path.join(os.path.dirname(_HERE), 'build_dbg', 'listing_dpp1', 'dl_kernels-hip-amdgcn-amd-amdhsa-gfx950.s')),
}


def _sources():
    return [os.path.join(CSRC, s) for s in _SOURCES] + [os.path.join(INCLUDE, 'drloco_hip.h')]


def _build_cmd(variant, listing):
    v = VARIANTS[variant]
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # -ffile-prefix-map: no absolute path of the build directory inside the code object (line tables, __FILE__), so that the same sources hash to the same
    # device code wherever they are built (device_code_sha16 is what profiles/*.json are stamped with)
    return [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffile-prefix-map=' + os.path.dirname(_HERE) + '=.', '-I' + INCLUDE, '-I' + CSRC] + EXTRA_FLAGS + \
        v['flags'] + os.environ.get('DL_EXTRA_FLAGS', '').split() + (['-save-temps'] if listing else []) + [os.path.join(CSRC, 'dl_kernels.hip'), '-o', v['path']]


def _fresh(path):
    return os.path.exists(path) and all(os.path.getmtime(s) <= os.path.getmtime(path) for s in _sources())


def build(force=False, verbose=False, listing=False, variant='w2'):
    """hipcc --offload-arch=gfx950 of the kernels + C-ABI into an in-tree shared library (one of VARIANTS).  listing=True keeps the device assembly of the
    same compilation (-save-temps, build_dbg/listing*/) for tools/check_dpp_hazards.py -- the hand-written DPP statements carry their own
    wait states, which the compiler's hazard recogniser does not check."""
    v = VARIANTS[variant]
    if variant == 'w2' and os.environ.get('DL_LIB_PATH'):
        return LIB_PATH
    if not force and _fresh(v['path']) and (not listing or _fresh(v['listing'])):
        return v['path']
    cmd = _build_cmd(variant, listing)
    if verbose:
        print(' '.join(cmd))
    if listing:
        os.makedirs(os.path.dirname(v['listing']), exist_ok=True)
    subprocess.check_call(cmd, cwd=os.path.dirname(v['listing']) if listing else None)
    return v['path']


def build_all(force=False, verbose=False, listing=True):
    """Both code objects, compiled side by side (two hipcc processes; a compilation is single-threaded and takes a few minutes)."""
    todo = [k for k, v in VARIANTS.items() if force or not (_fresh(v['path']) and (not listing or _fresh(v['listing'])))]
    procs = []
    for k in todo:
        cmd = _build_cmd(k, listing)
        if verbose:
            print(' '.join(cmd))
        if listing:
            os.makedirs(os.path.dirname(VARIANTS[k]['listing']), exist_ok=True)
        procs.append((k, cmd, subprocess.Popen(cmd, cwd=os.path.dirname(VARIANTS[k]['listing']) if listing else None)))
    for k, cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return [VARIANTS[k]['path'] for k in VARIANTS]


def loaded_path():
    """The library load() chose (after a load), else the default one."""
    if SELECTED and SELECTED.get('variant') in VARIANTS:
        return VARIANTS[SELECTED['variant']]['path']
    return LIB_PATH


def _elf_sections(data):
    """{name: bytes} of a 64-bit little-endian ELF image"""
    import struct
    if data[:4] != b'\x7fELF' or data[4] != 2:
        raise DrlocoError('not a 64-bit ELF image')
    shoff, = struct.unpack_from('<Q', data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', data, 0x3A)
    sec = lambda i: struct.unpack_from('<IIQQQQIIQQ', data, shoff + i * shentsize)          # name, type, flags, addr, offset, size, ...
    str_off = sec(shstrndx)[4]
    out = {}
    for i in range(shnum):
        name_off, typ, _, _, off, size = sec(i)[:6]
        name = data[str_off + name_off:data.index(b'\0', str_off + name_off)].decode()
        out[name] = b'' if typ == 8 else data[off:off + size]          # (SHT_NOBITS occupies no file space)
    return out


def device_code_sha16(path=None):
    """Identity of the DEVICE code of a built library: sha256 over the INSTRUCTIONS and constants (`.text` + `.rodata`) of the gfx950 code object inside the library's
    `.hip_fatbin` (an uncompressed clang offload bundle), first 16 hex digits.  Comments, host code, the directory the library was built in and how it was built (with or
    without -save-temps: the code object's symbol / note sections differ, its instructions do not) leave it alone; profiles/*.json carry it so that bench.py replays
    counter-derived figures only for the kernels they were measured on.  Default: the library load() selected."""
    import hashlib
    import struct
    path = path or loaded_path()
    with open(path, 'rb') as f:
        host = _elf_sections(f.read())
    fb = host.get('.hip_fatbin')
    if fb is None:
        raise DrlocoError(f'{path}: no .hip_fatbin section (not a hipcc build?)')
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    if not fb.startswith(magic):
        raise DrlocoError(f'{path}: .hip_fatbin is not an uncompressed clang offload bundle')
    n, = struct.unpack_from('<Q', fb, len(magic))
    pos = len(magic) + 8
    h = hashlib.sha256()
    found = False
    for _ in range(n):
        off, size, tlen = struct.unpack_from('<QQQ', fb, pos)
        triple = fb[pos + 24:pos + 24 + tlen].decode()
        pos += 24 + tlen
        if 'gfx950' in triple and size:
            dev = _elf_sections(fb[off:off + size])
            for name in ('.text', '.rodata'):
                h.update(name.encode()); h.update(dev.get(name, b''))
            found = True
    if not found:
        raise DrlocoError(f'{path}: no gfx950 code object in the offload bundle')
    return h.hexdigest()[:16]


def check_dpp_hazards(variant='w2'):
    """Run tools/check_dpp_hazards.py over the listing of a product build (building it if needed) against ITS number of wait states; raises on a violation."""
    build(listing=True, variant=variant)
    v = VARIANTS[variant]
    tool = os.path.join(os.path.dirname(_HERE), 'tools', 'check_dpp_hazards.py')
    p = subprocess.run([os.environ.get('PYTHON', 'python3'), tool, v['listing'], '--need', str(v['need'])], capture_output=True, text=True)
    if p.returncode != 0:
        raise DrlocoError(f'DPP read-after-write hazard in the device code ({variant}):\n' + p.stdout[-4000:] + p.stderr[-2000:])
    return p.stdout.strip().splitlines()[-1]


def check_mfma_overlap(variant='w2'):
    """Run tools/check_mfma_overlap.py over the listing of a product build (building it if needed); raises on a violation: an MFMA whose
    destination differs from its source C and lies over an A / B operand, a relocated 4x4x1 accumulator, or a missing hand-written wait state."""
    build(listing=True, variant=variant)
    tool = os.path.join(os.path.dirname(_HERE), 'tools', 'check_mfma_overlap.py')
    p = subprocess.run([os.environ.get('PYTHON', 'python3'), tool, VARIANTS[variant]['listing']], capture_output=True, text=True)
    if p.returncode != 0:
        raise DrlocoError(f'MFMA operand overlap / wait-state violation in the device code ({variant}):\n' + p.stdout[-4000:] + p.stderr[-2000:])
    return p.stdout.strip().splitlines()[-1]


def check_snop(variant='w2'):
    """Run tools/survey_snop.py --check over the listing of a product build: raises if a kernel whose workgroups hand over with s_wakeup rests a software-managed hazard on ONE
    `s_nop N`, N >= 2, of the compiler's (another wave's s_wakeup ends an s_nop after one wait state: tools/ubench/snop_wakeup.hip; round 6's crash on the time-out path)."""
    build(listing=True, variant=variant)
    tool = os.path.join(os.path.dirname(_HERE), 'tools', 'survey_snop.py')
    p = subprocess.run([os.environ.get('PYTHON', 'python3'), tool, VARIANTS[variant]['listing'], '--check'], capture_output=True, text=True)
    if p.returncode != 0:
        bad = [l for l in p.stdout.splitlines() if '<--' in l]
        raise DrlocoError(f'multi-state s_nop in a kernel that uses s_wakeup ({variant}):\n' + '\n'.join(bad[-20:]) + p.stderr[-2000:])
    return p.stdout.strip().splitlines()[-1]


_V, _I, _P = C.c_void_p, C.c_int32, C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)       -- every symbol include/drloco_hip.h declares
    'dl_last_error': (C.c_char_p, []),
    'dl_abi_version': (C.c_int, []),
    'dl_dpp_wait_states': (C.c_int, []),
    'dl_hw_probe': (C.c_int, [_I, _I, C.POINTER(C.c_uint64)]),
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
    'dl_get_ref_offsets': (C.c_int, [_V, _P, _P]),
    'dl_set_ref_offsets': (C.c_int, [_V, _P, _P]),
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
    'dl_profile_launch_config': (C.c_int, [_V, C.POINTER(_I)]),
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


def _open(path):
    lib = C.CDLL(path)
    for name, (res, args) in {**_SIGNATURES, **_EXTRA}.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.dl_abi_version() != abi.DL_ABI_VERSION:
        raise DrlocoError(f'ABI version mismatch between drloco_amd/abi.py and {os.path.basename(path)}')
    for which, struct in enumerate((abi.ModelDesc, abi.RefsDesc, abi.Config)):
        if lib.dl_abi_sizeof(which) != C.sizeof(struct):
            raise DrlocoError(f'struct size mismatch for {struct.__name__}')
    return lib


def hw_probe(lib=None, device=-1, iters=64):
    """dl_hw_probe on `device` (-1: the current one): dict(stale_dpp=[0, 1, 2 wait states], stale_mfma=[single s_nop 7, 8 x v_nop, two s_nop], lane_reads_per_cell, cells)
    -- or None without a HIP device."""
    lib = lib or load()
    out = (C.c_uint64 * 8)()
    rc = lib.dl_hw_probe(device, iters, out)
    if rc == abi.DL_E_NODEVICE:
        return None
    if rc != 0:
        raise DrlocoError(f'dl_hw_probe failed: {lib.dl_last_error().decode()}')
    return dict(stale_dpp=[int(out[0]), int(out[1]), int(out[2])], stale_mfma=[int(out[3]), int(out[4]), int(out[5])], lane_reads_per_cell=int(out[6]), cells=int(out[7]))


SELECTED = None          # after load(): dict(variant, why, probe) -- what bench.py reports


def load():
    """Load the HIP library; raises DrlocoError if it has not been built.  Which code object: DL_LIB_PATH (an experiment build) as it is; otherwise the default,
    spec-conformant one (two wait states in front of the hand-written DPP reads) unless this process's device PROVES that one is enough -- dl_hw_probe shows stale reads
    with no wait (the test can fail) and none with one state --, in which case the one-state build is used.  DL_DPP_WAIT=2 / 1 in the environment forces a choice
    (1 still has to pass the probe: dl_create of that build runs it and refuses otherwise)."""
    global _lib, SELECTED
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DrlocoError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    lib = _open(LIB_PATH)
    SELECTED = dict(variant='experiment build (DL_LIB_PATH)' if os.environ.get('DL_LIB_PATH') else 'w2', dpp_wait_states=int(lib.dl_dpp_wait_states()), why='default', probe=None)
    want = os.environ.get('DL_DPP_WAIT', 'auto')
    fast = VARIANTS['w1']['path']
    if not os.environ.get('DL_LIB_PATH') and want != '2' and os.path.exists(fast):
        probe = None
        try:
            probe = hw_probe(lib)
        except DrlocoError as e:
            SELECTED['why'] = f'probe failed ({e}): spec-conformant build'
        if probe is None:
            if SELECTED['why'] == 'default':
                SELECTED['why'] = 'no HIP device in this process: spec-conformant build'
        elif probe['stale_dpp'][0] > 0 and probe['stale_dpp'][1] == 0 and probe['stale_dpp'][2] == 0:
            lib = _open(fast)
            SELECTED = dict(variant='w1', dpp_wait_states=int(lib.dl_dpp_wait_states()), probe=probe,
                            why=f"this device needs ONE wait state in front of a DPP read: {probe['stale_dpp'][0]} stale reads with none, 0 with one, of {probe['cells']} x {probe['lane_reads_per_cell']} lane-reads (dl_hw_probe)")
        else:
            SELECTED.update(probe=probe, why=f"probe: stale DPP reads with 0 / 1 / 2 wait states = {probe['stale_dpp']}: the one-state build is not proven on this device, spec-conformant build kept")
    _lib = lib
    return lib


def check(rc):
    if rc == abi.DL_E_FAULT:
        raise DrlocoFault(f'drloco_hip fault: {load().dl_last_error().decode()}')
    if rc != 0:
        raise DrlocoError(f'drloco_hip error {rc}: {load().dl_last_error().decode()}')
