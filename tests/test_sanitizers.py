"""CPU sanitizer legs (SURVEY.md 5 'race detection / sanitizers': the reference has none; GPU AddressSanitizer is not available on this pool, so the
sanitizers run where the same SOURCE runs on the host):
  * the oracle (oracle/dl_oracle.c) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle SAN=1`), the oracle's own test files
    run against that build in a child process (ASan has to be the first library of the process: LD_PRELOAD);
  * the kernel source compiled for the host (tests/host_emu: drloco_amd/csrc/dl_core.hpp, dl_env.hpp, dl_group.hpp, dl_group_env.hpp under the fiber
    emulation of a wave) rebuilt with UndefinedBehaviorSanitizer (ASan and the fibers' hand-made stacks do not mix), the host-emulation tests run
    against it.
Both builds abort at the first finding (-fno-sanitize-recover=all), so a green child run means a clean run.  DL_SKIP_SANITIZERS=1 skips the legs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(os.environ.get('DL_SKIP_SANITIZERS') == '1', reason='DL_SKIP_SANITIZERS=1')


def _child(env_extra, args, timeout):
    env = dict(os.environ, **env_extra)
    env.pop('PYTEST_CURRENT_TEST', None)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider', '-m', 'not gpu'] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = p.stdout[-3000:] + p.stderr[-3000:]
    assert p.returncode == 0, tail
    assert 'runtime error' not in tail and 'AddressSanitizer' not in tail, tail
    return p.stdout


@pytest.mark.timeout(900)
def test_oracle_under_asan_and_ubsan():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'SAN=1'])
    libasan = subprocess.check_output(['gcc', '-print-file-name=libasan.so'], text=True).strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), libasan
    out = _child({'LD_PRELOAD': libasan, 'DL_ORACLE_LIB': os.path.join(ROOT, 'oracle', 'libdl_oracle_san.so'),
                  'ASAN_OPTIONS': 'detect_leaks=0:abort_on_error=1', 'UBSAN_OPTIONS': 'print_stacktrace=1:halt_on_error=1'},
                 ['tests/test_oracle_golden.py', 'tests/test_oracle_physics.py'], 800)
    assert ' passed' in out, out[-500:]


@pytest.mark.timeout(900)
def test_kernel_source_on_host_under_ubsan():
    out = _child({'DL_EMU_SANITIZE': '1', 'UBSAN_OPTIONS': 'print_stacktrace=1:halt_on_error=1'},
                 ['tests/test_host_logic.py', 'tests/test_split_protocol_emu.py', '-k', 'kernel_source or schedule_independent or 19dof'], 800)          # incl. the two-wave hand-over of the split workgroups
    assert ' passed' in out, out[-500:]
    assert os.path.exists(os.path.join(ROOT, 'tests', 'host_emu', 'libdl_emu_ubsan.so'))
