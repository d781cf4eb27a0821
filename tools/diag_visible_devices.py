import os, sys, subprocess
code = r'''
import os, ctypes as C
var = os.environ.get('WHICH')
if var: os.environ[var] = ''
import torch
print(var, 'torch.cuda.device_count()', torch.cuda.device_count(), 'is_available', torch.cuda.is_available())
from drloco_amd import abi, lib, mocap, models
L = lib.load(); h = C.c_void_p(); m = models.make_model(); r = mocap.RefTable.load(); d = r.as_desc(); c = abi.default_config()
rc = L.dl_create(C.byref(m), C.byref(d), C.byref(c), 4, 0, C.byref(h))
print(var, 'dl_create rc', rc, L.dl_last_error().decode()[:300], lib.SELECTED['why'][:80])
'''
for var in ('', 'CUDA_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):
    env = dict(os.environ, WHICH=var)
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print(p.stdout, p.stderr[-500:])
# and without torch imported first
code2 = code.replace("import torch\nprint(var, 'torch.cuda.device_count()', torch.cuda.device_count(), 'is_available', torch.cuda.is_available())\n", "")
p = subprocess.run([sys.executable, '-c', code2], env=dict(os.environ, WHICH='CUDA_VISIBLE_DEVICES'), capture_output=True, text=True)
print('no torch:', p.stdout, p.stderr[-500:])
