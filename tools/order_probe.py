import sys, os, ctypes
sys.path.insert(0, '/root/repo')
from meshflow_amd import _lib
n = ctypes.c_int(0); print('mf_device_count rc', _lib.lib.mf_device_count(ctypes.byref(n)), n.value)
p = ctypes.c_void_p(); print('mf_malloc', _lib.lib.mf_malloc(ctypes.byref(p), 1024))
import torch
print('torch avail after lib init:', torch.cuda.is_available(), torch.cuda.device_count())
try:
    print(torch.zeros(3, device='cuda:0'))
except Exception as e:
    print('ERR', e)
