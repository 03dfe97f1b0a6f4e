"""tools/pcie_staged's `direct` mode from INSIDE a Python process that has torch loaded and the GPU initialised (what the host pipeline's
callers are): does the process environment cost duplex PCIe rate?   python tools/pcie_staged_in_python.py [torch|notorch] MiB U D"""
import ctypes, os, sys
mode = sys.argv[1]
if mode == 'torch':
    import torch
    torch.zeros(1, device='cuda:0')
    import numpy as np
    np.fft.rfft(np.ones(1024))
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libpcie_staged.so'))
args = [b'pcie_staged'] + [a.encode() for a in sys.argv[2:]]
argv = (ctypes.c_char_p * len(args))(*args)
getattr(lib, '_Z16pcie_staged_mainiPPc')(len(args), argv)      # (C++ linkage: the tool's main under another name)
