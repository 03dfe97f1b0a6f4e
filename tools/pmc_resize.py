"""A few launches of crop + resize at config-2 size (for the counter passes of tools/pmc_kernel.sh and tools/kernel_power.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meshflow_amd import ops, synthetic
H, W, F = 1080, 1920, 300
dev = torch.device('cuda:0')
src = synthetic.frames_torch(F, H, W, dev, seed=0)
dst = torch.empty_like(src)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ops.crop_resize(src, (13, 11, 1909, 1068), out=dst)
torch.cuda.synchronize()
print('done')
