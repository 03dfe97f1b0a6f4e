"""Footprint classes of a workload and their window layouts (the plan's path bits, csrc/mf_common.h): hot / pair / multi / border / rest,
and within each the share with a COMPACT (one-load, 9 x 112-byte) window.

    python tools/class_census.py [cfg2|cfg3|cfg4shard ...] [--lib path/to/lib.so]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = {'cfg2': (1080, 1920, 300, 16, 16, 10, 100), 'cfg3': (1080, 1920, 600, 32, 32, 30, 200), 'cfg4shard': (2160, 3840, 150, 16, 16, 10, 100)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    libpath = os.path.join(REPO, 'meshflow_amd', 'libmeshflow_hip.so')
    if '--lib' in sys.argv:
        libpath = sys.argv[sys.argv.index('--lib') + 1]
        args = [a for a in args if a != libpath]
    lib = ctypes.CDLL(os.path.abspath(libpath))
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.mf_cell_table_bytes.restype = ctypes.c_size_t
    lib.mf_cell_table_bytes.argtypes = [i] * 5
    lib.mf_cell_table_f64.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp]
    dev = torch.device('cuda:0')
    for wl in (args or ['cfg2', 'cfg3']):
        H, W, F, R, C, omega, iters = WORK[wl]
        disp, hom = synthetic.motion(F, R, C, seed=0)
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device='cuda:0')
        d_unstab = torch.from_numpy(disp).to(dev)
        d_stab = s._stabilized_vertex_displacements_device(d_unstab, W, H, 0, hom)
        table = torch.empty(lib.mf_cell_table_bytes(F, W, H, R, C), dtype=torch.uint8, device=dev)
        crop = torch.empty((F, 4), dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.mf_cell_table_f64(d_unstab.data_ptr(), d_stab.data_ptr(), F, W, H, R, C, table.data_ptr(), crop.data_ptr(), status.data_ptr(), st) == 0
        torch.cuda.synchronize()
        nrec = F * R * C
        plan_off = (nrec * (32 * 8 + 8 + (16 + 12) * 4) + 15) & ~15
        nfp = F * ((H + 7) // 8) * ((W + 31) // 32)
        plan = table[plan_off:plan_off + 16 * nfp].view(torch.int32).view(nfp, 4).cpu().numpy().view(np.uint32)
        region = table[plan_off + 16 * nfp:plan_off + 24 * nfp].view(torch.int32).view(nfp, 2).cpu().numpy().view(np.uint32)[:, 0]
        x, y, z = plan[:, 0], plan[:, 1], plan[:, 2]
        hot = ((x >> 16) & 0x2000) != 0
        border = (((x >> 16) & (0x4000 | 0x1000)) == 0x1000) & ~hot
        pair = ((y & 0x2000) != 0) & ~hot & ~border
        multi = ((z & 0x2000) != 0) & ~hot & ~border & ~pair
        rest = ~(hot | border | pair | multi)
        compact, staged = (region & 0x20000000) != 0, (region & 0x80000000) != 0
        print(f'{wl}: {nfp} footprints; COMPACT windows {compact.mean():.4f}, staged {staged.mean():.4f}')
        for name, m in (('hot', hot), ('pair', pair), ('  pair lane-uniform (FAST)', pair & ((y & 1) != 0)), ('  pair vertical edge', pair & ((y & 2) != 0)),
                        ('multi', multi), ('border', border), ('rest', rest)):
            print(f'   {name:28s} {m.mean():8.4f} of the footprints, COMPACT {compact[m].mean() if m.any() else 0:.4f} of them')
        # the multi class by list length and by how many of its entries carry a two-edge code (bit 3 of the 6-bit code in e[4 + i])
        w = plan[:, 3]
        cnt = ((z >> 6) & 3) + 1
        codes = np.stack([z & 0x3F, (z >> 16) & 0x3F, w & 0x3F, (w >> 16) & 0x3F], axis=1)
        ent = np.stack([x & 0xFFFF, x >> 16, y & 0xFFFF, y >> 16], axis=1)
        is_in = (ent & 0x8000) != 0
        for k in (2, 3, 4):
            mk = multi & (cnt == k)
            if not mk.any():
                continue
            two = ((codes[mk][:, :k] & 8) != 0) & ~is_in[mk][:, :k]
            last_in = is_in[mk][np.arange(mk.sum()), k - 1]
            print(f'   multi with {k} cells: {mk.mean():.4f} of the footprints ({mk.sum() / max(multi.sum(), 1):.3f} of multi); last entry IN {last_in.mean():.3f}; '
                  f'two-edge entries per list: ' + ', '.join(f'{j}: {(two.sum(1) == j).mean():.3f}' for j in range(k + 1)))


if __name__ == '__main__':
    main()
