"""Build-time guard of warp_kernel's inline-asm invariants, on the ISA of the library the suite just built (tools/isa_guard.py):
zero scratch / <= 64 VGPRs / <= 80 SGPRs in both instantiations; no instruction writes a destination register of the speculative
`s_load_dwordx16` + `s_load_dwordx2` (or of any other 16-dword scalar load) before an `s_waitcnt lgkmcnt(0)` on any path; the speculative
load exists only in the instantiation that has a hot path; every run of byte-tap LDS loads is closed by its own wait.  A hipcc bump or an
edit that breaks one of these fails HERE, on the CPU, instead of as one corrupted launch in 200 on the GPU (round 5)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import isa_guard  # noqa: E402

LIB = os.path.join(REPO, 'meshflow_amd', 'libmeshflow_hip.so')
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-fno-slp-vectorize']      # csrc/Makefile's


def test_built_library_keeps_the_invariants():
    report = isa_guard.check_library(LIB)
    assert len(report) == 5 and 'speculative load at instruction' in report[2] and 'runs of ds_read_u8' in report[3]
    print('\n'.join(report))


def test_header_exports_match_and_warp_kernels_have_no_scratch():
    import codeobj
    ks = codeobj.all_kernels(LIB)
    warp = {k: v for k, v in ks.items() if 'warp_kernel' in k}
    assert len(warp) == 2
    for name, md in warp.items():
        assert md['private_segment_fixed_size'] == 0 and md['vgpr_count'] <= 64 and md['sgpr_count'] <= 80, (name, md)
        assert md['wavefront_size'] == 64 and md['max_flat_workgroup_size'] == 64


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc to build the deliberately broken variant')
def test_guard_fails_on_round_5s_bug(tmp_path):
    """-DMF_GUARD_SELFTEST puts the speculative load into warp_kernel<false> as well -- the configuration that corrupted about one launch
    in 200 at the end of round 5 (the compiler hands the dead destination registers to the next loads).  The guard must see both the
    leak and the register write that made it a bug."""
    obj = str(tmp_path / 'warp_selftest.o')
    subprocess.run([HIPCC] + FLAGS + ['-DMF_GUARD_SELFTEST=1', '-c', os.path.join(REPO, 'meshflow_amd', 'csrc', 'warp.hip'), '-o', obj], check=True)
    with pytest.raises(AssertionError, match='speculative load leaked'):
        isa_guard.check_library(obj)
    kernels = isa_guard.disassemble(obj)
    kf = next(k for sym, k in kernels.items() if sym.startswith(isa_guard.WARP_FALSE))
    with pytest.raises(AssertionError, match='may still be in flight'):
        isa_guard.check_speculative_load(kf)
    kt = next(k for sym, k in kernels.items() if sym.startswith(isa_guard.WARP_TRUE))
    isa_guard.check_speculative_load(kt)                      # (the instantiation with the hot path is still fine in that build)


def test_walk_lists_a_path_from_the_built_library():
    """The path listing that replaces the hand-kept profiles/*_isa.txt: the first decisions of the hot path (the footprint exists, its
    window is staged) can be followed on the built ISA, and the listing accounts every instruction it meets."""
    kernels = isa_guard.disassemble(LIB)
    k = next(v for sym, v in kernels.items() if sym.startswith(isa_guard.WARP_TRUE))
    lines, n, cyc, used = isa_guard.walk(k, 'NN')
    assert used == 2 and n['SMEM'] >= 8 and n['SALU'] >= 40 and any('s_load_dwordx16' in l for l in lines)
    assert lines[-1].startswith('-- stopped at conditional branch 2')
