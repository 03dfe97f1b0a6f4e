"""Build-time guard of the inline-asm invariants of warp_kernel, checked on the ISA of the BUILT library (tests/test_isa_guard.py), and
the instruction listing of one execution path of it (what used to be hand-kept under profiles/*_isa.txt).

    python tools/isa_guard.py [lib.so]                       checks; prints resources and what was verified; exit code 1 on a violation
    python tools/isa_guard.py lib.so --hazards-only          experiment builds (paths compiled out): only the in-flight scalar-load hazard
    python tools/isa_guard.py [lib.so] --walk NNTN...        + every instruction a wavefront issues on ONE path: T / N per conditional branch met

Why.  Two places in csrc/warp.hip issue loads the COMPILER DOES NOT KNOW ABOUT:
  * the speculative matrix load -- `s_load_dwordx16` + `s_load_dwordx2` from inline asm at the top of the kernel.  Scalar loads return
    out of order and nothing tracks these two, so on EVERY path from them an `s_waitcnt lgkmcnt(0)` must come before the first instruction
    that WRITES one of their destination registers -- otherwise the late load overwrites the new value (round 5: about one corrupted
    launch in 200 in an instantiation where the registers were dead and had been handed to the plan words' loads).
  * the byte taps -- runs of `ds_read_u8` / `ds_read_u8_d16_hi` inside one asm statement that ends with its own `s_waitcnt lgkmcnt(0)`: a
    run must reach that wait with nothing but tap loads in between.
And the kernel's occupancy rests on three numbers: zero scratch, <= 64 VGPRs (8 wavefronts per SIMD), <= 80 SGPRs (8 one-wavefront
workgroups per CU, MI355X_MICROARCH.md).  A hipcc bump can break any of these silently; this turns it into a failing CPU test."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WARP_TRUE = '_ZN2mf11warp_kernelILb1EEE'
WARP_FALSE = '_ZN2mf11warp_kernelILb0EEE'
LIMITS = {'vgpr_count': 64, 'sgpr_count': 80, 'private_segment_fixed_size': 0, 'vgpr_spill_count': 0}


class Kernel:
    """Instructions of one kernel (llvm-objdump --symbolize-operands): `code[i]` = text, `labels[name]` = index of the instruction it marks."""

    def __init__(self, name, code, labels):
        self.name, self.code, self.labels = name, code, labels

    def successors(self, i):
        op = self.code[i].split()[0]
        if op == 's_endpgm':
            return []
        if op == 's_branch':
            return [self.labels[self.code[i].split()[1]]]
        if op.startswith('s_cbranch'):
            return [i + 1, self.labels[self.code[i].split()[1]]]
        if op in ('s_setpc_b64', 's_swappc_b64'):
            raise AssertionError(f'{self.name}: indirect jump at {i}: {self.code[i]}')
        return [i + 1]


def disassemble(so_path):
    """{symbol: Kernel} for every kernel of the library whose name contains 'warp_kernel'."""
    out = {}
    for co in codeobj.code_objects(so_path):
        md = codeobj.kernel_metadata(co)
        if not any('warp_kernel' in k for k in md):
            continue
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(co); f.flush()
            txt = subprocess.run([os.path.join(codeobj.LLVM, 'llvm-objdump'), '-d', '--symbolize-operands', '--no-show-raw-insn', f.name],
                                 check=True, capture_output=True, text=True).stdout
        cur = None
        for line in txt.split('\n'):
            m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
            if m and re.fullmatch(r'L\d+', m.group(1)):             # a branch target inside the current kernel
                if cur is not None:
                    cur.labels[m.group(1)] = len(cur.code)
                continue
            if m:
                cur = None
                if 'warp_kernel' in m.group(1):
                    cur = out[m.group(1)] = Kernel(m.group(1), [], {})
                continue
            if cur is None:
                continue
            if line.startswith('\t'):
                cur.code.append(re.sub(r'\s*//.*$', '', line.strip()))
    return out


def _sgprs(operand):
    """SGPR numbers an operand like s[4:19] / s7 names (empty for anything else)."""
    m = re.fullmatch(r's\[(\d+):(\d+)\]', operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r's(\d+)', operand)
    return {int(m.group(1))} if m else set()


def _dest_sgprs(ins):
    """SGPRs an instruction writes: its first operand when that is an SGPR (compares, branches, waits and stores write none; the
    implicit destinations -- scc, vcc, exec, m0 -- are no SGPR numbers)."""
    parts = ins.replace(',', ' ').split()
    op = parts[0]
    if len(parts) < 2 or op.startswith(('s_cmp', 's_bitcmp', 's_cbranch', 's_branch', 's_waitcnt', 's_nop', 's_endpgm', 's_barrier', 's_sleep',
                                        's_setprio', 's_store', 's_buffer_store', 'global_store', 'buffer_store', 'ds_write', 'v_cmpx')):
        return set()
    dest = _sgprs(parts[1])
    if op.startswith('v_cmp') or op.startswith(('v_add_co', 'v_sub_co', 'v_subrev_co', 'v_addc_co', 'v_subb_co', 'v_mad_u64_u32', 'v_mad_i64_i32', 'v_div_scale')):
        for p in parts[1:3]:                                      # VOP3 forms: sdst is the first or second operand
            dest |= _sgprs(p)
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
        dest |= _sgprs(parts[1])
    return dest


def _waits_lgkm0(ins):
    return ins.startswith('s_waitcnt') and 'lgkmcnt(0)' in ins


def _check_scalar_load(k, i0, n_loads):
    """No path from the scalar load(s) at i0 .. i0 + n_loads - 1 writes one of their destination registers before an s_waitcnt lgkmcnt(0)
    (scalar loads return out of order: only a zero count covers them).  Returns (instructions walked, covering waits)."""
    guarded = set()
    for j in range(n_loads):
        guarded |= _sgprs(k.code[i0 + j].replace(',', ' ').split()[1])
    seen, todo, checked, waits = set(), [i0 + n_loads], 0, set()
    while todo:
        i = todo.pop()
        if i in seen:
            continue
        seen.add(i)
        ins = k.code[i]
        if _waits_lgkm0(ins):
            waits.add(i)
            continue
        hit = _dest_sgprs(ins) & guarded
        assert not hit, (f'{k.name}: instruction {i} `{ins}` writes s{sorted(hit)} while the scalar load at instruction {i0} `{k.code[i0]}` may '
                         f'still be in flight: no s_waitcnt lgkmcnt(0) on the way')
        checked += 1
        todo.extend(k.successors(i))
    return checked, len(waits)


def check_speculative_load(k):
    """The speculative matrix load is the FIRST s_load_dwordx16 of the kernel (record offset 0x48, in front of the plan words' loads), with
    its dwordx2 sibling (offset 0x88) right behind it on the same address pair.  Every path from them reaches `s_waitcnt lgkmcnt(0)` -- or
    the end of the program -- before any instruction writes one of their 18 destination registers.  The same is checked for every other
    16-dword scalar load of the kernel (the compiler's own matrix loads: it tracks those itself, so this must hold trivially)."""
    x16 = [i for i, ins in enumerate(k.code) if ins.startswith('s_load_dwordx16')]
    assert x16, f'{k.name}: no s_load_dwordx16 at all: the speculative matrix load is gone'
    i0 = x16[0]
    first_plan = next(i for i, ins in enumerate(k.code) if ins.startswith('s_load_dwordx4') and 'offset:0x0' in ins)      # pw[0..3]: base + scalar offset
    assert i0 < first_plan and k.code[i0].rstrip().endswith('0x48'), f'{k.name}: the first s_load_dwordx16 (instruction {i0}: {k.code[i0]}) is not in front of the plan load ({first_plan})'
    assert k.code[i0 + 1].startswith('s_load_dwordx2') and k.code[i0 + 1].rstrip().endswith('0x88'), f'{k.name}: the x2 sibling does not follow: {k.code[i0 + 1]}'
    dst16, base = k.code[i0].replace(',', ' ').split()[1:3]
    assert k.code[i0 + 1].replace(',', ' ').split()[2] == base, 'the two speculative loads use different base registers'
    guarded = _sgprs(dst16) | _sgprs(k.code[i0 + 1].replace(',', ' ').split()[1])
    assert len(guarded) == 18 and not (guarded & _sgprs(base)), f'destinations {sorted(guarded)} overlap the address pair {base}'
    checked, waits = _check_scalar_load(k, i0, 2)
    for i in x16[1:]:
        _check_scalar_load(k, i, 1)
    return (f'speculative load at instruction {i0} -> {dst16} + 2: {checked} instructions on all paths up to {waits} covering s_waitcnt lgkmcnt(0), '
            f'no write to its 18 registers; the {len(x16) - 1} other s_load_dwordx16 likewise')


def check_tap_blocks(k):
    """Every run of LDS byte-tap loads ends in `s_waitcnt lgkmcnt(0)` with nothing but tap loads on the way (the asm statements of
    taps_pair / taps_clamped: loads and wait are one unit)."""
    runs, i, n = 0, 0, len(k.code)
    while i < n:
        if k.code[i].startswith('ds_read_u8'):
            j = i
            while j < n and k.code[j].startswith('ds_read_u8'):
                assert j not in k.labels.values() or j == i, f'{k.name}: a branch target inside the tap loads at instruction {j}'
                j += 1
            assert j < n and _waits_lgkm0(k.code[j]), f'{k.name}: tap loads {i}..{j - 1} are followed by `{k.code[j]}`, not by s_waitcnt lgkmcnt(0)'
            assert (j - i) % 12 == 0, f'{k.name}: a run of {j - i} tap loads (whole pixels are 12)'
            runs += 1
            i = j
        i += 1
    assert runs >= 4, f'{k.name}: only {runs} tap-load runs found'
    return f'{runs} runs of ds_read_u8 / ds_read_u8_d16_hi, each closed by its own s_waitcnt lgkmcnt(0)'


def check_resources(md, name):
    got = {key: md.get(key, 0) for key in LIMITS}
    for key, limit in LIMITS.items():
        assert got[key] <= limit, f'{name}: {key} = {got[key]} exceeds {limit}'
    return ', '.join(f'{key} {got[key]} (<= {LIMITS[key]})' for key in LIMITS)


def check_hazards_only(so_path):
    """For EXPERIMENT builds (tools/phase_profile.sh: code paths compiled out, the speculative load possibly with them): only the hazard
    itself -- no 16-dword scalar load of either warp kernel has a destination register rewritten before an s_waitcnt lgkmcnt(0)."""
    report = []
    for sym, k in disassemble(so_path).items():
        x16 = [i for i, ins in enumerate(k.code) if ins.startswith('s_load_dwordx16')]
        for i in x16:
            n = 2 if k.code[i + 1].startswith('s_load_dwordx2') and k.code[i + 1].replace(',', ' ').split()[2] == k.code[i].replace(',', ' ').split()[2] else 1
            _check_scalar_load(k, i, n)
        report.append(f'{sym[:40]}...: {len(x16)} s_load_dwordx16, none rewritten in flight')
    return report


def check_library(so_path):
    """All checks on one built library; returns the report lines (raises AssertionError on a violation)."""
    kernels = disassemble(so_path)
    meta = codeobj.all_kernels(so_path)
    report = []
    names = {WARP_TRUE: None, WARP_FALSE: None}
    for sym in kernels:
        for prefix in names:
            if sym.startswith(prefix):
                names[prefix] = sym
    assert all(names.values()), f'warp_kernel<true> / <false> not found in {so_path}: {list(kernels)}'
    for prefix, sym in names.items():
        report.append(f'{sym[:40]}...: ' + check_resources(meta[sym], sym))
    kt, kf = kernels[names[WARP_TRUE]], kernels[names[WARP_FALSE]]
    report.append('warp_kernel<true>: ' + check_speculative_load(kt))
    report.append('warp_kernel<true>: ' + check_tap_blocks(kt))
    # the other instantiation must NOT carry the speculative load (its registers would be dead: round 5's bug)
    first_plan = next(i for i, ins in enumerate(kf.code) if ins.startswith('s_load_dwordx4') and 'offset:0x0' in ins)
    x16 = [i for i, ins in enumerate(kf.code) if ins.startswith('s_load_dwordx16')]
    assert all(i > first_plan for i in x16), 'warp_kernel<false> issues an s_load_dwordx16 in front of its plan load: the speculative load leaked into it'
    for i in x16:
        _check_scalar_load(kf, i, 1)
    report.append(f'warp_kernel<false>: no speculative load in front of the plan load; its {len(x16)} s_load_dwordx16 are covered')
    return report


# ---- path listing ------------------------------------------------------------------------------------------------------------------
TWO = {'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_mov_b32', 'v_lshrrev_b32',
       'v_add_f32', 'v_sub_f32', 'v_mul_f32', 'v_fma_f32', 'v_fmac_f32', 'v_fmamk_f32', 'v_fmaak_f32', 'v_not_b32'}


def issue_cost(op):
    """VALU issue cycles by class (tools/ubench_issue.hip, profiles/r02_ubench_issue.txt)."""
    o = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if not o.startswith('v_'):
        return 0.0
    if o in TWO:
        return 2.0
    return 16.0 if o.startswith('v_rcp_f64') else 4.0


def walk(k, decisions):
    """Follows ONE path: `decisions` = T / N per conditional branch met.  Returns (lines, counts, cycles, branches used)."""
    from collections import Counter
    at = {v: name for name, v in k.labels.items()}
    pc, d, n, cyc, lines = 0, 0, Counter(), 0.0, []
    while True:
        ins = k.code[pc]
        op = ins.split()[0]
        if pc in at:
            lines.append(f'{at[pc]}:')
        c = issue_cost(op)
        lines.append(f'    {c:4.1f}  {ins}' if c else f'          {ins}')
        if op.startswith('v_'):
            n['VALU'] += 1; cyc += c
        elif op.startswith(('s_load', 's_buffer_load')):
            n['SMEM'] += 1
        elif op.startswith('s_'):
            n['SALU'] += 1
        elif op.startswith('ds_'):
            n['LDS'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            n['VMEM'] += 1
        if op == 's_endpgm':
            break
        if op == 's_branch':
            pc = k.labels[ins.split()[1]]
            continue
        if op.startswith('s_cbranch'):
            if d >= len(decisions):
                lines.append(f'-- stopped at conditional branch {d}: extend the decision string')
                break
            taken = decisions[d] == 'T'
            d += 1
            if taken:
                pc = k.labels[ins.split()[1]]
                continue
        pc += 1
    return lines, n, cyc, d


def main():
    argv = sys.argv[1:]
    decisions = None
    if '--walk' in argv:
        at = argv.index('--walk')
        decisions = argv[at + 1]
        del argv[at:at + 2]
    so = argv[0] if argv else os.path.join(REPO, 'meshflow_amd', 'libmeshflow_hip.so')
    hazards_only = '--hazards-only' in argv
    if hazards_only:
        argv.remove('--hazards-only')
        so = argv[0] if argv else so
    try:
        for line in (check_hazards_only(so) if hazards_only else check_library(so)):
            print('ok  ' + line)
    except AssertionError as e:
        print('VIOLATION  ' + str(e))
        raise SystemExit(1)
    if decisions is not None:
        kernels = disassemble(so)
        k = next(v for s, v in kernels.items() if s.startswith(WARP_TRUE))
        lines, n, cyc, d = walk(k, decisions)
        print('\n'.join(lines))
        print(f'\nper wavefront on this path: {n["VALU"]} VALU ({cyc:.0f} issue cycles by class), {n["SALU"]} SALU/branch/waitcnt, {n["SMEM"]} SMEM, '
              f'{n["LDS"]} LDS, {n["VMEM"]} VMEM   [{d} conditional branches: {decisions[:d]}]')


if __name__ == '__main__':
    main()
