"""Walks warp_kernel's gfx950 assembly along ONE execution path and counts what a wavefront issues on it.

    python tools/warp_walk.py DECISIONS [-D...]      e.g.  python tools/warp_walk.py NNTN... -v

DECISIONS: one letter per conditional branch in the order the walk meets them -- T = taken, N = not taken ('?' stops and prints
where the walk stands, to extend the string by hand).  Unconditional branches are followed, s_endpgm ends the walk.
-v prints every instruction.  Classes: VALU (with the issue-cost classes of tools/ubench_issue.hip), SALU (s_* except loads;
branches and s_waitcnt / s_nop included: each takes an issue slot of the wavefront), SMEM, LDS, VMEM.
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fno-slp-vectorize', '-S',
         '--cuda-device-only']
TWO = {'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_mov_b32', 'v_lshrrev_b32',
       'v_add_f32', 'v_sub_f32', 'v_mul_f32', 'v_fma_f32', 'v_fmac_f32', 'v_fmamk_f32', 'v_fmaak_f32', 'v_not_b32'}


def cost(op):
    o = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if not o.startswith('v_'):
        return 0.0
    if o in TWO:
        return 2.0
    if o.startswith('v_rcp_f64'):
        return 16.0
    return 4.0


def main():
    decisions = sys.argv[1] if len(sys.argv) > 1 else '?'
    verbose = '-v' in sys.argv
    extra = [a for a in sys.argv[2:] if a.startswith('-D')]
    asm = [a[6:] for a in sys.argv[2:] if a.startswith('--asm=')]           # reuse an assembly listing made earlier
    if asm and os.path.exists(asm[0]):
        lines = open(asm[0]).read().split('\n')
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = asm[0] if asm else os.path.join(tmp, 'warp.s')
            subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + extra + [os.path.join(REPO, 'meshflow_amd', 'csrc', 'warp.hip'), '-o', out],
                           check=True, stderr=subprocess.DEVNULL)
            lines = open(out).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith('_ZN2mf11warp_kernel')][0]
    code, labels = [], {}
    for l in lines[start + 1:]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = len(code)
        elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
            code.append(l.strip())
        if 'codeLenInByte' in l:
            break
    meta = [l.strip('; ').strip() for l in lines[start:] if re.search(r'; (NumVgprs|TotalNumSgprs|codeLenInByte|Occupancy|ScratchSize)', l)][:5]
    print('kernel resources:', ', '.join(meta))
    names = {v: k for k, v in labels.items()}
    pc, d, n = 0, 0, Counter()
    hist, cyc = Counter(), 0.0
    while True:
        ins = code[pc]
        op = ins.split()[0]
        if pc in names and verbose:
            print(f'{names[pc]}:')
        if verbose:
            c = cost(op)
            print(f'    {c:4.1f}  {ins}' if c else f'          {ins}')
        if op.startswith('v_'):
            n['VALU'] += 1; cyc += cost(op); hist[re.sub(r'_(e32|e64)$', '', op)] += 1
        elif op.startswith('s_load') or op.startswith('s_buffer_load'):
            n['SMEM'] += 1
        elif op.startswith('s_'):
            n['SALU'] += 1; hist[op] += 1
        elif op.startswith('ds_'):
            n['LDS'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
            n['VMEM'] += 1
        if op == 's_endpgm':
            break
        if op == 's_branch':
            pc = labels[ins.split()[1]]
            continue
        if op.startswith('s_cbranch'):
            if d >= len(decisions) or decisions[d] == '?':
                print(f'-- stopped at decision {d}: {ins}   (walked {sum(n.values())} instructions; last label {max((k for k in names if k <= pc), default=0) and names[max(k for k in names if k <= pc)]})')
                for k in range(max(0, pc - 12), pc + 1):
                    print('      ', code[k])
                break
            taken = decisions[d] == 'T'
            d += 1
            if taken:
                pc = labels[ins.split()[1]]
                continue
        pc += 1
    print(f'\nper wavefront on this path: {n["VALU"]} VALU ({cyc:.0f} issue cycles by class), {n["SALU"]} SALU/branch/waitcnt, {n["SMEM"]} SMEM, '
          f'{n["LDS"]} LDS, {n["VMEM"]} VMEM   [{d} conditional branches]')
    print('VALU:', ', '.join(f'{k} {v}' for k, v in hist.most_common() if k.startswith('v_')))
    print('SALU:', ', '.join(f'{k} {v}' for k, v in hist.most_common() if k.startswith('s_')))


if __name__ == '__main__':
    main()
