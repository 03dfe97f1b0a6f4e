"""Instruction listing of the warp kernel's hot path (one certified single-owner footprint) with issue-cost classes.

    python tools/warp_hot_path.py > profiles/r02_warp_hot_path_isa.txt

Compiles meshflow_amd/csrc/warp.hip to gfx950 assembly with the library's flags, walks the basic blocks a wavefront
executes when its footprint has ONE IN cell, a plan-certified denominator (MF_PLAN_UNIT) and a certified interior staged
window (MF_REGION_DEEP) -- 66 % of the cfg2 footprints -- and prints every instruction with its class.  The block chain
below belongs to THIS compile (labels move when the source changes; the script checks the landmarks it relies on).
Cost classes from tools/ubench_issue.hip (profiles/r02_ubench_issue*.txt, cycles per wave64 instruction with 8 waves per SIMD):
  2  v_and/or/xor/add/sub/mov/lshrrev_b32, f32 add/sub/mul/fma(c/mk)         4  every other 32-bit VALU op (lshl, bfe, perm,
  alignbyte, mad24, dot4, cmp, cndmask, cvt, min/max, add3, lshl_add, ...)    4.6  f64 add/mul/fma      16  v_rcp_f64
"""
import os, re, subprocess, sys, tempfile
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -S --cuda-device-only'.split()
# (label, first instruction, one past the last) of the blocks on the path, in execution order
CHAIN = [('entry', 0, None), ('.LBB0_3', 0, None), ('.LBB0_5', 0, None), ('.LBB0_9', 0, 58), ('.LBB0_15', 0, None), ('.LBB0_17', 0, None),
         ('.LBB0_113', 0, None), ('.LBB0_115', 0, 4), ('.LBB0_117', 0, None), ('.LBB0_119', 0, 2), ('.LBB0_170', 0, 3), ('.LBB0_196', 0, None),
         ('.LBB0_200', 0, 2), ('.LBB0_202', 0, None), ('.LBB0_203', 0, None), ('.LBB0_204', 0, 19), ('.LBB0_217', 0, None), ('.LBB0_219', 0, None)]
TWO = {'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_mov_b32', 'v_add_f32', 'v_sub_f32', 'v_mul_f32',
       'v_fma_f32', 'v_fmac_f32', 'v_fmamk_f32', 'v_fmaak_f32', 'v_lshrrev_b32', 'v_mov_b64'}


def cost(op):
    o = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if not o.startswith('v_'):
        return 0.0
    if o in TWO:
        return 2.0
    if o.startswith('v_rcp_f64'):
        return 16.0
    if 'f64' in o and not o.startswith(('v_cvt', 'v_frexp', 'v_cmp')):
        return 4.6
    return 4.0


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'warp.s')
        subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [os.path.join(REPO, 'meshflow_amd', 'csrc', 'warp.hip'), '-o', out],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith('_ZN2mf11warp_kernel')][0]
    blocks, cur = {}, ('entry', [])
    for l in lines[start + 1:]:
        m = re.match(r'^(\.LBB0_\d+):', l)
        if m:
            blocks[cur[0]] = cur[1]
            cur = (m.group(1), [])
        elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
            cur[1].append(l.strip())
        if 'codeLenInByte' in l:
            break
    blocks[cur[0]] = cur[1]
    meta = [l.strip('; ').strip() for l in lines if re.search(r'; (NumVgprs|TotalNumSgprs|codeLenInByte|Occupancy|LDSByteSize)', l)][:5]
    # landmarks of the chain in this compile
    assert any(i.startswith('global_load_lds_dwordx4') for i in blocks['.LBB0_5']), 'chain is stale: staging block moved'
    assert sum(i.startswith('v_dot4_u32_u8') for i in blocks['.LBB0_202']) == 24, 'chain is stale: blend block moved'
    assert sum(i.startswith('v_rcp_f64') for i in blocks['.LBB0_15']) == 1, 'chain is stale: coordinate block moved'
    print(__doc__)
    print('kernel resources:', ', '.join(meta))
    tot, cyc, n_valu, n_salu, n_lds, n_vmem, n_smem = Counter(), 0.0, 0, 0, 0, 0, 0
    for label, a, b in CHAIN:
        ins = blocks[label][a:b]
        bv = sum(1 for i in ins if i.startswith('v_'))
        bc = sum(cost(i.split()[0]) for i in ins)
        print(f'\n{label}  [{a}:{b if b is not None else len(blocks[label])}]   {len(ins)} instructions, {bv} VALU, {bc:.0f} VALU issue cycles')
        for i in ins:
            op = i.split()[0]
            c = cost(op)
            print(f'    {c:4.1f}  {i}' if c else f'          {i}')
            if op.startswith('v_'):
                n_valu += 1; cyc += c; tot[re.sub(r'_(e32|e64)$', '', op)] += 1
            elif op.startswith('s_load'):
                n_smem += 1
            elif op.startswith('s_'):
                n_salu += 1
            elif op.startswith('ds_'):
                n_lds += 1
            elif op.startswith(('global_', 'buffer_')):
                n_vmem += 1
    print(f'\nTOTAL per wavefront (256 pixels): {n_valu} VALU ({n_valu / 4:.1f} per 64-pixel slot), {n_salu} SALU/branch/waitcnt, {n_smem} SMEM, {n_lds} LDS, {n_vmem} VMEM')
    print(f'VALU issue cycles by class: {cyc:.0f} per wavefront = {cyc / 4:.0f} per 64-pixel slot')
    print('VALU opcode histogram:', ', '.join(f'{k} {v}' for k, v in tot.most_common()))


if __name__ == '__main__':
    main()
