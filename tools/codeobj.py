"""Device code objects of a HIP shared library: extraction, kernel resource usage, disassembly.

    python tools/codeobj.py meshflow_amd/libmeshflow_hip.so              resource usage of every kernel
    python tools/codeobj.py meshflow_amd/libmeshflow_hip.so warp_kernel  + disassembly of the kernels whose name contains the substring

The .so carries one clang offload bundle per object file in its .hip_fatbin section; each bundle holds a gfx950 ELF whose
NT_AMDGPU_METADATA note lists every kernel with its register counts, scratch and LDS sizes.  Used by tests/test_isa_guard.py."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(so_path, arch='gfx950'):
    """The device ELFs (bytes) for `arch` inside the library's .hip_fatbin section, in link order."""
    with tempfile.TemporaryDirectory() as tmp:
        fb = os.path.join(tmp, 'fatbin')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fb}', so_path, os.path.join(tmp, 'copy')], check=True)
        blob = open(fb, 'rb').read()
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from('<Q', blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if triple.startswith('hip') and triple.endswith(arch) and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + len(MAGIC))
    return out


def kernel_metadata(elf_bytes):
    """{kernel name: {field: value}} from the code object's AMDGPU metadata note (llvm-readelf --notes prints it as YAML)."""
    import yaml
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(elf_bytes); f.flush()
        txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], check=True, capture_output=True, text=True).stdout
    start = txt.index('amdhsa.kernels:')
    end = txt.find('\n...', start)
    doc = yaml.safe_load(txt[start:end if end > 0 else None])
    return {k['.name']: {key.lstrip('.'): val for key, val in k.items() if key != '.args'} for k in doc['amdhsa.kernels']}


def disassemble(elf_bytes):
    """{symbol: [instruction lines]} of one code object (llvm-objdump -d)."""
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(elf_bytes); f.flush()
        txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', f.name], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for line in txt.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
        elif cur is not None and line.startswith('\t'):
            cur.append(re.sub(r'\s*//.*$', '', line.strip()))
    return funcs


def all_kernels(so_path):
    res = {}
    for co in code_objects(so_path):
        for name, md in kernel_metadata(co).items():
            res[name] = md
    return res


def main():
    so = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else None
    for co in code_objects(so):
        md = kernel_metadata(co)
        dis = disassemble(co) if pat else {}
        for name, m in sorted(md.items()):
            print(f"{name[:110]:110s} vgpr {m.get('vgpr_count'):3d} agpr {m.get('agpr_count', 0):3d} sgpr {m.get('sgpr_count'):3d} scratch {m.get('private_segment_fixed_size')} "
                  f"lds {m.get('group_segment_fixed_size')} spills {m.get('sgpr_spill_count', 0)}/{m.get('vgpr_spill_count', 0)}")
            if pat and pat in name:
                sym = m['symbol'][:-3] if m['symbol'].endswith('.kd') else m['symbol']
                for ins in dis.get(sym, []):
                    print('    ' + ins)


if __name__ == '__main__':
    main()
