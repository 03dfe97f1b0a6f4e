"""Socket power and shader clock while ONE instruction class runs back to back on the whole chip (tools/ubench_issue power OP SECONDS):
what the instruction classes of the warp kernel cost in energy, not only in issue cycles.  On the GPU box:

    make -C tools ubench_issue && python3 tools/ubench_power.py [seconds]
"""
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = ['MOV_B32', 'AND_B32', 'ADD_U32', 'FMA_F32', 'MAD_U24', 'DOT2_U16', 'PK_MAD_U16', 'PERM_B32', 'LSHL_ADD', 'MAX3_U32', 'CVT_I32_F32',
       'CVT_F32_F64', 'ADD_F64', 'MUL_F64', 'FMA_F64', 'RCP_F64', 'DS_READ_B32']


def sample():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    p = re.search(r'Power \(W\): ([0-9.]+)', out)
    f = re.search(r'sclk clock level.*\((\d+)Mhz\)', out)
    return (float(p.group(1)), int(f.group(1))) if p and f else None


def main():
    seconds = sys.argv[1] if len(sys.argv) > 1 else '2.5'
    for op in OPS:
        proc = subprocess.Popen([os.path.join(HERE, 'ubench_issue'), 'power', op, seconds], stdout=subprocess.PIPE, text=True)
        time.sleep(1.0)
        got = []
        while proc.poll() is None:
            s = sample()
            if s and s[1] > 400:
                got.append(s)
            time.sleep(0.15)
        line = proc.stdout.read().strip()
        if got:
            w = sum(g[0] for g in got) / len(got)
            f = sum(g[1] for g in got) / len(got)
            m = re.search(r'\(([0-9.]+) ns per instruction', line)
            ns = float(m.group(1)) if m else float('nan')
            # energy of one wave64 instruction on one SIMD: socket power above idle / (1024 SIMDs * instructions per second per SIMD)
            print(f'{line}   | {w:5.0f} W  {f:5.0f} MHz  {ns * f * 1e-3:5.2f} cycles/instr  {(w - 250.0) * ns / 1024:6.2f} nJ per wave64 instruction (above 250 W idle)  [{len(got)} samples]')
        else:
            print(line, '  | no samples')


if __name__ == '__main__':
    main()
