"""Socket power and shader clock while ONE kernel of the path runs back to back (rocm-smi sampled from the parent):

    python3 tools/kernel_power.py jacobi cfg3 | jacobi cfg2 | warp cfg2 | warp cfg3 | plan cfg2      [seconds]
"""
import os, re, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))


def child(kind, wl, seconds):
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    from meshflow_amd import ops, synthetic, host
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    H, W = 1080, 1920
    F, R, C, om, it = {'cfg2': (300, 16, 16, 10, 100), 'cfg3': (600, 32, 32, 30, 200)}[wl]
    dev = torch.device('cuda:0')
    disp, hom = synthetic.motion(F, R, C, seed=0)
    d_disp = torch.from_numpy(disp).to(dev)
    taps, lam, inv_on = host.jacobi_band_coefficients(F, W, H, 0, hom, om)
    b = d_disp.reshape(F, -1)
    tt = [torch.from_numpy(a).to(dev) for a in (taps, lam, inv_on)]
    out = torch.empty_like(b)
    if kind == 'jacobi':
        fn = lambda: ops.jacobi(b, *tt, om, it, out=out)
    else:
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=om, optimization_num_iterations=it, device='cuda:0')
        d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
        table = ops.cell_table(d_disp, d_stab, W, H, R, C)
        if kind == 'plan':
            fn = lambda: ops.cell_table(d_disp, d_stab, W, H, R, C, table=table)
        else:
            frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern')
            dst = torch.empty_like(frames)
            fn = lambda: ops.warp(frames, table, (0, 0, 255), out=dst)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20): fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    print(f'{kind} {wl}: {n} launches, {e0.elapsed_time(e1) / n * 1e3:.1f} us each', flush=True)


def sample():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    p = re.search(r'Power \(W\): ([0-9.]+)', out)
    f = re.search(r'sclk clock level.*\((\d+)Mhz\)', out)
    return (float(p.group(1)), int(f.group(1))) if p and f else None


if __name__ == '__main__':
    if sys.argv[1] == 'child':
        child(sys.argv[2], sys.argv[3], float(sys.argv[4]))
        sys.exit(0)
    kind, wl = sys.argv[1], sys.argv[2]
    seconds = sys.argv[3] if len(sys.argv) > 3 else '4'
    proc = subprocess.Popen([sys.executable, __file__, 'child', kind, wl, seconds], stdout=subprocess.PIPE, text=True)
    got = []
    t0 = time.time()
    while proc.poll() is None:
        s = sample()
        if s and s[0] > 600:                 # (only while the kernel loop runs: start-up draws far less)
            got.append(s)
        time.sleep(0.1)
    print(proc.stdout.read().strip())
    if got:
        got = got[len(got) // 4:]
        print(f'  {sum(g[0] for g in got) / len(got):5.0f} W  {sum(g[1] for g in got) / len(got):5.0f} MHz  [{len(got)} samples above 600 W]')
    else:
        print('  no samples above 600 W')
