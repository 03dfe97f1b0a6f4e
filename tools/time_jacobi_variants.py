"""Jacobi kernel variants: wavefronts per series at cfg2 size (MF_JACOBI_SPLIT), and the generic kernel at radii without a specialisation."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from meshflow_amd import ops, synthetic, host
    dev = torch.device('cuda:0')
    def t(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    for F, R, C, om, it in [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]]:
        disp, hom = synthetic.motion(F, R, C, seed=0)
        taps, lam, inv_on = host.jacobi_band_coefficients(F, 1920, 1080, 0, hom, om)
        b = torch.from_numpy(disp.reshape(F, -1)).to(dev)
        tt = [torch.from_numpy(a).to(dev) for a in (taps, lam, inv_on)]
        out = torch.empty_like(b)
        ms = t(lambda: ops.jacobi(b, *tt, om, it, out=out))
        flops = it * F * b.shape[1] * (2 * (2 * om + 1) + 3)
        print(f'  F={F} mesh={R}x{C} omega={om} iters={it}: {ms * 1e3:.1f} us  {flops / ms / 1e9:.2f} TFLOP/s', flush=True)
    sys.exit(0)
for split in ('1', '2', '4'):
    print('MF_JACOBI_SPLIT =', split, flush=True)
    subprocess.run([sys.executable, __file__, 'child', '300,16,16,10,100', '240,16,16,10,100', '300,8,8,10,100'], env=dict(os.environ, MF_JACOBI_SPLIT=split))
print('default selection', flush=True)
subprocess.run([sys.executable, __file__, 'child', '300,16,16,10,100', '300,32,32,10,100', '2400,16,16,10,100', '600,32,32,30,200'])
print('other radii: specialised ahead of time up to 32, the run-time-radius kernel beyond', flush=True)
subprocess.run([sys.executable, __file__, 'child', '300,16,16,7,100', '300,16,16,12,100', '300,16,16,25,100', '300,16,16,40,100', '600,32,32,25,200', '2400,16,16,12,100', '300,16,16,64,100'])
print('specialised radii through the run-time-radius kernel (MF_JACOBI_RUNTIME=1)', flush=True)
subprocess.run([sys.executable, __file__, 'child', '300,16,16,10,100', '300,16,16,5,100', '300,16,16,20,100', '600,32,32,30,200', '2400,16,16,10,100'], env=dict(os.environ, MF_JACOBI_RUNTIME='1'))
print('... and through their own kernels', flush=True)
subprocess.run([sys.executable, __file__, 'child', '300,16,16,10,100', '300,16,16,5,100', '300,16,16,20,100', '600,32,32,30,200', '2400,16,16,10,100'])
