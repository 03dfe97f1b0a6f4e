"""Stateful fuzz of ONE MeshFlowStabilizer per mesh shape: random sequences of stabilize_resident / stabilize_clip / the drop-in methods / finish()
over clips of random small shapes -- some with a degenerate mesh --, every result checked on its own against the oracles (paths within 1e-9,
frames / rectangle / crop values bit-exact), every degenerate clip reported exactly once and by number.     python tools/stateful_fuzz.py [steps] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
from oracle import clib, meshflow_oracle as mo
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
meshes = [(2, 3), (4, 4), (1, 1)]
stabs = {m: MeshFlowStabilizer(mesh_row_count=m[0], mesh_col_count=m[1], temporal_smoothing_radius=3, optimization_num_iterations=6, device='cuda:0') for m in meshes}
for s in stabs.values():
    s.resident_table_shapes = 2                          # small LRU: evictions happen
shapes = [(24, 32), (40, 64), (33, 45), (48, 80)]
pending = {m: [] for m in meshes}                        # (serial, degenerate?) of resident clips whose verdict is still out
serial = {m: 0 for m in meshes}
bad = 0
t0 = time.time()


def make_clip(m):
    R, C = m
    H, W = shapes[int(rng.integers(len(shapes)))]
    F = int(rng.integers(2, 9))
    frames = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    vel = rng.normal(0, 0.5, size=(F, R + 1, C + 1, 2)); vel[0] = 0
    disp = np.cumsum(vel, axis=0)
    degenerate = rng.random() < 0.15
    if degenerate:                                        # collapse one cell of one frame: three collinear corners
        f = int(rng.integers(F))
        disp[f, 0, 0] = disp[f, 0, 1] + np.array([W / C, 0.0]) + 1000.0      # far away: the smoothed path will not undo it... (checked below)
    hom = np.tile(np.eye(3), (F, 1, 1)); hom[:-1, :2, 2] = rng.normal(0, 1, size=(F - 1, 2))
    return frames, disp, hom, (F, H, W, R, C)


def expect(frames, disp, stab, R, C):
    want, want_crop, wbad = clib.warp_clip(frames, R, C, disp, np.ascontiguousarray(stab), (0, 0, 255))
    rect = (int(want_crop[:, 0].max()), int(want_crop[:, 1].max()), int(want_crop[:, 2].min()), int(want_crop[:, 3].min()))
    return want, rect, wbad


for step in range(steps):
    m = meshes[int(rng.integers(len(meshes)))]
    s = stabs[m]
    op = rng.choice(['resident', 'resident', 'clip', 'dropin', 'finish'])
    if op == 'finish':
        try:
            s.finish()
            if any(d for _, d in pending[m]):
                bad += 1; print(step, 'finish() did not report a degenerate clip', pending[m])
        except ValueError as e:
            if not any(d for _, d in pending[m]):
                bad += 1; print(step, 'finish() raised without cause:', e)
        pending[m].clear()
        continue
    frames, disp, hom, (F, H, W, R, C) = make_clip(m)
    want_stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 3, 6)
    want, rect, wbad = expect(frames, disp, want_stab, R, C)
    if op == 'resident':
        serial[m] += 1
        # round 6: the public default checks at once (the error belongs to THIS call); pipelined callers pass check='deferred'
        deferred = bool(rng.integers(2))
        issued_before = s.resident_serial
        try:
            out, bounds, d_stab = s.stabilize_resident(torch.from_numpy(frames).to(dev), torch.from_numpy(disp).to(dev), hom,
                                                       check='deferred' if deferred else True)
        except ValueError as e:
            if s.resident_serial != issued_before:            # THIS clip was issued and its own verdict came back at once
                if deferred or not wbad or getattr(e, 'clip_serial', None) != s.resident_serial:
                    bad += 1; print(step, 'a synchronous verdict that is not this clip\'s:', e)
                continue
            # nothing was issued: the deferred verdict of an EARLIER clip (its table slot came up again): must have been degenerate
            old = [p for p in pending[m] if p[1]]
            if not old:
                bad += 1; print(step, 'a deferred verdict without a degenerate clip before it')
            pending[m] = [p for p in pending[m] if not p[1] or p is not old[0]]
            serial[m] -= 1
            continue
        if wbad and not deferred:
            bad += 1; print(step, 'a degenerate clip passed the synchronous check')
        pending[m].append((serial[m], bool(wbad) and deferred))
        pending[m] = pending[m][-2:] if not any(d for _, d in pending[m][:-2]) else pending[m]
        stab = d_stab.cpu().numpy()
        if not np.allclose(stab, want_stab, rtol=0, atol=1e-9):
            bad += 1; print(step, 'resident paths differ')
        if not wbad:
            w2, r2, _ = expect(frames, disp, stab, R, C)
            if not (np.array_equal(out.cpu().numpy(), w2) and bounds.tolist() == list(r2)):
                bad += 1; print(step, 'resident frames / rectangle differ', (F, H, W, R, C))
    elif op == 'clip':
        try:
            o, r, st, score, cropped = s.stabilize_clip(list(frames), disp, hom, crop=True)
            if wbad or rect[2] < rect[0] or rect[3] < rect[1]:
                bad += 1; print(step, 'stabilize_clip accepted a degenerate / empty clip')
                continue
            w2, r2, _ = expect(frames, disp, st, R, C)
            if not (np.array_equal(np.stack(o), w2) and tuple(int(v) for v in r) == r2 and np.array_equal(np.stack(cropped), np.stack(mo.crop_frames(list(w2), r2)))):
                bad += 1; print(step, 'stabilize_clip differs', (F, H, W, R, C))
        except ValueError as e:
            if not (wbad or rect[2] < rect[0] or rect[3] < rect[1]):
                bad += 1; print(step, 'stabilize_clip raised without cause:', str(e)[:80])
    else:
        st = s._get_stabilized_vertex_displacements(F, list(frames), 0, disp, hom)
        try:
            o, r = s._get_stabilized_frames_and_crop_boundaries(F, list(frames), disp, st)
            w2, r2, b2 = expect(frames, disp, st, R, C)
            if b2 or not (np.array_equal(np.stack(o), w2) and tuple(int(v) for v in r) == r2):
                bad += 1; print(step, 'drop-in pair differs', (F, H, W, R, C))
        except ValueError:
            if not wbad:
                bad += 1; print(step, 'drop-in raised without cause')
for m, s in stabs.items():
    try:
        s.finish()
    except ValueError:
        pass
print(f'{steps} steps in {time.time() - t0:.1f} s: {bad} bad')
sys.exit(1 if bad else 0)
