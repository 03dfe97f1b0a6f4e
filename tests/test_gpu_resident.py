"""-m gpu: the device-resident clip pipeline -- `mf_warp_clip_u8c3` (csrc/clippipe.hip: cell tables, crop scan and clip rectangle on a
prep stream, the warp in frame-range chunks on the caller's stream, each waiting for its own table only) and
`MeshFlowStabilizer.stabilize_resident` on top of it (the sweep on the prep stream as well, consecutive clips overlapping).
Everything must equal the plain sequence mf_cell_table_f64 -> mf_warp_u8c3 -> mf_crop_reduce, byte for byte, whatever the chunking
and the stream arrangement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _clip(F, H, W, R, C, seed, **kw):
    from meshflow_amd import synthetic
    from oracle import meshflow_oracle as mo
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=seed, kind='noise', **kw)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 3, 10)
    return frames, disp, stab


def _plain(dev, d_fr, d_un, d_st, R, C, border):
    from meshflow_amd import ops
    n, H, W, _ = d_fr.shape
    table = ops.cell_table(d_un, d_st, W, H, R, C)
    out = ops.warp(d_fr, table, border)
    bounds = ops.crop_reduce(table.crop, W, H)
    torch.cuda.synchronize()
    table.check()
    return out, table.crop.clone(), bounds


@pytest.mark.parametrize('F,H,W,R,C', [(13, 72, 100, 3, 5), (16, 96, 128, 8, 8), (3, 48, 64, 2, 2), (40, 136, 256, 4, 4)])
@pytest.mark.parametrize('chunks', [0, 1, 3, 4, 32])           # 0: in order (rectangle behind the warp, or early on a prep stream of the caller's)
@pytest.mark.parametrize('streams', ['internal', 'own', 'one'])
def test_warp_clip_equals_the_three_calls(dev, F, H, W, R, C, chunks, streams):
    from meshflow_amd import ops
    from oracle import clib
    frames, disp, stab = _clip(F, H, W, R, C, seed=F + W, jitter_sigma=0.8)
    d_fr, d_un, d_st = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (frames, disp, stab))
    want, want_crop, want_bounds = _plain(dev, d_fr, d_un, d_st, R, C, (5, 6, 7))
    table = ops.CellTable(F, W, H, R, C, dev)
    prep = {'internal': None, 'own': torch.cuda.Stream(device=dev), 'one': torch.cuda.current_stream(dev)}[streams]
    if streams == 'own':
        prep.wait_stream(torch.cuda.current_stream(dev))          # the uploads above
    for _ in range(2):                                            # second call: the same table and events again
        out, bounds = ops.warp_clip(d_fr, d_un, d_st, table, (5, 6, 7), chunks=chunks, prep_stream=prep)
        torch.cuda.synchronize()
        table.check()
        assert torch.equal(out, want)
        assert torch.equal(table.crop, want_crop)
        assert torch.equal(bounds, want_bounds)
    ref, ref_crop, bad = clib.warp_clip(frames, R, C, disp, stab, (5, 6, 7))
    assert bad == 0
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    np.testing.assert_array_equal(table.crop.cpu().numpy(), ref_crop)


def test_warp_clip_counts_degenerate_cells_and_checks_arguments(dev):
    from meshflow_amd import _lib, ops, synthetic
    F, H, W, R, C = 9, 64, 96, 4, 4
    frames = synthetic.frames_numpy(F, H, W, seed=1)
    z = np.zeros((F, R + 1, C + 1, 2))
    s = z.copy()
    grid_x = np.array([np.ceil((W - 1) * c / C) for c in range(C + 1)])
    s[7, :, :, 0] = -grid_x[None, :]                   # frame 7 (third chunk of four): every vertex onto x = 0
    d_fr, d_un, d_st = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (frames, z, s))
    table = ops.CellTable(F, W, H, R, C, dev)
    ops.warp_clip(d_fr, d_un, d_st, table, chunks=4)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match='degenerate'):
        table.check()
    with pytest.raises(ValueError):
        ops.warp_clip(d_fr[:5], d_un, d_st, table)                                     # frames do not match the table
    assert _lib.lib.mf_warp_clip_u8c3(None, None, None, None, 1, 64, 64, 2, 2, None, None, None, None, None, 4, None, None) == _lib.MF_ERR_INVALID_ARG


@pytest.mark.parametrize('mode', ['in order', 'early rectangle', '4 frame ranges'])
@pytest.mark.parametrize('frame_range', [None, (10, 31)])
def test_stabilize_resident_equals_the_staged_device_path(dev, frame_range, mode):
    """Three clips back to back through the product's resident pipeline (the second clip's sweep and tables run beside the first
    clip's warp; two tables take turns), whole clip and a frame-range shard: frames, rectangle and paths equal the staged methods."""
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    F, H, W, R, C = 48, 136, 256, 4, 6
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=5, optimization_num_iterations=20, device=str(dev))
    s.resident_chunks = 4 if mode == '4 frame ranges' else 0
    s.resident_rectangle = 'early' if mode == 'early rectangle' else 'fused'
    lo, hi = frame_range or (0, F)
    clips = []
    for seed in (1, 2, 3):
        frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=seed, kind='noise', jitter_sigma=0.6)
        clips.append((torch.from_numpy(frames[lo:hi]).to(dev), torch.from_numpy(disp).to(dev), hom))
    want = []
    for d_fr, d_disp, hom in clips:
        d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
        out, crop = s._stabilized_frames_device(d_fr, d_disp[lo:hi], d_stab[lo:hi])
        want.append((out.clone(), ops.crop_reduce(crop, W, H).clone(), d_stab.clone()))
    torch.cuda.synchronize()
    ready = torch.cuda.Event()
    ready.record()
    got = []
    for d_fr, d_disp, hom in clips:                       # issued back to back, nothing synchronises in between
        out, bounds, d_stab = s.stabilize_resident(d_fr, d_disp, hom, frame_range=frame_range, inputs_ready=ready, check=False)
        got.append((out, bounds.clone(), d_stab))
    torch.cuda.synchronize()
    for (o, b, st), (wo, wb, wst) in zip(got, want):
        assert torch.equal(o, wo) and torch.equal(b, wb) and torch.equal(st, wst)
    # and with the default (safe) input ordering, checking the mesh
    out, bounds, d_stab = s.stabilize_resident(*clips[0], frame_range=frame_range)
    torch.cuda.synchronize()
    assert torch.equal(out, want[0][0]) and torch.equal(bounds, want[0][1])


def test_stabilize_resident_full_cfg2(dev):
    """BASELINE config 2 at full size through the resident pipeline, twice back to back: equal to the staged device path."""
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    F, H, W, R, C = 300, 1080, 1920, 16, 16
    s = MeshFlowStabilizer(device=str(dev))
    disp, hom = synthetic.motion(F, R, C, seed=0)
    d_disp = torch.from_numpy(disp).to(dev)
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern')
    d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
    want, crop = s._stabilized_frames_device(d_frames, d_disp, d_stab)
    want_bounds = ops.crop_reduce(crop, W, H).clone()
    torch.cuda.synchronize()
    out = torch.empty_like(d_frames)
    for chunks in (0, 0, 4):
        s.resident_chunks = chunks
        out.zero_()
        got, bounds, stab2 = s.stabilize_resident(d_frames, d_disp, hom, out=out)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(bounds, want_bounds) and torch.equal(stab2, d_stab)


def _degenerate(disp, stab, frame, W, C):
    """Frame `frame` of the paths with every vertex moved onto x = 0: no cell of it has a homography."""
    bad = stab.copy()
    grid_x = np.array([np.ceil((W - 1) * c / C) for c in range(C + 1)])
    bad[frame, :, :, 0] = disp[frame, :, :, 0] - grid_x[None, :]
    return bad


def test_resident_bounds_are_the_callers_own_across_six_calls(dev):
    """VERDICT r4 item 6 / ADVICE: the rectangle `stabilize_resident` returns is 16 bytes of the clip's own (the kernels fold it
    together there), not a view into one of two rotating tables: three clips' bounds, held across six back-to-back calls on the same
    table slots (alternating shapes included), are intact at the end."""
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    H, W, R, C = 136, 256, 4, 6
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=5, optimization_num_iterations=20, device=str(dev))
    clips, want = [], []
    for seed, F in ((1, 48), (2, 48), (3, 31), (4, 48), (5, 31), (6, 48)):
        frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=seed, kind='noise', jitter_sigma=0.6)
        clips.append((torch.from_numpy(frames).to(dev), torch.from_numpy(disp).to(dev), hom))
    for d_fr, d_disp, hom in clips:
        d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
        out, crop = s._stabilized_frames_device(d_fr, d_disp, d_stab)
        want.append((out.clone(), ops.crop_reduce(crop, W, H).tolist()))
    torch.cuda.synchronize()
    held = []
    for d_fr, d_disp, hom in clips:                       # six calls, nothing synchronises, nothing is cloned
        out, bounds, _ = s.stabilize_resident(d_fr, d_disp, hom, check='deferred')
        held.append((out, bounds))
    s.finish()
    torch.cuda.synchronize()
    assert len({b.data_ptr() for _, b in held}) == 6
    for (out, bounds), (wo, wb) in zip(held, want):
        assert torch.equal(out, wo) and bounds.tolist() == wb
    # one geometry, clips of 48 and 31 frames: ONE pair of tables (grow-only in the clip length), not a pair per length
    assert len(s._resident['tables']) == 1 and all(slot['table'].capacity == 48 for slot in next(iter(s._resident['tables'].values())))


@pytest.mark.parametrize('mode', ['in order', '4 frame ranges'])
def test_resident_degenerate_clip_is_reported_once_and_later_clips_pass(dev, mode):
    """the degenerate-cell counter is per clip.  A clip with a degenerate mesh raises -- deferred: when its table
    slot comes up again two clips later, or at finish() -- naming the clip; the valid clips before and after it (same slot included)
    give the right frames and raise nothing; check=True raises at once."""
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    F, H, W, R, C = 24, 96, 128, 4, 4
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=3, optimization_num_iterations=10, device=str(dev))
    s.resident_chunks = 4 if mode == '4 frame ranges' else 0
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=8, kind='noise', jitter_sigma=0.5)
    d_fr, d_good = torch.from_numpy(frames).to(dev), torch.from_numpy(disp).to(dev)
    # a displacement tensor whose SMOOTHED paths are degenerate at frame 7 cannot be built from outside; drive the two stages instead
    d_stab = s._stabilized_vertex_displacements_device(d_good, W, H, 0, hom)
    stab = d_stab.cpu().numpy()
    d_bad = torch.from_numpy(_degenerate(disp, stab, 7, W, C)).to(dev)
    want, crop = s._stabilized_frames_device(d_fr, d_good, d_stab)
    want = want.clone()

    def clip(d_st, **kw):
        out, bounds, table = s._resident_warp(d_fr, d_good, d_st, **kw)
        return out
    outs = [clip(d_stab), clip(d_bad), clip(d_stab)]        # serials 1, 2 (degenerate), 3: slot A, slot B, slot A (settles clip 1: fine)
    with pytest.raises(ValueError, match=r'degenerate mesh cell\(s\) in resident clip #2'):
        clip(d_stab)                                        # slot B comes up again: clip 2's verdict, before anything is issued
    outs.append(clip(d_stab))                               # the same call again goes through: the counter is per clip
    outs.append(clip(d_stab))
    s.finish()                                              # nothing pending is bad
    torch.cuda.synchronize()
    for i in (0, 2, 3, 4):
        assert torch.equal(outs[i], want), i
    clip(d_bad)
    with pytest.raises(ValueError, match='degenerate'):
        s.finish()
    s.finish()                                              # reported once
    with pytest.raises(ValueError, match='degenerate'):
        clip(d_bad, check=True)                             # the blocking form
    assert torch.equal(clip(d_stab, check=True), want)
    clip(d_bad, check='never')
    s.finish()                                              # never looked at
    # the public default is the synchronous check; the error is a type of its own that names the clip; and a
    # deferred verdict is raised BEFORE anything of the next clip -- its sweep included -- is queued
    import inspect
    from meshflow_amd import DegenerateMeshError
    assert inspect.signature(s.stabilize_resident).parameters['check'].default is True
    with pytest.raises(DegenerateMeshError) as err:
        clip(d_bad, check=True)
    assert err.value.clip_serial == s.resident_serial and err.value.cells >= 1 and isinstance(err.value, ValueError)
    clip(d_stab)
    clip(d_bad)                                             # deferred, serial k: its slot comes up again two clips on
    bad_serial = s.resident_serial
    clip(d_stab)
    swept, serial = s._resident.get('swept'), s.resident_serial
    with pytest.raises(DegenerateMeshError) as err:
        s.stabilize_resident(d_fr, d_good, hom, check='deferred')
    assert err.value.clip_serial == bad_serial
    assert s._resident.get('swept') is swept and s.resident_serial == serial       # no sweep queued, no clip issued
    out, bounds, _ = s.stabilize_resident(d_fr, d_good, hom)                         # the same call again goes through (default: checked at once)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    s.finish()


def test_resident_table_cache_is_bounded(dev):
    """a stream of clips of many geometries keeps at most `resident_table_shapes` table pairs."""
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    s = MeshFlowStabilizer(mesh_row_count=2, mesh_col_count=2, temporal_smoothing_radius=2, optimization_num_iterations=4, device=str(dev))
    s.resident_table_shapes = 3
    for k in range(7):
        H, W = 48 + 8 * k, 64 + 16 * k
        frames, disp, hom = synthetic.clip(5, H, W, 2, 2, seed=k)
        out, bounds, _ = s.stabilize_resident(torch.from_numpy(frames).to(dev), torch.from_numpy(disp).to(dev), hom)
        assert out.shape == (5, H, W, 3)
    s.finish()
    assert len(s._resident['tables']) == 3


def test_frame_range_shards_down_to_empty_ones():
    """`stabilize_resident(frame_range=(lo, hi))` on every shard of an uneven 8-way split of a 12-frame clip (ceil(12 / 8) = 2 frames per
    rank: six ranks with two frames, two EMPTY ones) and on one-frame shards: the shards' frames are the slices of the whole clip's, the
    paths the same everywhere, and max / min over the shards' rectangles (an empty shard contributes the neutral element,
    mfs.py:992-995) is the whole clip's rectangle -- what the 16-byte all-reduce of an N-GPU job computes."""
    import numpy as np
    import torch
    from meshflow_amd import host, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    dev = torch.device('cuda:0')
    F, H, W, R, C = 12, 120, 160, 4, 4
    disp, hom = synthetic.motion(F, R, C, seed=1)
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=1)
    d_disp = torch.from_numpy(disp).to(dev)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=3, optimization_num_iterations=5, device='cuda:0')
    full, b_full, stab_full = s.stabilize_resident(d_frames, d_disp, hom)
    s.finish()
    for rectangle in ('late', 'early'):
        s.resident_rectangle = rectangle
        rects, empties = [], 0
        ranges = [host.shard_range(F, 8, g) for g in range(8)] + [(0, 1), (11, 12)]
        for lo, hi in ranges:
            frames, bounds, stab = s.stabilize_resident(d_frames[lo:hi], d_disp, hom, frame_range=(lo, hi))
            s.finish()
            assert tuple(frames.shape) == (hi - lo, H, W, 3) and torch.equal(frames, full[lo:hi]) and torch.equal(stab, stab_full)
            if hi == lo:
                empties += 1
                assert bounds.tolist() == [0, 0, W - 1, H - 1]
            rects.append(bounds.tolist())
        assert empties == 2
        r = np.array(rects[:8])
        assert [int(r[:, 0].max()), int(r[:, 1].max()), int(r[:, 2].min()), int(r[:, 3].min())] == b_full.tolist()


def test_resident_pipelines_of_several_threads_on_streams_of_their_own():
    """Four host threads, each with its own stabilizer and torch stream, 100 resident clips each without waiting for one another (in
    order and in two frame ranges, whose events the library shares per device): every clip's frames, rectangle and paths equal the
    single-threaded result."""
    import threading
    import torch
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    dev = torch.device('cuda:0')
    jobs = []
    for i, (F, H, W, R, C) in enumerate(((40, 360, 640, 16, 16), (30, 250, 333, 7, 5), (24, 720, 1280, 8, 8), (50, 96, 128, 4, 4))):
        disp, hom = synthetic.motion(F, R, C, seed=i)
        d_frames = synthetic.frames_torch(F, H, W, dev, seed=i)
        d_disp = torch.from_numpy(disp).to(dev)
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=5, optimization_num_iterations=20, device='cuda:0')
        out, b, st = s.stabilize_resident(d_frames, d_disp, hom)
        s.finish()
        torch.cuda.synchronize()
        jobs.append((d_frames, d_disp, hom, (R, C), out.clone(), b.tolist(), st.clone()))
    errors = []

    def worker(i, chunks):
        d_frames, d_disp, hom, (R, C), ref, ref_bounds, ref_stab = jobs[i]
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=5, optimization_num_iterations=20, device='cuda:0')
        s.resident_chunks = chunks
        stream = torch.cuda.Stream(device=dev)
        try:
            with torch.cuda.stream(stream):
                queue = []
                for _ in range(100):
                    queue.append(s.stabilize_resident(d_frames, d_disp, hom, check='deferred'))
                    if len(queue) >= 4:
                        out, b, st = queue.pop(0)
                        stream.synchronize()
                        if not (torch.equal(out, ref) and b.tolist() == ref_bounds and torch.equal(st, ref_stab)):
                            errors.append(('mismatch', i, chunks))
                s.finish()
                stream.synchronize()
                for out, b, st in queue:
                    if not (torch.equal(out, ref) and b.tolist() == ref_bounds and torch.equal(st, ref_stab)):
                        errors.append(('mismatch at the end', i, chunks))
        except Exception as e:                                       # noqa: BLE001 -- reported below
            errors.append((type(e).__name__, str(e)[:100]))

    for chunks in (0, 2):
        threads = [threading.Thread(target=worker, args=(i, chunks)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:4]
