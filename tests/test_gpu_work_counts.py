"""Round 4: what a launch EXECUTES is counted by the launch itself (SRT_RENDER_COUNT_WORK / srt_get_work_counts) and the launch
SHAPE srt_render chooses — tile height, sample chunks — is a function of the request, the grid and the band's recorded loop
counts, never of a clock.

* counting does not change a bit of the frame (analytic, box, mesh scenes; full tiles, small tiles, sample chunks, block grids);
* the counts are identical in every run and obey the identities the kernel's structure implies (every closest_hit call runs all
  uniform spheres and boxes for 64 lanes, ...), and the ray count they imply brackets the oracle's;
* two fresh contexts given the same calls report the same sequence of launch shapes, first launch (static rule) and second launch
  (work record) included; a launch with an in-flight record WAITS for it, so the sequence does not depend on host timing either;
* a scene whose image lives in HBM cannot count: valid = 0, nothing breaks."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _tracer(srt, oracle, name, w, h, mesh=0):
    objs = oracle.load_scene_json_py(scene_path(name))
    meshes = []
    if mesh:
        big = objs[64]
        objs[64] = dict(type=oracle.OBJ_MESH, position=big["position"], mesh=0, base=big["base"], emissive=big["emissive"],
                        smoothness=big["smoothness"], specular_amount=big["specular_amount"], specular=big["specular"])
        meshes = [oracle.uv_sphere(1.0, mesh, mesh)]
    oarr, n = oracle.make_objects(objs)
    marr, mn, keep = oracle.make_meshes(meshes)
    pt = srt.PathTracer(w, h)
    pt.set_meshes(C.cast(marr, C.POINTER(srt.Mesh)), mn)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    return pt, oarr, n, keep


CASES = [  # name, mesh, w, h, rows, spp, bounces, steps
    ("Scene1", 0, 640, 360, None, 8, 8, 1),            # full tiles
    ("Scene_indirect", 0, 320, 180, None, 4, 8, 1),    # boxes
    ("Scene3", 0, 640, 64, None, 24, 6, 1),            # few blocks, >= 16 spp: small tiles (multi-sample hand-out)
    ("Scene1", 0, 1920, 96, (900, 996), 128, 8, 1),    # a narrow band at 128 spp: sample chunks + fold kernel
    ("Scene1", 48, 480, 270, None, 40, 6, 1),          # a mesh, >= 32 spp: chunked mesh launch
    ("Scene2", 0, 400, 300, None, 1, 5, 4),            # progressive blocks: block grid
]


@pytest.mark.parametrize("name,mesh,w,h,rows,spp,bounces,steps", CASES)
def test_counting_changes_nothing_and_counts_are_consistent(srt, oracle, name, mesh, w, h, rows, spp, bounces, steps):
    if rows is not None:
        h = 1080
    pt, oarr, n, keep = _tracer(srt, oracle, name, w, h, mesh)
    rows = rows or (0, h)
    kw = dict(spp=spp, bounces=bounces, seed=5, rows=rows, steps=steps, stripe_width=(w + 15) // 16 + 1 if steps > 1 else 0)
    pt.render(count_rays=True, **kw)
    fb0, acc0, st0 = pt.framebuffer(), pt.accumulator(), pt.stats()
    with pytest.raises(srt.SrtError) as e:  # the last render did not count
        pt.work_counts()
    assert e.value.code == srt.capi.ERR_STATE
    pt.render(count_rays=True, count_work=True, **kw)
    fb1, acc1, st1 = pt.framebuffer(), pt.accumulator(), pt.stats()
    c1 = pt.work_counts().as_dict()
    assert np.array_equal(fb0, fb1) and np.array_equal(acc0.view(np.uint32), acc1.view(np.uint32)) and st0.rays == st1.rays
    # a second context: the same counts, the same shapes (first launch AND second launch)
    pt2, _, _, keep2 = _tracer(srt, oracle, name, w, h, mesh)
    pt2.render(count_rays=True, **kw)
    s0 = pt2.stats()
    pt2.render(count_rays=True, count_work=True, **kw)
    s1 = pt2.stats()
    c2 = pt2.work_counts().as_dict()
    assert c1 == c2
    shape = lambda s: (s.tile_rows, s.sample_chunks, s.chunk_samples, s.shape_source)
    assert shape(s0) == shape(st0) and shape(s1) == shape(st1)
    assert c1["valid"] == 1 and c1["waves"] > 0
    # identities of the kernel's structure (srt_pathtrace.h, srt_work_counts)
    assert c1["closest_hit_calls"] == c1["pool_steps"] + c1["waves"]
    objs = oracle.load_scene_json_py(scene_path(name))
    n_box = sum(1 for o in objs if o["type"] == oracle.OBJ_BOX)
    assert c1["box_tests"] == c1["closest_hit_calls"] * n_box * 64
    assert c1["uniform_sphere_tests"] % (c1["closest_hit_calls"] * 64) == 0
    # K = 4: an item is tested against one group of four, by one, two or four lanes (a round runs 4, 2 or 1 candidate tests in all 64 lanes)
    assert c1["cluster_sphere_tests"] % 64 == 0 and c1["cluster_items"] * 4 <= c1["cluster_sphere_tests"]
    if mesh:
        assert c1["bvh_child_tests"] > 0 and c1["triangle_tests"] > 0 and c1["mesh_phases"] > 0 and c1["bvh_node_rounds"] > 0
    else:
        assert c1["bvh_child_tests"] == c1["triangle_tests"] == c1["mesh_phases"] == 0
    if steps == 1:
        # every traced ray occupies one lane of one pool step: bounce rays <= 64 x steps (the pool's lane efficiency <= 1)
        px = w * (rows[1] - rows[0])
        pt.render(spp=1, bounces=1, seed=5, rows=rows, count_rays=True)
        traced_px = pt.stats().rays - px  # pixels whose primary ray hits something
        bounce_rays = st1.rays - px * spp if traced_px else 0  # (rays = primary per sample + bounce rays)
        assert 0 <= bounce_rays <= 64 * c1["pool_steps"]
        if traced_px:
            assert bounce_rays / (64.0 * c1["pool_steps"]) > 0.3  # ... and the pool is not idling
    pt.close()
    pt2.close()


def test_second_launch_takes_the_recorded_shape_in_every_run(srt, oracle):
    """A 135-row band of 1080p at 512 spp (config 3's launch): the first launch has the static rule's shape, every later one
    the shape the band's work record asks for — and both are the same in every context, however the host paces its calls
    (no wait between the launches here: srt_render itself waits for the record of the launch before)."""
    seqs = []
    for pace in (False, True, False):
        pt, _, _, keep = _tracer(srt, oracle, "Scene1", 1920, 1080)
        seq = []
        for i in range(3):
            pt.render(spp=512, bounces=8, seed=0, rows=(945, 1080))
            if pace:
                pt.wait()
            if i == 2 or pace:
                st = pt.stats()
                seq.append((st.tile_rows, st.sample_chunks, st.chunk_samples, st.shape_source))
        seqs.append(seq)
        fb = pt.framebuffer(rows=(945, 1080))
        pt.close()
        if len(seqs) > 1:
            assert np.array_equal(fb, fb_first)
        fb_first = fb
    assert seqs[1][0][3] == 0 and seqs[1][1][3] == 1 and seqs[1][1] == seqs[1][2]  # static rule, then the record, then the same again
    assert seqs[0][-1] == seqs[1][-1] == seqs[2][-1]                               # unpaced == paced, run after run
    assert seqs[1][1][1] > 1 and seqs[1][1][2] >= 24                               # chunked, chunks of at least 24 samples


def test_scene_in_hbm_cannot_count(srt, oracle):
    objs = [dict(type=oracle.OBJ_SPHERE, position=((i % 60) * 0.2 - 6, (i // 60) * 0.2 - 3, 12), radius=0.08, base=(.5, .6, .7)) for i in range(3400)]
    oarr, n = oracle.make_objects(objs)
    pt = srt.PathTracer(320, 180)
    pt.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    pt.set_camera(srt.default_camera())
    pt.render(spp=2, bounces=3, seed=1)
    fb = pt.framebuffer()
    pt.render(spp=2, bounces=3, seed=1, count_work=True)
    c = pt.work_counts().as_dict()
    assert c["valid"] == 0 and c["pool_steps"] == 0 and np.array_equal(fb, pt.framebuffer())
    pt.close()


def test_untimed_render_is_the_same_render(srt, oracle):
    """SRT_RENDER_NO_TIMING leaves the two timing events out of the stream: the same frame, kernel_ms reads 0, everything else of
    srt_stats (rays, path-samples, the launch shape) is reported as usual."""
    pt, _, _, keep = _tracer(srt, oracle, "Scene1", 640, 360)
    pt.render(spp=8, bounces=6, seed=2, count_rays=True)
    fb, st = pt.framebuffer(), pt.stats()
    assert st.kernel_ms > 0
    pt.render(spp=8, bounces=6, seed=2, count_rays=True, timing=False)
    st2 = pt.stats()
    assert st2.kernel_ms == 0 and st2.rays == st.rays and st2.path_samples == st.path_samples and st2.tile_rows == st.tile_rows
    assert np.array_equal(fb, pt.framebuffer())
    pt.render(spp=8, bounces=6, seed=2)  # ... and the next timed render is timed again
    assert pt.stats().kernel_ms > 0
    pt.close()


def test_asynchronous_read_back_equals_the_synchronous_one(srt, oracle):
    """srt_read_framebuffer_async on a second stream, behind the render: the same rows as srt_read_framebuffer; with two bound
    framebuffers in turn the copy of frame k and the render of frame k + 1 do not disturb each other."""
    import torch

    w, h = 640, 360
    pt, _, _, keep = _tracer(srt, oracle, "Scene1", w, h)
    dev = torch.device("cuda", 0)
    frames = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(2)]
    hosts = [torch.zeros((h, w), dtype=torch.int32).pin_memory() for _ in range(2)]
    copy_stream = torch.cuda.Stream(dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    stream = torch.cuda.Stream(dev)
    pt.set_stream(stream.cuda_stream)
    for k in range(4):  # frame k (seed k) renders into buffer k & 1 while frame k - 1 is still travelling out of the other one
        i = k & 1
        if k >= 2:
            stream.wait_event(copied[i])  # the copy of frame k - 2 has left this buffer
        pt.bind_output(d_framebuffer=frames[i].data_ptr())
        pt.render(spp=3, bounces=5, seed=k)
        pt.read_framebuffer_async(hosts[i].data_ptr(), copy_stream=copy_stream.cuda_stream)
        copied[i].record(copy_stream)
    torch.cuda.synchronize(dev)
    objs = oracle.load_scene_json_py(scene_path("Scene1"))
    oarr, n = oracle.make_objects(objs)
    ref = srt.PathTracer(w, h)
    ref.set_scene(C.cast(oarr, C.POINTER(srt.Object)), n)
    ref.set_camera(srt.default_camera())
    for k in (2, 3):  # the last frame of each buffer
        ref.render(spp=3, bounces=5, seed=k)
        assert np.array_equal(hosts[k & 1].numpy().view(np.uint32), ref.framebuffer()), k
    ref.close()
    # a band, on the launch stream itself (copy_stream = 0)
    part = torch.zeros((100, w), dtype=torch.int32).pin_memory()
    pt.read_framebuffer_async(part.data_ptr(), rows=(50, 150))
    pt.wait()
    assert np.array_equal(part.numpy().view(np.uint32), pt.framebuffer(rows=(50, 150)))
    pt.close()
