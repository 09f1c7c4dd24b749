"""Edge cases of the boundary on the MI355X: the options the reference's driver exposes but its shipped configs leave off
(lindisp, white background, no background prior, no fine pass), empty and maximum sizes, and the error behaviour."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from test_gpu_parity import T, close, dev

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def ops():
    return pkg("ops")


@pytest.fixture(scope="module")
def setup(ops, flat_weights):
    fw = flat_weights(0, 8.0, 30.0)
    flat = T(fw)
    g = load_golden("cond")
    frame = ops.fold_conditioning(flat, T(g["audio"]), T(g["pose"]))
    return fw, ops.pack_weights(flat), frame, oracle.audionet(fw, g["audio"]), oracle.pose_encoding(g["pose"])


def make_rays(N, seed, near=0.48, far=1.08):
    rng = np.random.default_rng(seed)
    rays = np.zeros((N, 8), np.float32)
    rays[:, 0:3] = rng.normal(0, 0.02, (N, 3)) + np.array([0, 0, 0.8])
    rays[:, 3:6] = rng.normal(0, 0.15, (N, 3)) + np.array([0, 0, -1.0])
    rays[:, 6], rays[:, 7] = near, far
    return rays, rng


@pytest.mark.parametrize("lindisp", [False, True])
@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("S", [1, 2, 64, 77])
def test_stratified_depths(ops, lindisp, perturb, S):
    """train_utils.py:93-113 incl. lindisp (1/(1/near (1-t) + 1/far t)) and S = 1, 2."""
    rays, rng = make_rays(33, S)
    rays[:, 6] = rng.uniform(0.2, 0.5, 33)
    rays[:, 7] = rng.uniform(0.8, 1.2, 33)
    t_rand = rng.uniform(0, 1, (33, S)).astype(np.float32) if perturb else None
    z = ops.stratified_depths(T(rays), S, lindisp, None if t_rand is None else T(t_rand))
    ref = oracle.stratified_depths(rays[:, 6], rays[:, 7], S, lindisp, t_rand)
    close(z, ref, 2e-7, 0.0, "depths")


@pytest.mark.parametrize("lindisp,white,use_bg,nf", [(True, False, True, 64), (False, True, False, 64), (True, True, False, 32), (False, False, True, 0)])
def test_render_rays_options(ops, setup, lindisp, white, use_bg, nf):
    """predict_and_render_radiance with the switches the shipped configs leave off, and without a fine pass (num_fine = 0:
    the reference then returns None for the fine outputs, train_utils.py:203-206)."""
    fw, packed, frame, drv, p36 = setup
    N, nc = 40, 64
    rays, rng = make_rays(N, 5 + nf)
    bg = np.concatenate([rng.uniform(0, 1, (N, 3)), np.ones((N, 1)), np.zeros((N, 11))], 1).astype(np.float32) if use_bg else None
    t_rand = rng.uniform(0, 1, (N, nc)).astype(np.float32)
    u = rng.uniform(0, 1, (N, max(nf, 1))).astype(np.float32)
    outs = ops.render_rays(packed, frame, T(rays), nc, nf, lindisp=lindisp, white_background=white, bg=None if bg is None else T(bg),
                           t_rand=T(t_rand), u=T(u) if nf else None)
    if nf == 0:
        assert outs[3] is None and outs[4] is None and outs[5] is None
        ref = oracle.render_rays(fw, rays, nc, 0, drv, p36, bg=bg, t_rand=t_rand, lindisp=lindisp, white_background=white)
        names = ["rgb_c", "disp_c", "acc_c", None, None, None, "w_bg", None]
    else:
        ref = oracle.render_rays(fw, rays, nc, nf, drv, p36, bg=bg, t_rand=t_rand, u=u, lindisp=lindisp, white_background=white)
        names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    for nm, o in zip(names, outs):
        if nm is not None:
            close(o, ref[nm], 2e-3, 2e-4, nm)


def test_empty_and_maximum_sizes(ops, setup):
    fw, packed, frame, drv, p36 = setup
    empty = torch.zeros(0, 8, device=dev())
    outs = ops.render_rays(packed, frame, empty, 64, 64, t_rand=torch.zeros(0, 64, device=dev()), u=torch.zeros(0, 64, device=dev()))
    assert all(o.shape[0] == 0 for o in outs)
    assert ops.stratified_depths(empty, 64).shape == (0, 64)
    assert ops.ray_uniforms(1, 0, 0, 0, 64, dev()).shape == (0, 64)
    # S = 256 is the largest per-ray sample count of composite / resample (one wave x 4 samples per lane)
    rays, rng = make_rays(9, 3)
    z = np.sort(rng.uniform(0.48, 1.08, (9, 256)).astype(np.float32), axis=1)
    raw = (rng.standard_normal((9, 256, 16)) * 1.5).astype(np.float32)
    rgb, disp, acc, w, depth = ops.composite_forward(T(raw), T(z), T(rays))
    r = oracle.composite(raw, z, rays[:, 3:6])
    close(w, r[3], 2e-5, 1e-7, "weights S=256")
    close(rgb, r[0], 2e-5, 2e-6, "rgb S=256")
    zs = ops.resample(T(z[:, :128]), T(np.abs(raw[:, :128, 0])), 128, u=T(rng.uniform(0, 1, (9, 128)).astype(np.float32)))
    assert zs.shape == (9, 256) and bool((zs[:, 1:] >= zs[:, :-1]).all())
    with pytest.raises(pkg("_lib").SahsError):
        ops.composite_forward(torch.zeros(2, 257, 16, device=dev()), torch.zeros(2, 257, device=dev()), T(rays[:2]))
    with pytest.raises(pkg("_lib").SahsError):
        ops.render_rays(packed, frame, T(rays), 200, 100, t_rand=None, u=None)          # 300 samples in the fine pass


def test_error_behaviour(ops, setup):
    """The product path refuses what it cannot run on the GPU instead of falling back."""
    fw, packed, frame, drv, p36 = setup
    E = pkg("_lib").SahsError
    rays, _ = make_rays(4, 1)
    with pytest.raises(E):
        ops.stratified_depths(torch.from_numpy(rays), 8)                      # CPU tensor
    with pytest.raises(E):
        ops.stratified_depths(T(rays).double(), 8)                            # wrong dtype
    with pytest.raises(E):
        ops.field_forward(packed, frame, 0, T(rays)[:, :6].contiguous(), torch.zeros(4, 8, device=dev()))   # rays need near/far columns
    with pytest.raises(E):
        ops.field_forward(packed, frame, 2, T(rays), torch.zeros(4, 8, device=dev()))                        # level must be 0 or 1
    with pytest.raises(E):
        ops.fold_conditioning(T(fw), torch.zeros(16, 28, device=dev()), torch.eye(4, device=dev())[:3])      # audio window shape
    with pytest.raises(E):
        ops.pack_weights(T(fw)[:-1])                                          # not this model's parameter count
    with pytest.raises(E):
        ops.pack_weights(T(fw), precision=7)
    assert "sahs_" in str(pytest.raises(E, ops.field_forward, packed, frame, 2, T(rays), torch.zeros(4, 8, device=dev())).value)


def test_ray_bundle_by_mask():
    """nerf_helpers.py:122-175: posed rays inside the mask, camera-frame rays from the origin outside."""
    sahs = pkg()
    g = load_golden("rays")
    H = W = 8
    c2w = T(g["c2w"])
    intr = g["intrinsics"]
    mask = (torch.rand(H, W, device=dev()) > 0.5).float()
    ro_m, rd_m = sahs.get_ray_bundle_by_mask(H, W, intr, c2w, mask)
    ro, rd = sahs.get_ray_bundle(H, W, intr, c2w)
    _, rd_cam = sahs.get_ray_bundle(H, W, intr, torch.eye(3, 4, device=dev()))
    inside = mask.bool()
    assert torch.equal(ro_m[inside], ro[inside]) and torch.equal(rd_m[inside], rd[inside])
    assert not bool(ro_m[~inside].any()) and torch.equal(rd_m[~inside], rd_cam[~inside])
    assert bool((rd_cam[..., 2] == -1.0).all())


def test_side_stream_and_graph_capture(ops, setup):
    """The library launches only on the stream it is given and never allocates or synchronises: the same chunk rendered on a side
    stream and replayed from a captured graph (torch.cuda.CUDAGraph = hipGraph) is bit-identical to the default-stream result."""
    fw, packed, frame, drv, p36 = setup
    N = 300
    rays, rng = make_rays(N, 17)
    rays_t = T(rays)
    bg = T(np.concatenate([rng.uniform(0, 1, (N, 3)), np.ones((N, 1)), np.zeros((N, 11))], 1).astype(np.float32))
    t_rand, u = T(rng.uniform(0, 1, (N, 64)).astype(np.float32)), T(rng.uniform(0, 1, (N, 64)).astype(np.float32))
    run = lambda: ops.render_rays(packed, frame, rays_t, 64, 64, bg=bg, t_rand=t_rand, u=u)
    ref = [o.clone() for o in run()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        outs = run()
    side.synchronize()
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run()                                   # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            captured = run()
    for o in captured:
        o.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(captured, ref):
        assert torch.equal(a, b)


def test_launch_probe_and_backward_precision_edges(ops, setup):
    """The round-3 additions to the C ABI at their edges: a probe smaller than the number of launches counts the rest as dropped (and says
    so instead of returning a short list), re-arming resets it, reading past the count is an error; the backward GEMM precision switch
    rejects anything but its two values and is restored."""
    L = pkg("_lib").lib()
    import ctypes
    _, packed, frame, _, _ = setup
    rays, z = T(make_rays(8, 3)[0]), torch.zeros(8, 4, device=dev()) + 0.6
    with ops.LaunchProbe(2) as probe:
        for _ in range(3):
            ops.field_forward(packed, frame, 0, rays, z)
        assert L.sahs_probe_count() == 2 and L.sahs_probe_dropped() == 1
        with pytest.raises(Exception):
            probe.records()
    with ops.LaunchProbe(4) as probe:
        ops.field_forward(packed, frame, 1, rays, z)
        recs = probe.records()
    assert len(recs) == 1 and recs[0]["level"] == 1 and recs[0]["part"] == 0 and recs[0]["samples"] == 32 and recs[0]["ms"] > 0.0
    k, n, ms = ctypes.c_int(), ctypes.c_long(), ctypes.c_float()
    assert L.sahs_probe_read(5, ctypes.byref(k), ctypes.byref(n), ctypes.byref(ms)) != 0
    assert L.sahs_probe_arm(0) != 0
    assert ops.backward_gemm_precision() == "bf16x3"
    assert L.sahs_backward_gemm_precision(1) == -1 and ops.backward_gemm_precision() == "bf16x3"
    assert ops.backward_gemm_precision("fp32") == "fp32" and ops.backward_gemm_precision() == "fp32"
    assert ops.backward_gemm_precision("bf16x3") == "bf16x3"
