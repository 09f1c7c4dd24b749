"""Stage-II SPADE building blocks: the CPU restatement against the reference's own outputs (fixtures from the
imported reference, tests/golden/make_golden_spade.py), the drop-in modules' state_dict ABI, and -- on the GPU -- the drop-in modules
(MIOpen convolutions + the fused HIP modulation kernel through the C ABI) against the same fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pkg

CASES = {"block": (False, False), "block_down": (True, False), "block_up": (False, True)}


def _sd(g, name, device="cpu"):
    pre = name + ":sd:"
    return {k[len(pre):]: torch.from_numpy(g[k]).to(device) for k in g if k.startswith(pre)}


def test_spade_restatement_vs_reference():
    from oracle import spade_eager as SE
    g = load_golden("spade")
    sd = {"L." + k: v for k, v in _sd(g, "layer").items()}
    y = SE.spade_layer(sd, "L.", torch.from_numpy(g["layer:x"]), torch.from_numpy(g["layer:fid"]))
    assert np.abs(y.numpy() - g["layer:y"]).max() <= 1e-4 * np.abs(g["layer:y"]).max()      # (the host's convolution summation order: 2e-6 here, 2e-5 on another CPU)
    for name, (down, up) in CASES.items():
        y = SE.spade_block(_sd(g, name), torch.from_numpy(g[name + ":x"]), torch.from_numpy(g[name + ":fid"]), downsample=down, upsample=up)
        assert y.shape == g[name + ":y"].shape
        assert np.abs(y.numpy() - g[name + ":y"]).max() <= 1e-4 * np.abs(g[name + ":y"]).max(), name


def test_spade_modules_have_the_reference_state_dict():
    S = pkg("spade")
    g = load_golden("spade")
    for name, (down, up) in CASES.items():
        cin, cout, fid = [int(v) for v in g[name + ":cfg"][:3]]
        m = S.SPADEBlock(cin, cout, fid, downsample=down, upsample=up)
        want = _sd(g, name)
        have = m.state_dict()
        assert list(have.keys()) == list(want.keys()), name           # same keys in the same order (incl. conv1 / conv1_sn duplicates)
        assert all(tuple(have[k].shape) == tuple(want[k].shape) for k in want), name
        m.load_state_dict(want)                                       # strict
    with pytest.raises(Exception):                                    # the fused modulation has no CPU path
        with torch.no_grad():
            S.SPADELayer(6, 5)(torch.zeros(1, 6, 4, 4), torch.zeros(1, 5, 4, 4))


def _seeded_generator(g):
    """The drop-in Generator under the fixture's seed: the reference's own initialisation, tensor for tensor (make_golden_spade.py asserts
    it against the imported reference; the checksum pins it here)."""
    S = pkg("spade")
    torch.manual_seed(int(g["generator:seed"]))
    G = S.Generator().eval()
    sd = G.state_dict()
    assert sum(v.numel() for v in sd.values()) == int(g["generator:param_count"])
    total = sum(float(v.double().abs().sum()) for v in sd.values())
    assert abs(total - float(g["generator:param_abs_sum"])) <= 1e-9 * total, "seeded initialisation differs from the fixture's (another torch build?)"
    return G


def test_generator_restatement_vs_reference():
    """The whole Stage-II generator (IdEncoder + RefineNetwork, 17.3 M parameters regenerated from the fixture's seed) by the CPU restatement
    against the reference's output; the identity encoder's deepest feature map as an inner seam."""
    from oracle import spade_eager as SE
    g = load_golden("spade")
    sd = _seeded_generator(g).state_dict()
    with torch.no_grad():
        y = SE.generator(sd, torch.from_numpy(g["generator:i_src"]), torch.from_numpy(g["generator:i_raw"]))
        f3 = SE.id_encoder({k[len("idencoder."):]: v for k, v in sd.items() if k.startswith("idencoder.")}, torch.from_numpy(g["generator:i_src"]))[2]
    assert np.abs(f3.numpy() - g["generator:fid3"]).max() <= 1e-4 * np.abs(g["generator:fid3"]).max()
    assert y.shape == g["generator:y"].shape
    assert np.abs(y.numpy() - g["generator:y"]).max() <= 2e-4 * np.abs(g["generator:y"]).max()


def _seeded_generator_audio(g):
    S = pkg("spade")
    torch.manual_seed(int(g["generator_audio:seed"]))
    G = S.Generator_audio().eval()
    sd = G.state_dict()
    assert sum(v.numel() for v in sd.values()) == int(g["generator_audio:param_count"])
    total = sum(float(v.double().abs().sum()) for v in sd.values())
    assert abs(total - float(g["generator_audio:param_abs_sum"])) <= 1e-9 * total, "seeded initialisation differs from the fixture's (another torch build?)"
    return G


def test_generator_audio_restatement_vs_reference():
    """Generator_audio (_init_spade.py:327-372: the audio code as the deepest modulation map) by the CPU restatement -- which materialises the
    reference's (1, 256, 64, 4096) map -- against the reference's output on a 48 x 80 frame (feature-map widths 10 and 20 do not divide 4096);
    and the drop-in's TiledCodeMap against nearest-neighbour interpolation of the materialised map."""
    from oracle import spade_eager as SE
    S = pkg("spade")
    g = load_golden("spade")
    G = _seeded_generator_audio(g)
    sd = G.state_dict()
    window = torch.from_numpy(g["generator_audio:window"])
    with torch.no_grad():
        code = SE.audio_code({k[len("AudioNet."):]: v for k, v in sd.items() if k.startswith("AudioNet.")}, window)
        assert np.abs(code.numpy() - g["generator_audio:code"]).max() <= 1e-5 * np.abs(g["generator_audio:code"]).max()
        assert np.abs(G.AudioNet(window.unsqueeze(0)).numpy() - g["generator_audio:code"]).max() <= 1e-5 * np.abs(g["generator_audio:code"]).max()
        y = SE.generator_audio(sd, torch.from_numpy(g["generator_audio:i_src"]), torch.from_numpy(g["generator_audio:i_raw"]), window)
    ref = g["generator_audio:y"]
    assert y.shape == ref.shape and np.abs(y.numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
    m = S.TiledCodeMap(torch.arange(64, dtype=torch.float32) * 0.5 - 7.0, channels=2, height=64, tiles=64)
    big = m.materialise()
    assert tuple(big.shape) == (1, 2, 64, 4096)
    for size in ((6, 10), (12, 20), (8, 8), (64, 64), (5, 4096), (3, 8192), (7, 97), (1, 1)):
        assert torch.equal(m.nearest(size), torch.nn.functional.interpolate(big, size=size, mode="nearest")), size


@pytest.mark.gpu
def test_generator_audio_vs_reference_on_gpu():
    g = load_golden("spade")
    dev = torch.device("cuda:0")
    G = _seeded_generator_audio(g).to(dev)
    T = lambda k: torch.from_numpy(g["generator_audio:" + k]).to(dev)
    with torch.no_grad():
        y = G(T("i_src"), T("i_raw"), T("window"))
    ref = g["generator_audio:y"]
    assert tuple(y.shape) == ref.shape
    assert float((y.cpu() - torch.from_numpy(ref)).abs().max()) <= 5e-4 * float(np.abs(ref).max())


@pytest.mark.gpu
def test_generator_vs_reference_on_gpu():
    g = load_golden("spade")
    dev = torch.device("cuda:0")
    G = _seeded_generator(g).to(dev)
    with torch.no_grad():
        y = G(torch.from_numpy(g["generator:i_src"]).to(dev), torch.from_numpy(g["generator:i_raw"]).to(dev))
    ref = g["generator:y"]
    assert tuple(y.shape) == ref.shape
    assert float((y.cpu() - torch.from_numpy(ref)).abs().max()) <= 5e-4 * float(np.abs(ref).max())      # 7 SPADE blocks of MIOpen convolutions deep


@pytest.mark.gpu
def test_spade_modules_vs_reference_on_gpu():
    S, ops = pkg("spade"), pkg("ops")
    dev = torch.device("cuda:0")
    g = load_golden("spade")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    norm_nc, label_nc = [int(v) for v in g["layer:cfg"][:2]]
    layer = S.SPADELayer(norm_nc, label_nc).to(dev).eval()
    layer.load_state_dict(_sd(g, "layer", dev))
    with torch.no_grad():
        y = layer(T(g["layer:x"]), T(g["layer:fid"]))
    scale = float(np.abs(g["layer:y"]).max())
    assert float((y.cpu() - torch.from_numpy(g["layer:y"])).abs().max()) <= 1e-4 * scale        # MIOpen's convolution order vs the CPU's
    for name, (down, up) in CASES.items():
        cin, cout, fid = [int(v) for v in g[name + ":cfg"][:3]]
        m = S.SPADEBlock(cin, cout, fid, downsample=down, upsample=up).to(dev).eval()
        m.load_state_dict(_sd(g, name, dev))
        with torch.no_grad():
            y = m(T(g[name + ":x"]), T(g[name + ":fid"]))
        ref = g[name + ":y"]
        assert tuple(y.shape) == ref.shape
        assert float((y.cpu() - torch.from_numpy(ref)).abs().max()) <= 1e-4 * float(np.abs(ref).max()), name
    # the fused kernel alone against the formula, incl. a ragged plane size and a constant plane (variance 0)
    x = torch.randn(3, 5, 7, 9, device=dev) * 3 + 1
    x[1, 2] = 0.25
    ga, be = torch.randn_like(x), torch.randn_like(x)
    want = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x, eps=1e-5) * (1 + ga) + be, 0.2)
    assert float((ops.spade_modulate(x, ga, be, slope=0.2) - want).abs().max()) <= 2e-5
    with pytest.raises(Exception):
        ops.spade_modulate(x, ga[:, :4], be)
