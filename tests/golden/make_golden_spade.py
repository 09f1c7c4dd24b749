#!/usr/bin/env python3
"""Golden vectors for the Stage-II SPADE refiner's building blocks from the REAL reference:
``SPADELayer`` and ``SPADEBlock`` of /root/reference/nerf-pytorch/nerf/_init_spade.py (:114-160, :235-282), its ``Generator`` (:315-325) and ``Generator_audio`` (:327-372), imported unmodified through
the same harness shim as make_golden.py (plus an empty ``torchvision`` stub: the file imports torchvision.models for its VGG loss class,
which is off this path; torchvision is not installed here).  Runs only in the build container.  Stores, per case, the module's
state_dict (random init under a fixed seed -- small channel counts keep the file small), the inputs and the outputs in eval mode
(spectral_norm then uses its stored u, v without a power iteration).  Data only: no reference text.

usage:  python tests/golden/make_golden_spade.py            (writes tests/golden/spade.npz)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/nerf-pytorch"


def import_spade():
    if not os.path.isdir(REF + "/nerf"):
        raise SystemExit("reference not present at %s: golden vectors can only be regenerated in the build container" % REF)
    pkg = types.ModuleType("nerf")
    pkg.__path__ = [REF + "/nerf"]
    sys.modules["nerf"] = pkg
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tv.models
    return importlib.import_module("nerf._init_spade")


def main():
    S = import_spade()
    out = {}
    torch.manual_seed(1234)
    cases = {"layer": dict(norm_nc=6, label_nc=5, N=2, H=12, W=10, fh=6, fw=5),          # F_id at half resolution: nearest upsampling
             "block": dict(cin=6, cout=8, fid=4, N=2, H=8, W=8, fh=8, fw=8, down=False, up=False),
             "block_down": dict(cin=6, cout=6, fid=4, N=1, H=8, W=12, fh=4, fw=6, down=True, up=False),
             "block_up": dict(cin=8, cout=4, fid=3, N=1, H=6, W=6, fh=12, fw=12, down=False, up=True)}
    for name, c in cases.items():
        if name == "layer":
            m = S.SPADELayer(c["norm_nc"], c["label_nc"]).eval()
            x = torch.randn(c["N"], c["norm_nc"], c["H"], c["W"]) * 2.0 + 0.5
            fid = torch.randn(c["N"], c["label_nc"], c["fh"], c["fw"])
        else:
            m = S.SPADEBlock(c["cin"], c["cout"], c["fid"], downsample=c["down"], upsample=c["up"]).eval()
            x = torch.randn(c["N"], c["cin"], c["H"], c["W"]) * 2.0 + 0.5
            fid = torch.randn(c["N"], c["fid"], c["fh"], c["fw"])
        with torch.no_grad():
            y = m(x, fid)
        out[name + ":x"], out[name + ":fid"], out[name + ":y"] = x.numpy(), fid.numpy(), y.numpy()
        out[name + ":cfg"] = np.array([int(v) for v in c.values()], np.int64)
        for k, v in m.state_dict().items():
            out[name + ":sd:" + k] = v.detach().numpy()
        print(name, tuple(x.shape), "->", tuple(y.shape), "state_dict keys:", len(m.state_dict()))
    # The whole Stage-II generator (:315-325; 17.3 M parameters -- not stored: a fixed seed regenerates them, the drop-in's constructor
    # creates its parameters in the reference's order, and a checksum pins that the two initialisations are the same tensor for tensor)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    mine = importlib.import_module("sahs-deformable-nerf_amd.spade")
    torch.manual_seed(7)
    G = S.Generator().eval()
    torch.manual_seed(7)
    Gm = mine.Generator().eval()
    sd, sdm = G.state_dict(), Gm.state_dict()
    assert list(sd.keys()) == list(sdm.keys()), "state_dict layout differs from the reference's"
    assert all(torch.equal(sd[k], sdm[k]) for k in sd), "seeded initialisation differs from the reference's"
    gen = torch.Generator().manual_seed(11)
    i_src, i_raw = torch.rand(1, 3, 64, 64, generator=gen), torch.rand(1, 3, 64, 64, generator=gen)
    with torch.no_grad():
        y = G(i_src, i_raw)
        f1, f2, f3 = G.idencoder(i_src)
    out["generator:i_src"], out["generator:i_raw"], out["generator:y"] = i_src.numpy(), i_raw.numpy(), y.numpy()
    out["generator:fid3"] = f3.numpy()
    out["generator:seed"] = np.array(7, np.int64)
    out["generator:param_count"] = np.array(sum(v.numel() for v in sd.values()), np.int64)
    out["generator:param_abs_sum"] = np.array(sum(float(v.double().abs().sum()) for v in sd.values()), np.float64)
    print("generator", tuple(i_src.shape), "->", tuple(y.shape), "state_dict entries:", len(sd), "values:", int(out["generator:param_count"]))
    # Generator_audio (:359-372): the same generator with the audio code as its deepest modulation map
    torch.manual_seed(9)
    Ga = S.Generator_audio().eval()
    torch.manual_seed(9)
    Gam = mine.Generator_audio().eval()
    sda, sdam = Ga.state_dict(), Gam.state_dict()
    assert list(sda.keys()) == list(sdam.keys()), "Generator_audio: state_dict layout differs from the reference's"
    assert all(torch.equal(sda[k], sdam[k]) for k in sda), "Generator_audio: seeded initialisation differs from the reference's"
    window = torch.randn(16, 29, generator=gen)
    i_src2, i_raw2 = torch.rand(1, 3, 48, 80, generator=gen), torch.rand(1, 3, 48, 80, generator=gen)      # feature maps 6 x 10 and 12 x 20: widths that do not divide 4096
    with torch.no_grad():
        ya = Ga(i_src2, i_raw2, window)
        code = Ga.AudioNet(window.unsqueeze(0))
    out["generator_audio:i_src"], out["generator_audio:i_raw"], out["generator_audio:window"] = i_src2.numpy(), i_raw2.numpy(), window.numpy()
    out["generator_audio:y"], out["generator_audio:code"] = ya.numpy(), code.numpy()
    out["generator_audio:seed"] = np.array(9, np.int64)
    out["generator_audio:param_count"] = np.array(sum(v.numel() for v in sda.values()), np.int64)
    out["generator_audio:param_abs_sum"] = np.array(sum(float(v.double().abs().sum()) for v in sda.values()), np.float64)
    print("generator_audio", tuple(i_src2.shape), "->", tuple(ya.shape), "state_dict entries:", len(sda))
    np.savez_compressed(os.path.join(HERE, "spade.npz"), **out)
    print("wrote", os.path.join(HERE, "spade.npz"), os.path.getsize(os.path.join(HERE, "spade.npz")), "bytes")


if __name__ == "__main__":
    main()
