import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def pkg(sub=None):
    """The product package; its directory name has a hyphen, so import it by string."""
    name = "sahs-deformable-nerf_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


@pytest.fixture(scope="session")
def weights_mod():
    return pkg("weights")


@pytest.fixture(scope="session")
def flat_weights(weights_mod):
    cache = {}

    def get(seed=0, density_bias=0.0, density_gain=1.0, hdr=False):
        key = (int(seed), float(density_bias), float(density_gain), bool(hdr))
        if key not in cache:
            cache[key] = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(key[0], key[1], key[2], hdr=key[3]))
        return cache[key]

    return get


def golden_rand(d):
    """Random tensors captured from the reference, in draw order."""
    return [(k.split("_")[-1], d[k]) for k in sorted(k for k in d if k.startswith("rand_"))]


VARIANT_KW = {"default": dict(), "boosted": dict(density_bias=8.0, density_gain=30.0),
              "hdr": dict(density_bias=2.0, density_gain=30.0, hdr=True)}     # = VARIANTS of tests/golden/make_golden.py


def golden_weights_kw(g):
    """The hash_state_dict arguments a golden file was generated with."""
    return dict(seed=int(g["weights_seed"]), density_bias=float(g["weights_density_bias"]), density_gain=float(g["weights_density_gain"]),
                hdr=bool(g["weights_hdr"]) if "weights_hdr" in g else False)


YARDSTICK_LOG = []     # (what, err_max, ref_err_max, err_rms, ref_err_rms, scale): printed at the end of the session


def yardstick(got, ref32, ref64, what, factor=2.0, ulps=32.0, outlier_rays=0.0, ray_shape=None, factor_max=3.0, scale_floor=0.0):
    """Accuracy against the float64 run of the reference model (tests/golden/make_golden.py, Float64Yardstick):
    the result must be as close to the exact value of the reference's formulas as the reference's own fp32 result is --
    rms |got - f64| <= 2 * rms |ref_fp32 - f64| + eps and max |got - f64| <= 3 * max |ref_fp32 - f64| + eps (the max over a few
    hundred samples of an error that the encodings amplify by up to 2^14 is heavy-tailed: two equally accurate fp32 runs differ
    by more than 2x in it), eps = ``ulps`` fp32 ulps of the tensor's largest magnitude: the f32 MFMA (and the oracle) sum each dense layer as ONE k-ordered fmaf chain over up to 373 terms, which
    carries more round-off than ATen's blocked sums (rms 2e-8 vs 6e-9 on the default-scale network; where a dot product
    cancels, e.g. fc_alpha over 256 features, up to ~20 ulps of the result).  32 ulps = 1.9e-6 relative.

    ``outlier_rays`` > 0 (fine-pass outputs only; ``ray_shape`` = leading dims that index rays): outputs chained through the
    importance resampling inherit the genuine discontinuities of the reference's inverse CDF (``denom < 1e-5 -> 1``,
    nerf_helpers.py:491-492, and searchsorted knots): a sample may land elsewhere in its bin, which moves that single ray's
    output by ~1e-3 on the high-dynamic-range network -- for the reference's own fp32 run as for any other, but not on the same
    rays.  So that fraction of rays (at least one), the worst of each side, is left out of both statistics and only held to
    25 x the reference's worst error.  Returns and records the observed errors."""
    got = got.detach().cpu().numpy() if hasattr(got, "detach") else np.asarray(got)
    got, ref32, ref64 = (np.asarray(a, np.float64).reshape(np.asarray(ref64).shape) for a in (got, ref32, ref64))
    fin = np.isfinite(ref64)
    assert np.array_equal(np.isfinite(got), fin), what + ": non-finite entries differ from the reference's"
    e, r = np.where(fin, np.abs(got - ref64), 0.0), np.where(fin, np.abs(ref32 - ref64), 0.0)
    scale = max(float(np.abs(ref64[fin]).max()), scale_floor)   # scale_floor=1 for rendered outputs: weights, colours, acc live in [0,1]
    eps = ulps * 2.0 ** -24 * scale
    if outlier_rays > 0.0:
        nr = int(np.prod(ray_shape if ray_shape is not None else e.shape[:1]))
        e, r = e.reshape(nr, -1), r.reshape(nr, -1)
        k = max(1, int(np.ceil(outlier_rays * nr)))
        keep_e, keep_r = np.argsort(e.max(1))[:nr - k], np.argsort(r.max(1))[:nr - k]
        assert e.max() <= 25.0 * r.max() + eps, "%s: an outlier ray is off by %.3e (reference fp32's worst: %.3e)" % (what, e.max(), r.max())
        e, r = e[keep_e], r[keep_r]
    rec = (what, float(e.max()), float(r.max()), float(np.sqrt((e ** 2).mean())), float(np.sqrt((r ** 2).mean())), scale)
    YARDSTICK_LOG.append(rec)
    assert rec[1] <= factor_max * rec[2] + eps, "%s: max |got - f64| = %.3e, reference fp32's own = %.3e (scale %.3e)" % (what, rec[1], rec[2], scale)
    assert rec[3] <= factor * rec[4] + eps, "%s: rms |got - f64| = %.3e, reference fp32's own = %.3e (scale %.3e)" % (what, rec[3], rec[4], scale)
    return rec


def pytest_terminal_summary(terminalreporter):
    if YARDSTICK_LOG:
        terminalreporter.write_line("float64 yardstick (max / rms error vs the reference model in float64; 'ref' = the reference's own fp32):")
        for what, e, r, er, rr, sc in YARDSTICK_LOG:
            terminalreporter.write_line("  %-44s max %.2e (ref %.2e)  rms %.2e (ref %.2e)  scale %.2e" % (what, e, r, er, rr, sc))


def free_port():
    """A TCP port that is free now (bind to 0, read it back): each multi-process test gets its own rendezvous port instead of a fixed one
    that the previous test may have left in TIME_WAIT."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
