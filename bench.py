#!/usr/bin/env python3
"""bench.py -- rendered rays/s of the deformable-NeRF hot path on MI355X.

Workload W512 (BASELINE.json configs[1]): one 512x512 frame = 262,144 rays, 64 coarse +
128 fine field evaluations per ray (num_coarse 64, num_fine 64 as in config/audio/person_2_auto.yml),
8x256 radiance MLP + 6x128 warp + 6x64 hyper-sheet MLPs, fp32, validation mode (perturb on, noise off),
background prior on, hash-filled HIGH-DYNAMIC-RANGE weights (weights.hash_state_dict(hdr=True): O(1) activations,
semi-transparent volume, so that accuracy figures mean something), synthetic audio/pose.  A "step" is one frame
through the drop-in driver seam ITSELF -- get_ray_bundle + run_one_iter_of_nerf(mode="validation"): conditioning fold, and
per 131,072-ray chunk: depths, coarse field, composite, resample+sort, fine field, composite -- every ray's 8-tuple
written in place into one (R, 36) row block.  Inputs are resident in HBM.  fp32: the fine field is two launches -- the deformation nets for the
64 NEW depths, then the radiance net for all 128 sorted depths -- because the deformed points of the 64 coarse depths
are kept from the coarse launch instead of being recomputed as the reference does (bit-identical outputs; the
roofline still counts the reference's algorithmic FLOPs, frac_executed the instructions actually issued).

N GPUs, one process per GPU: `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run) when no launcher
has; under torchrun it is a rank.  The step is then run_one_iter_of_nerf's ray-sharded mode (_shard): the frame's rays
are split into N contiguous blocks (no exchange while rendering), then ONE in-place all-gather of the 36 floats/ray rows
(RCCL).  Total work is fixed => "scaling": "strong".

Prints ONE JSON line (rank 0).  roofline: the DOMINANT KERNEL -- the radiance-net launch over the fine samples of a ray block,
59 % of a frame -- against the fp32 MFMA peak: achieved = its algorithmic FLOPs (2 x 757,760 MAC per sample) / its average
launch time, measured with HIP events on the launch stream inside the timed region by the library's launch probe
(include/sahs_nerf.h: sahs_probe_*); frac_executed prices the MACs the kernel actually issues (padded tiles, per-frame
constants folded away); roofline.chain prices all field launches together against the reference's algorithmic work
(1,855,744 FLOP per sample evaluation).  At N=1 the line also carries:
  cpu_baseline        the reference's CPU path: the torch-eager restatement (oracle/torch_eager.py, pinned to the reference
                      by tests/test_torch_eager_vs_golden.py) with torch.set_num_threads(all host cores) on the central 64x64
                      crop of the same frame, whose outputs are also checked against the GPU frame (configs[0]-size parity);
  torch_gpu_baseline  the same restatement in plain PyTorch-ROCm on this GPU, full frame: the denominator of the north star's
                      ">= 10x the reference single-GPU PyTorch path";
  bf16                configs[2], with the PSNR protocol of SURVEY.md section 8d on the high-dynamic-range network;
  nerface_fp32, num_fine128, train_T2048 (backward products in f32), train_T2048_bf16x3 (the default: split-bf16 backward), spade
                      the secondary workloads of SURVEY.md section 8d / 8f.
Only this file's cpu_baseline / torch_gpu_baseline legs import anything under oracle/ (the thing timed there, never the product path).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_SAMPLE = {"audio": 1_855_744,            # BASELINE.md section 3 (GEMM work as the reference writes it)
                   "nerface": 2 * 719_168}        # the same count for NeRFaceModel (config/expression/person_2.yml)
PEAK_TFLOPS = {"fp32": 157.3,        # MI355X_MICROARCH.md: FP32 matrix peak (v_mfma_f32_16x16x4_f32)
               "bf16": 2500.0,       # dense BF16 MFMA peak (never the 2:1-sparse figure)
               "bf16x3": 2500.0 / 3}   # three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi): the peak of the ALGORITHMIC work
KERNEL = {"fp32": "field_forward_f32_kernel", "bf16": "field_forward_bf16w_kernel", "bf16x3": "field_radiance_bf16x3_kernel"}
HDR = dict(seed=0, density_bias=2.0, density_gain=30.0, hdr=True)      # = VARIANTS["hdr"] of tests/golden/make_golden.py
# HBM-side bytes of the dominant dispatch (the launch over the 16.8 M fine samples of a ray chunk) from the committed rocprofv3 PMC passes
# (WRITE_SIZE + 2 x FETCH_SIZE, the gfx950 correction for 16 B/lane streaming reads).  STATIC: bench.py cannot collect PMCs itself; the
# numbers are read from the summary that tools/profile_r2.sh wrote under profiles/.
def _static_traffic():
    for name in ("r4_pmc_summary.json", "r3_pmc_summary.json", "r2_pmc_summary.json"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", name)))
            out = {"fp32": (d["f32_radiance"]["traffic_bytes"], "profiles/%s:f32_radiance" % name),
                   "bf16": (d["bf16_radiance"]["traffic_bytes"], "profiles/%s:bf16_radiance" % name)}
            if "bf16x3_radiance" in d:
                out["bf16x3"] = (d["bf16x3_radiance"]["traffic_bytes"], "profiles/%s:bf16x3_radiance" % name)
            return out
        except (OSError, KeyError, ValueError):
            continue
    return {}


TRAFFIC = _static_traffic()


# ---------------------------------------------------------------------------------------------------------------------------
# control flow shared by every leg and by tests/test_distributed_gloo.py (which runs main() itself over gloo with a CPU renderer)
# ---------------------------------------------------------------------------------------------------------------------------
def timed_steps(step, steps, warmup, dist=None, sync=lambda: None, device="cpu"):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides; MAX over ranks."""
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1

    def barrier():
        if multi:
            dist.barrier()
        sync()

    out = None
    for _ in range(warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def headline_record(value, ms_per_step, world, steps, warmup, dtype, config, roofline, ranks, backend):
    return {"metric": "rendered rays/sec (coarse64+fine128, 8x256 MLP)", "value": value, "unit": "rays/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic", "config": config, "roofline": roofline, "rccl_ranks": ranks, "collective_backend": backend}


def run_headline(renderer, world, steps, warmup, dist, sync, dtype, config, roofline_fn, device="cpu", backend=None):
    """The headline measurement -> the JSON record (every rank computes it; rank 0 prints).  A step is renderer.frame(): ONE whole
    frame, every rank rendering its ray block and the all-gather completing it (the product's own sharded path)."""
    dt, out = timed_steps(renderer.frame, steps, warmup, dist, sync, device)
    renderer.check(out)
    ranks = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rec = headline_record(renderer.num_rays * steps / dt, dt / steps * 1e3, world, steps, warmup, dtype, config, roofline_fn(dt), ranks, backend)
    return rec, out


# ---------------------------------------------------------------------------------------------------------------------------
# the renderer: the drop-in driver seam itself (run_one_iter_of_nerf), its field launches timed by the library's launch probe
# ---------------------------------------------------------------------------------------------------------------------------
class ProductRenderer:
    """One frame = get_ray_bundle + run_one_iter_of_nerf(mode="validation") -- what evaluation.render_frames does per frame
    (eval_stage_rays.py:454-475).  shard=True: the ray-sharded mode of that same function (every rank its block + one all-gather)."""

    def __init__(self, pkg, dev, size, precision="fp32", arch="audio", num_fine=None, shard=False):
        W = pkg.weights
        self.pkg, self.ops, self.dev, self.arch, self.precision_name, self.shard = pkg, pkg.ops, dev, arch, precision, shard
        rng = np.random.default_rng(42)
        if arch == "audio":
            self.cfg = pkg.default_config()
            self.fw = W.flatten_state_dict(W.hash_state_dict(**HDR))
            self.model = pkg.AudioFaceModel(self.cfg, precision=precision).to(dev).load_flat(self.fw).eval()
            self.audio = torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev)
            cam_z = 0.8
        else:   # expression-driven NeRFaceModel: driving = 76-d expression, near/far 0.2/0.8 (config/expression/person_2.yml:43-45)
            self.cfg = pkg.default_config("expression")
            # high-dynamic-range weights; the density logit placed so that rays spread their weight over many samples (mean w_bg ~0.2)
            self.fw = W.flatten_state_dict(W.hash_state_dict(0, -3.0, 10.0, model="nerface", hdr=True), model="nerface")
            self.model = pkg.NeRFaceModel(self.cfg, precision=precision).to(dev).load_flat(self.fw).eval()
            self.audio = torch.from_numpy((rng.standard_normal(76) * 0.5).astype(np.float32)).to(dev)
            cam_z = 0.5
        for p in self.model.parameters():
            p.requires_grad_(False)
        self.pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam_z]]], axis=1).astype(np.float32)).to(dev)
        self.H = self.W = size
        self.R = self.num_rays = size * size
        self.intr = np.array([1200.0 * size / 512, 1200.0 * size / 512, 0.5, 0.5], np.float32)
        bg = np.concatenate([rng.uniform(0, 1, (self.R, 3)), np.ones((self.R, 1)), np.zeros((self.R, 11))], axis=1).astype(np.float32)
        self.bg_all = torch.from_numpy(bg).to(dev)
        opt = self.cfg.nerf.validation
        if num_fine is not None:
            opt.num_fine = int(num_fine)
        self.nc, self.nf, self.chunk = int(opt.num_coarse), int(opt.num_fine), int(opt.chunksize)
        self.near, self.far = float(self.cfg.dataset.near), float(self.cfg.dataset.far)
        self.seed = int(self.cfg.experiment.randomseed)
        self.prec = self.model.precision
        self.flop_per_sample = FLOP_PER_SAMPLE[arch]
        self.split = arch != "nerface_static" and self.nf > 0      # the driver evaluates the deformation nets once per depth
        self.mixed = self.ops.is_mixed(arch, self.prec)            # split-chain-only precisions (split-operand deformation + low-precision radiance launches)
        ex = lambda part: 2 * self.ops.executed_macs_per_sample(arch, self.prec, part)
        self.exec_flop_per_sample = ex(0)
        # executed FLOPs per RAY: coarse = nc whole-network evaluations; fine = nf deformation + (nc + nf) radiance evaluations when split
        self.exec_flop_per_ray = (self.nc * ex(0) + self.nf * ex(1) + (self.nc + self.nf) * ex(2)) if self.split else (2 * self.nc + self.nf) * ex(0)

    def frame(self):
        """-> the reference's 8-tuple for the whole frame (on every rank).  Draws keyed by (seed, global ray index): the frame is the same
        for any chunking and any number of GPUs."""
        pkg = self.pkg
        ro, rd = pkg.get_ray_bundle(self.H, self.W, self.intr, self.pose)
        with torch.no_grad(), pkg.train_utils.partition_invariant_rng(self.seed):
            return pkg.run_one_iter_of_nerf(self.H, self.W, self.intr, self.model, ro, rd, self.cfg, mode="validation", driving=self.audio,
                                            pose=self.pose, background_prior=self.bg_all, _shard=True if self.shard else None)

    def check(self, out):
        assert len(out) == 8 and tuple(out[3].shape) == (self.H, self.W, 15) and tuple(out[7].shape) == (self.H, self.W), [tuple(o.shape) for o in out]
        assert all(bool(torch.isfinite(o).all()) for o in out)

    def roofline(self, dt, recs):
        """recs: the launch probe's records of the timed steps (this rank's field launches, HIP events on the launch stream).
        The PRIMARY figures (achieved / frac) are the dominant kernel's -- the radiance launch over the fine samples of a ray block --
        on its own algorithmic FLOPs; `chain` prices all field launches together against the reference's algorithmic work."""
        field_ms = sum(r["ms"] for r in recs)
        # sample evaluations the reference would have made: its fine pass runs the whole network on all nc + nf depths
        samples = sum(r["samples"] for r in recs if r["part"] != 1)
        peak = PEAK_TFLOPS[self.precision_name]
        traffic, src = TRAFFIC.get(self.precision_name, (None, None))
        chain_achieved = samples * self.flop_per_sample / (field_ms * 1e-3) / 1e12
        rays_done = samples / (2 * self.nc + self.nf)
        chain = {"what": "all field launches of the timed steps against the REFERENCE's algorithmic work (%d FLOP per sample evaluation, %d "
                         "evaluations per ray)" % (self.flop_per_sample, 2 * self.nc + self.nf) +
                         ("; the fine pass here skips the reference's second evaluation of the deformation nets at the coarse depths (bit-identical "
                          "results), so this exceeds frac_executed by more than constant folding alone" if self.split else ""),
                 "achieved": chain_achieved, "frac": chain_achieved / peak,
                 "frac_executed": rays_done * self.exec_flop_per_ray / (field_ms * 1e-3) / 1e12 / peak,
                 "launches": len(recs), "field_ms_per_step": field_ms / max(1, self.steps_timed), "field_time_share": field_ms * 1e-3 / dt,
                 "flop_per_sample": self.flop_per_sample, "flop_per_sample_executed": self.exec_flop_per_sample, "shared_deformation": self.split}
        if self.mixed:      # split-operand deformation launches (three bf16 MFMAs per product) + bf16 / split-operand radiance launches: ONE pipe
            chain["frac_executed"] = rays_done * self.exec_flop_per_ray / (field_ms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"]
            chain["mixed_precision"] = ("deformation launches with split bf16 operands (three MFMAs per product) + low-precision radiance launches, all on the "
                                        "bf16 matrix pipe: frac_executed prices the issued MFMA work against its 2.5 PFLOP/s; the chain figures are NOT a kernel roofline")
        # the dominant dispatch = the LEVEL-1 radiance launch (all nc + nf sorted depths of a ray block); a split chain's level-0 launch
        # (whole network, or radiance nets over the nc coarse depths) is a different, smaller launch and is not averaged into it
        rad = [r for r in recs if r["part"] == 2 and r["level"] == 1] or [r for r in recs if r["part"] == 0 and r["level"] == 1]
        kname = {"fp32": "field_forward_f32_kernel<false, 2>", "bf16": "field_forward_bf16w_kernel<2>", "bf16x3": "field_radiance_bf16x3_kernel"}[self.precision_name]
        rms, rsm = sum(r["ms"] for r in rad), sum(r["samples"] for r in rad)
        part = 2 if rad and rad[0]["part"] == 2 else 0
        alg = {("audio", 2): 2 * 757_760, ("audio", 0): FLOP_PER_SAMPLE["audio"], ("nerface", 0): FLOP_PER_SAMPLE["nerface"],
               ("nerface_static", 0): None}.get((self.arch, part))
        executed = rsm * 2 * self.ops.executed_macs_per_sample(self.arch, self.prec, part) / (rms * 1e-3) / 1e12
        # bf16x3 issues three MFMAs per product: its executed FLOPs are priced against the pipe's 2.5 PFLOP/s
        frac_exec = executed / (PEAK_TFLOPS["bf16"] if self.precision_name == "bf16x3" else peak)
        achieved = rsm * alg / (rms * 1e-3) / 1e12 if alg else executed
        return {"bound": "mfma", "kernel": kname + (" (radiance nets over the fine samples of a ray block: the dominant dispatch)" if part == 2 else ""),
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": (achieved / peak) if alg else frac_exec,
                "frac_executed": frac_exec, "executed_tflops": executed,
                "avg_launch_ms": rms / max(1, len(rad)), "launches": len(rad), "samples_per_launch": rsm // max(1, len(rad)),
                "flop_per_sample": alg, "timing": "HIP events recorded on the launch stream around each launch by the library's launch probe "
                                                  "(sahs_probe_*), timed steps only",
                "traffic": traffic, "traffic_source": None if src is None else src + " (static: rocprofv3 --pmc passes of this command, per "
                "launch of this kernel; bench.py cannot collect PMCs)",
                "chain": chain}


_T0 = time.perf_counter()


def progress(msg):
    """Leg-by-leg progress on stderr (stdout carries only the JSON line)."""
    print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def psnr(a, b):
    mse = float(torch.mean((a.clamp(0.0, 1.0) - b.clamp(0.0, 1.0)) ** 2))
    return 150.0 if mse == 0.0 else -10.0 * float(np.log10(mse))


def measure(pkg, dev, size, precision, steps, warmup, arch="audio", num_fine=None, world=1, dist=None):
    """Timed frames of one workload through the drop-in driver seam -> (JSON record, the last frame's 8-tuple, renderer)."""
    r = ProductRenderer(pkg, dev, size, precision, arch, num_fine, shard=world > 1)
    config = {"workload": "W%d: %dx%d rays, %d coarse + %d fine evaluations/ray, deform(6x128+6x64)+radiance(%s) MLPs, validation mode "
                          "(perturb on), bg prior, hash-filled high-dynamic-range weights, through run_one_iter_of_nerf"
                          % (size, size, size, r.nc, r.nc + r.nf, "8x256" if arch == "audio" else "4x256"),
              "rays_per_step": r.R, "ray_chunk": r.chunk, "parallelism": "rays x%d" % world + (" (contiguous ray blocks, one in-place all-gather of "
              "(R,36) rows per frame)" if world > 1 else ""), "precision": precision, "model": arch}
    r.steps_timed = steps
    with pkg.ops.LaunchProbe(capacity=16384) as probe:
        def roofline(dt):
            recs = probe.records()
            per_step = len(recs) // (steps + warmup)
            assert per_step * (steps + warmup) == len(recs) and per_step > 0, (len(recs), steps, warmup)
            return r.roofline(dt, recs[-steps * per_step:])          # the timed region's launches only

        rec, out = run_headline(r, world, steps, warmup, dist, torch.cuda.synchronize,
                                {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3 (hi + lo bf16 operands, three MFMAs per product, f32 accumulate; all nets)"}[precision],
                                config, roofline, device=dev, backend="nccl (RCCL)" if world > 1 else None)
    return rec, out, r


def rgb_fine(out):
    """(R, 3) colours of the fine pass from the driver's 8-tuple."""
    return out[3].reshape(-1, 15)[:, :3]


def rows_of(out):
    """The 8-tuple as (R, 36) rows (SAHS_ROW_* order)."""
    R = out[1].numel()
    return torch.cat([o.reshape(R, -1) for o in out], dim=1)


def add_secondary_legs(result, pkg, dev, args):
    """configs[2] (bf16) with the PSNR protocol, NeRFaceModel, num_fine 128 and the T2048 training step, beside the fp32 headline."""
    size = args.size
    # BASELINE.json configs[2]: same workload through the bf16-MFMA field kernel (fp32 accumulate)
    progress("bf16 leg")
    rec16, out16, r16 = measure(pkg, dev, size, "bf16", max(args.steps, 5), 2)
    # PSNR protocol of SURVEY.md section 8d on the high-dynamic-range network: the fp32 frame is the reference image; the pseudo-target
    # T is the SAME network rendered under other random draws (another seed of the keyed uniforms), fp32
    progress("bf16 PSNR protocol")
    r32 = ProductRenderer(pkg, dev, size, "fp32")
    f32 = r32.frame()
    r32.seed += 1000
    tgt = r32.frame()
    p_b, p_f = psnr(rgb_fine(out16), rgb_fine(tgt)), psnr(rgb_fine(f32), rgb_fine(tgt))
    result["bf16"] = {"value": rec16["value"], "unit": "rays/s", "ms_per_step": rec16["ms_per_step"], "dtype": "bf16", "roofline": rec16["roofline"],
                      "psnr_bf16_vs_fp32_db": psnr(rgb_fine(out16), rgb_fine(f32)), "psnr_bf16_vs_target_db": p_b, "psnr_fp32_vs_target_db": p_f,
                      "delta_psnr_db": abs(p_b - p_f), "max_abs_rgb_diff": float((rgb_fine(out16) - rgb_fine(f32)).abs().max()),
                      "psnr_note": "rgb_fine of the same frame (same high-dynamic-range weights, rays and draws) by the bf16 and the fp32 kernel; "
                                   "target T = the same network under other draws (fp32); the north-star bound is delta_psnr <= 0.05 dB.  The "
                                   "reference has no bf16 run and no checkpoint is available offline: accuracy here is 'parity unpinned' "
                                   "(synthetic weights), judged by that bound -- individual pixels differ by up to max_abs_rgb_diff"}
    del r16, out16, f32, tgt
    # the expression-driven NeRFaceModel (config/expression/person_2.yml) on the same frame, fp32
    progress("nerface leg")
    rec, _, _ = measure(pkg, dev, size, "fp32", min(args.steps, 5), 1, arch="nerface")
    result["nerface_fp32"] = {"value": rec["value"], "unit": "rays/s", "ms_per_step": rec["ms_per_step"], "dtype": "f32",
                              "workload": rec["config"]["workload"], "roofline": rec["roofline"]}
    # the same model in mixed precision (deformation nets with split bf16 operands, plain-bf16 radiance nets) and the section-8d PSNR protocol on it
    progress("nerface mixed-precision leg")
    recm, outm, rm = measure(pkg, dev, size, "bf16", min(args.steps, 5), 1, arch="nerface")
    rn = ProductRenderer(pkg, dev, size, "fp32", arch="nerface")
    f32n = rn.frame()
    rn.seed += 1000
    tgtn = rn.frame()
    p_b, p_f = psnr(rgb_fine(outm), rgb_fine(tgtn)), psnr(rgb_fine(f32n), rgb_fine(tgtn))
    result["nerface_mixed_bf16"] = {"value": recm["value"], "unit": "rays/s", "ms_per_step": recm["ms_per_step"], "dtype": "bf16 radiance nets + deformation nets with split bf16 operands (hi + lo, three MFMAs per product); f32 accumulate",
                                    "speedup_vs_nerface_fp32": recm["value"] / rec["value"], "roofline": recm["roofline"],
                                    "psnr_mixed_vs_fp32_db": psnr(rgb_fine(outm), rgb_fine(f32n)), "psnr_mixed_vs_target_db": p_b,
                                    "psnr_fp32_vs_target_db": p_f, "delta_psnr_db": abs(p_b - p_f), "w_bg_mean": float(f32n[6].mean())}
    del rm, rn, outm, f32n, tgtn
    # near-fp32 on the bf16 pipe (round 3: the deformation nets too): every net with bf16 hi/lo operands (3 MFMAs per product)
    progress("bf16x3 leg")
    recx, outx, rx = measure(pkg, dev, size, "bf16x3", min(args.steps, 5), 1)
    f32a = rows_of(ProductRenderer(pkg, dev, size, "fp32").frame())
    rowx = rows_of(outx)
    dx = (rowx - f32a).abs()
    coarse_cols = list(range(0, 17))
    result["bf16x3"] = {"value": recx["value"], "unit": "rays/s", "ms_per_step": recx["ms_per_step"],
                        "dtype": "bf16 hi/lo operands in every net (3 MFMAs per product, f32 accumulate)",
                        "speedup_vs_fp32": recx["value"] / result["value"], "roofline": recx["roofline"],
                        "psnr_vs_fp32_db": psnr(rgb_fine(outx), f32a[:, 17:20]), "max_abs_diff_coarse_outputs": float(dx[:, coarse_cols].max()),
                        "rays_within_4x_fp32_tolerance": float(((dx <= 4e-5 + 4e-4 * f32a.abs()).all(dim=1)).float().mean())}
    del rx, outx, rowx, f32a, dx
    # the num_fine 128 reading (fine pass of 192 samples, 256 evaluations per ray)
    progress("num_fine128 leg")
    rec, _, _ = measure(pkg, dev, size, "fp32", 2, 1, num_fine=128)
    result["num_fine128"] = {"value": rec["value"], "unit": "rays/s", "ms_per_step": rec["ms_per_step"], "dtype": "f32",
                             "workload": rec["config"]["workload"], "roofline": rec["roofline"]}
    torch.cuda.empty_cache()
    progress("train_T2048 leg (backward products in f32: the reference's arithmetic)")
    result["train_T2048"] = train_leg(pkg, dev, backward="fp32")
    progress("train_T2048_bf16x3 leg (the library's default: backward products as three bf16 MFMAs)")
    result["train_T2048_bf16x3"] = train_leg(pkg, dev, backward="bf16x3")
    progress("train_T2048_x3fwd leg (opt-in: the saving forward on the split-operand kernels too)")
    result["train_T2048_x3fwd"] = train_leg(pkg, dev, backward="bf16x3", forward="bf16x3")
    progress("spade leg (Stage-II generator inference, SURVEY.md section 8f-4)")
    result["spade"] = spade_leg(pkg, dev)


def train_leg(pkg, dev, rays=2048, steps=5, warmup=2, backward="fp32", forward="fp32"):
    """BASELINE.json configs[4]: 2048 semantically-weighted rays, train mode (noise 0.1), forward + backward through the HIP autograd
    op and the reference's loss recipe (train_stage_rays_auto.py:437-499); no optimiser step (not part of the path).
    backward: arithmetic of the backward's dense-layer products -- "fp32" = f32 MFMAs, the reference's precision (the entry named
    train_T2048); "bf16x3" = split bf16 operands, three bf16 MFMAs per product (train_T2048_bf16x3, the library's default).
    forward: arithmetic of the saving forward launches -- "fp32" (the fp32 kernel, both entries above) or "bf16x3" (the split-operand
    kernels also write the saved activations and sign bits: ops.training_forward_precision, the entry train_T2048_x3fwd)."""
    W, Tr = pkg.weights, pkg.training
    cfg = pkg.default_config()
    model = pkg.AudioFaceModel(cfg).to(dev).load_flat(W.flatten_state_dict(W.hash_state_dict(**HDR))).train()
    g = torch.Generator(device=dev).manual_seed(3)
    H = Wd = 128
    mask = torch.zeros(H, Wd, 12, device=dev)
    mask.scatter_(2, torch.randint(0, 12, (H, Wd, 1), device=dev, generator=g), 1.0)
    probs = Tr.semantic_ray_probs(torch.ones(12, device=dev) / 12, mask)
    sel = Tr.sample_training_rays(probs, rays, g)
    audio = torch.randn(16, 29, device=dev, generator=g)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    intr = np.array([1200.0 * H / 512, 1200.0 * H / 512, 0.5, 0.5], np.float32)
    ro, rd = pkg.get_ray_bundle(H, Wd, intr, pose)
    ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
    m = mask.reshape(-1, 12)[sel]
    target = torch.rand(rays, 3, device=dev, generator=g)
    bg = torch.cat([torch.rand(rays, 3, device=dev, generator=g), torch.ones(rays, 1, device=dev), torch.zeros(rays, 11, device=dev)], 1)

    cw = Tr.sample_prob_weights(dev)

    def step():
        # as training.train_step: the objective and its gradient inside the HIP launches (sahs_stage1_loss_forward, composite backward)
        outs = pkg.run_one_iter_of_nerf(H, Wd, intr, model, ro, rd, cfg, mode="train", driving=audio, pose=pose, background_prior=bg, inHead=m,
                                        _loss=(target, m, cw))
        loss = outs[8]
        model.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    before, before_fwd = pkg.ops.backward_gemm_precision(), pkg.ops.training_forward_precision()
    pkg.ops.backward_gemm_precision(backward)
    pkg.ops.training_forward_precision(forward)
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        pkg.ops.backward_gemm_precision(before)
        pkg.ops.training_forward_precision(before_fwd)
    assert bool(torch.isfinite(loss))
    return train_record(pkg, rays, dt, steps, backward, forward=forward)


def train_record(pkg, rays, dt, steps, backward, nc=64, nf=64, forward="fp32"):
    """The JSON entry of a training leg.  roofline.frac = the time the step's EXECUTED matrix work needs at each pipe's peak, summed over
    the pipes, divided by the step time: forward launches (fp32 kernel: nc whole-network + nf deformation + (nc + nf) radiance evaluations
    per ray, padded tiles as issued) on the f32 MFMA pipe; backward = two products per forward MAC (data gradient, weight gradient) on the
    f32 pipe ("fp32") or as three bf16 MFMAs each on the bf16 pipe ("bf16x3").  frac_algorithmic is SURVEY.md section 8d's rule
    (3 x forward GEMM FLOPs per training ray / fp32 peak) and is NOT a pipe roofline when the backward runs on the bf16 pipe."""
    ex = lambda part: 2 * pkg.ops.executed_macs_per_sample("audio", pkg.ops.SAHS_F32, part)
    fwd_exec = rays * (nc * ex(0) + nf * ex(1) + (nc + nf) * ex(2))                  # FLOPs issued by the saving forward launches
    bwd_products = 2 * fwd_exec                                                        # dX and dW: one product each per forward MAC
    f32_flops = (fwd_exec if forward == "fp32" else 0) + (bwd_products if backward == "fp32" else 0)
    bf16_flops = 3 * bwd_products if backward == "bf16x3" else 0
    if forward == "bf16x3":      # the split-operand kernels' own padded tiles (32-row tiles), as three bf16 MFMAs per product
        ex3 = lambda part: 2 * pkg.ops.executed_macs_per_sample("audio", pkg.ops.SAHS_BF16X3, part)
        bf16_flops += rays * ((nc + nf) * ex3(1) + (nc + nc + nf) * ex3(2))
    t_f32, t_bf16 = f32_flops / (PEAK_TFLOPS["fp32"] * 1e12), bf16_flops / (PEAK_TFLOPS["bf16"] * 1e12)
    alg = rays * (2 * nc + nf) * FLOP_PER_SAMPLE["audio"] * 3 / dt / 1e12
    dtype = {"fp32": "f32", "bf16x3": "f32 forward + split-bf16 backward GEMMs (hi + lo operands, 3 bf16 MFMAs per product, f32 accumulate)"}[backward]
    if forward == "bf16x3":
        dtype = "split-bf16 GEMMs in forward and backward (hi + lo operands, 3 bf16 MFMAs per product, f32 accumulate)" if backward == "bf16x3" else \
                "split-bf16 forward GEMMs (hi + lo operands, 3 bf16 MFMAs per product, f32 accumulate) + f32 backward GEMMs"
    return {"workload": "T%d: %d semantically-weighted rays, 64+128 evaluations/ray, train mode (noise 0.1), forward + loss recipe + backward"
                        % (rays, rays), "ms_per_step": dt * 1e3, "value": rays / dt, "unit": "rays/s", "dtype": dtype,
            "backward_gemm_precision": backward, "training_forward_precision": forward, "steps": steps,
            "backward_walk": "fused (one data-gradient chain + job-table weight-gradient launches per part)" if pkg.ops.fused_backward() else "per-layer GEMM launches",
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "frac": (t_f32 + t_bf16) / dt,
                         "what": "executed MFMA FLOPs per pipe / that pipe's peak, summed, / step time",
                         "pipes": {"f32_mfma": {"executed_tflop": f32_flops / 1e12, "peak": PEAK_TFLOPS["fp32"], "ms_at_peak": t_f32 * 1e3},
                                   "bf16_mfma": {"executed_tflop": bf16_flops / 1e12, "peak": PEAK_TFLOPS["bf16"], "ms_at_peak": t_bf16 * 1e3}},
                         "achieved_algorithmic": alg, "peak_algorithmic": PEAK_TFLOPS["fp32"], "frac_algorithmic": alg / PEAK_TFLOPS["fp32"],
                         "flop_rule_algorithmic": "3 x forward GEMM FLOPs per training ray (SURVEY.md section 8d) against the fp32 MFMA peak; "
                                                  "a pipe roofline only for backward_gemm_precision fp32"}}


def spade_leg(pkg, dev, size=512, steps=10, warmup=3):
    """SURVEY.md section 8f-4 (Stage-II refiner, _init_spade.py:284-325): `Generator` inference on a size x size pair (convolutions on MIOpen),
    and the ONE hand-written kernel pair of that path -- instance statistics + fused normalise / modulate / activate -- on the largest SPADE
    layer of the network (64 channels at size/2 x size/2), timed with HIP events against its algorithmic 20 B per element (HBM roof)."""
    ops = pkg.ops
    g = torch.Generator(device=dev).manual_seed(1)
    torch.manual_seed(1)
    net = pkg.spade.Generator().to(dev).eval()
    I_src, I_raw = torch.rand(1, 3, size, size, device=dev, generator=g), torch.rand(1, 3, size, size, device=dev, generator=g)
    with torch.no_grad():
        for _ in range(warmup):
            out = net(I_src, I_raw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = net(I_src, I_raw)
        torch.cuda.synchronize()
        gen_ms = (time.perf_counter() - t0) / steps * 1e3
    assert tuple(out.shape) == (1, 3, size, size) and bool(torch.isfinite(out).all())
    C, H = 64, size // 2
    x, ga, be = (torch.randn(1, C, H, H, device=dev, generator=g) for _ in range(3))
    for _ in range(3):
        ops.spade_modulate(x, ga, be, slope=0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        ops.spade_modulate(x, ga, be, slope=0.2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 20 * x.numel()
    return {"workload": "Stage-II Generator (IdEncoder + 6 SPADE blocks) on a %dx%d source / raw pair, inference, fp32; convolutions on MIOpen" % (size, size),
            "ms_per_image": gen_ms, "value": 1e3 / gen_ms, "unit": "images/s", "dtype": "f32",
            "roofline": {"bound": "hbm", "kernel": "instance_stats_kernel + spade_modulate_kernel<true> on (1, %d, %d, %d)" % (C, H, H),
                         "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0,
                         "avg_pair_ms": ms, "bytes_per_element": 20, "elements": x.numel(),
                         "timing": "HIP events on the launch stream around %d statistics + modulate launch pairs" % reps, "traffic": None}}


def host_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a job a share of
    its cores; 256 threads on a 16-core share made the eager CPU path 30x slower than 16 threads)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("SAHS_BENCH_CPU_THREADS")
    return int(env) if env else n


def add_baselines(result, out, rend, pkg, dev):
    """torch_gpu_baseline and cpu_baseline: the reference's own op sequence (oracle/torch_eager.py) on this GPU and on the host cores."""
    from oracle import torch_eager as TE        # baseline legs only: the thing timed here, never the product path
    W = pkg.weights
    sd_np = W.hash_state_dict(**HDR)
    R, H, Wd, nc, nf = rend.R, rend.H, rend.W, rend.nc, rend.nf
    ro, rd = pkg.get_ray_bundle(H, Wd, rend.intr, rend.pose)
    # the frame's own draws (keyed by global ray index), so the baselines render the SAME frame as the HIP path
    t_rand, u = pkg.ops.ray_uniforms(rend.seed, 0, 0, R, nc, dev), pkg.ops.ray_uniforms(rend.seed, 1, 0, R, nf, dev)
    chunk = rend.chunk
    rand = [dict(t_rand=t_rand[s:s + chunk], u=u[s:s + chunk]) for s in range(0, R, chunk)]
    progress("torch_gpu_baseline (eager restatement on this GPU)")
    field = TE.EagerField({k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()})
    run_gpu = lambda: TE.run_one_iter(field, ro, rd, rend.near, rend.far, rend.audio, rend.pose, bg=rend.bg_all, rand=rand, perturb=True, chunksize=chunk)
    gtimes = []
    with torch.no_grad():
        run_gpu()
        torch.cuda.synchronize()
        for _ in range(3):
            t0 = time.perf_counter()
            eager = run_gpu()
            torch.cuda.synchronize()
            gtimes.append(time.perf_counter() - t0)
    gdt = float(np.median(gtimes))
    names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    hip = [o.reshape(R, -1) for o in out]
    result["torch_gpu_baseline"] = {"value": R / gdt, "unit": "rays/s", "ms_per_frame": gdt * 1e3, "dtype": "f32",
                                    "what": "plain PyTorch-ROCm eager restatement of the reference (same op sequence, chunksize 131072) on the same "
                                            "GPU, frame, weights and draws; one warm frame, then the median of 3 timed frames (%s ms)"
                                            % ", ".join("%.0f" % (t * 1e3) for t in gtimes),
                                    "speedup_fp32": result["value"] / (R / gdt),
                                    "psnr_hip_vs_eager_db": psnr(hip[3][:, :3], eager[3].reshape(R, -1)[:, :3]),
                                    "max_abs_diff": {n: float((a - b.reshape(R, -1)).abs().max()) for n, a, b in zip(names, hip, eager)}}
    if "bf16" in result:
        result["torch_gpu_baseline"]["speedup_bf16"] = result["bf16"]["value"] / (R / gdt)
    del field, eager
    torch.cuda.empty_cache()
    # ---- CPU: central 64x64 crop of the same frame, all host cores, warm, median of 3 ----
    cores = host_cores()
    progress("cpu_baseline (eager restatement on %d host cores)" % cores)
    torch.set_num_threads(cores)
    cs = min(64, H)
    c0 = (H - cs) // 2
    idx = (torch.arange(c0, c0 + cs, device=dev)[:, None] * Wd + torch.arange(c0, c0 + cs, device=dev)[None, :]).reshape(-1)
    cpu = lambda t: t.detach().cpu()
    ro_c, rd_c = cpu(ro.reshape(-1, 3)[idx]), cpu(rd.reshape(-1, 3)[idx])
    rnd = [dict(t_rand=cpu(t_rand[idx]), u=cpu(u[idx]))]
    field_c = TE.EagerField({k: torch.from_numpy(v) for k, v in sd_np.items()})
    run_cpu = lambda n: TE.run_one_iter(field_c, ro_c[:n], rd_c[:n], rend.near, rend.far, cpu(rend.audio), cpu(rend.pose), bg=cpu(rend.bg_all[idx])[:n],
                                        rand=[dict(t_rand=rnd[0]["t_rand"][:n], u=rnd[0]["u"][:n])], perturb=True)
    times = []
    with torch.no_grad():
        run_cpu(256)                                     # warm: thread pool, oneDNN primitives
        for _ in range(3):
            t0 = time.perf_counter()
            ref = run_cpu(cs * cs)
            times.append(time.perf_counter() - t0)
            progress("cpu run %.1f s" % times[-1])
            if sum(times) > 30.0:                        # bounded sample: stop early on a slow host
                break
    cdt = float(np.median(times))
    # configs[0]-size parity for free: the crop by the reference's CPU path against the same rays of the GPU frame.  The network is
    # the high-dynamic-range one, on which the reference's own fp32 run is 1e-4..1e-3 away from its float64 run in the chained
    # fine-pass outputs, a few rays by much more (inverse-CDF discontinuities; tests/golden/e2e_hdr_*.npz, conftest.yardstick), so the
    # check is statistical: 99 % of the rays within 1e-3, none beyond 0.1.
    worst, frac_ok = {}, 1.0
    for n, a, b in zip(names, hip, ref):
        d = (cpu(a[idx]) - b.reshape(cs * cs, -1)).abs().max(dim=1).values
        worst[n] = float(d.max())
        frac_ok = min(frac_ok, float((d <= 1e-3).float().mean()))
    assert frac_ok >= 0.99 and max(worst.values()) <= 0.1, ("configs[0]-size parity (CPU reference path vs GPU frame)", frac_ok, worst)
    result["cpu_baseline"] = {"value": cs * cs / cdt, "unit": "rays/s", "cores": cores, "kind": "port",
                              "sample": "central %dx%d ray crop of the same frame (same weights and draws, 64+128 evaluations/ray): the reference's "
                                        "PyTorch-CPU path as its torch-eager restatement, torch.set_num_threads(%d), warm, median of %d runs (%s s)"
                                        % (cs, cs, cores, len(times), ", ".join("%.1f" % t for t in times)),
                              "parity_vs_gpu_frame": {"rays_within_1e-3": frac_ok, "max_abs_diff": worst}}
    progress("cpu_port (C oracle)")
    from oracle import oracle                      # the C/OpenMP port of the oracle, for orientation (not the reference's own path)
    t0 = time.perf_counter()
    oracle.run_one_iter_of_nerf(rend.fw, ro_c.numpy(), rd_c.numpy(), rend.near, rend.far, nc, nf, cpu(rend.audio).numpy(), cpu(rend.pose).numpy(),
                                background_prior=cpu(rend.bg_all[idx]).numpy(), rand=[dict(t_rand=rnd[0]["t_rand"].numpy(), u=rnd[0]["u"].numpy())])
    pdt = time.perf_counter() - t0
    result["cpu_port"] = {"value": cs * cs / pdt, "unit": "rays/s", "cores": cores, "kind": "port",
                          "sample": "the same crop by the C oracle (oracle/sahs_oracle.c, OpenMP over points), %.1f s" % pdt}


def launch_children(gpus, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh processes (one per GPU) through torch.distributed.run and pass
    their output through.  The parent has made NO GPU call (nothing here touches torch.cuda), is never replaced by exec, and exits
    with the children's status; rank 0's JSON line is the children's only stdout.
    Wall-clock guard: a wedged rendezvous or collective must not end as "killed at the driver's limit, nothing written".  After
    SAHS_BENCH_GUARD_S seconds (default 420) the parent terminates the children's process group (its own fresh children only), prints ONE
    JSON line {"error": ..., "n_gpus": N} and exits non-zero."""
    import signal
    import socket
    import subprocess
    import threading
    guard_s = float(os.environ.get("SAHS_BENCH_GUARD_S", "420"))
    rc = 1
    for attempt in range(2):      # the rendezvous port is free when picked but not reserved: if the launcher then finds it taken, pick another, once
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        progress("no launcher in the environment: starting %d ranks (guard %.0f s): %s" % (gpus, guard_s, " ".join(cmd)))
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors="replace", start_new_session=True)      # own process group
        port_taken, json_lines = [], []

        def relay():
            for line in child.stderr:
                sys.stderr.write(line)
                if "ddress already in use" in line or "EADDRINUSE" in line:
                    port_taken.append(line)
            sys.stderr.flush()

        def relay_out():
            for line in child.stdout:
                sys.stdout.write(line)
                sys.stdout.flush()
                if line.lstrip().startswith("{"):
                    json_lines.append(line)

        th = threading.Thread(target=relay, daemon=True)
        th.start()
        tho = threading.Thread(target=relay_out, daemon=True)
        tho.start()
        try:
            rc = child.wait(timeout=guard_s)
        except subprocess.TimeoutExpired:
            for sig, grace in ((signal.SIGTERM, 15), (signal.SIGKILL, 15)):      # the group we started, by its exact id -- never a pattern
                try:
                    os.killpg(child.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    child.wait(timeout=grace)
                    break
                except subprocess.TimeoutExpired:
                    continue
            th.join(timeout=5)
            tho.join(timeout=5)
            print(json.dumps({"error": "bench.py --gpus %d: the ranks did not finish within the %.0f s wall-clock guard (SAHS_BENCH_GUARD_S) and were "
                                       "terminated: a wedged rendezvous or collective; rerun with NCCL_DEBUG=INFO, and SAHS_ALLGATHER_OUT_OF_PLACE=1 "
                                       "selects the out-of-place all-gather" % (gpus, guard_s), "n_gpus": gpus}), flush=True)
            return 124
        th.join(timeout=10)
        tho.join(timeout=10)
        if rc == 0 or not port_taken:
            if rc != 0 and not json_lines:      # a rank died (exception, collective timeout, watchdog abort): say so on stdout, as ONE JSON line
                print(json.dumps({"error": "bench.py --gpus %d: the ranks exited with status %d before rank 0 printed its line (see stderr); "
                                           "collective timeout SAHS_BENCH_DIST_TIMEOUT_S, fallback SAHS_ALLGATHER_OUT_OF_PLACE=1" % (gpus, rc),
                                  "n_gpus": gpus}), flush=True)
            break
        progress("rendezvous port %d was taken by another process: retrying once on a new port" % port)
    return rc


def load_injected_renderer(spec, args, world, rank):
    """SAHS_BENCH_RENDERER=<file.py>:<factory> -- tests/test_distributed_gloo.py rehearses main() on CPU (SAHS_BENCH_BACKEND=gloo) with a
    renderer of its own; factory(args, world, rank) -> object with .frame(), .check(out), .num_rays.  Never set on a GPU run."""
    import importlib.util
    path, name = spec.rsplit(":", 1)
    sp = importlib.util.spec_from_file_location("sahs_bench_injected_renderer", path)
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    return getattr(mod, name)(args, world, rank)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="frame is size x size rays")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="fp32 = configs[1] (exact, headline); bf16 = configs[2] (bf16 MFMA operands, fp32 accumulate)")
    ap.add_argument("--arch", default="audio", choices=["audio", "nerface"], help="audio = AudioFaceModel (the headline); nerface = the "
                    "expression-driven NeRFaceModel of config/expression/person_2.yml (profiling of that leg; fp32 or bf16 = mixed precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the torch_gpu_baseline / cpu_baseline legs")
    ap.add_argument("--no-secondary", action="store_true", help="headline only (no bf16 / NeRFace / num_fine128 / training legs)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # BEFORE any GPU call: the parent only launches and relays
        return launch_children(args.gpus, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (torchrun --nproc-per-node must equal --gpus)" % (args.gpus, world))
    backend = os.environ.get("SAHS_BENCH_BACKEND", "nccl")
    injected = os.environ.get("SAHS_BENCH_RENDERER")
    rehearsal = backend != "nccl"
    # SAHS_BENCH_ONE_GPU=1 (tests/test_gpu_sharded.py): every rank drives cuda:0 and the collectives go over gloo (RCCL refuses two ranks
    # on one device) -- the REAL renderer, launch probe and roofline code of an N > 1 run, rehearsed on a one-GPU box; not a measurement
    one_gpu = rehearsal and os.environ.get("SAHS_BENCH_ONE_GPU") == "1"
    if rehearsal and not injected and not one_gpu:
        raise SystemExit("bench.py: SAHS_BENCH_BACKEND=%s is a rehearsal of the control flow and needs SAHS_BENCH_RENDERER or SAHS_BENCH_ONE_GPU=1" % backend)
    dist = None
    if rehearsal and not one_gpu:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a MI355X"
        torch.cuda.set_device(0 if one_gpu else local)
        dev = torch.device("cuda", 0 if one_gpu else local)
    if world > 1:
        import datetime
        import torch.distributed as dist
        # a short collective timeout: a rank that never arrives fails the others' rendezvous / collective with an exception (non-zero exit,
        # relayed by launch_children) instead of hanging until the driver's limit
        tmo = datetime.timedelta(seconds=float(os.environ.get("SAHS_BENCH_DIST_TIMEOUT_S", "120")))
        dist.init_process_group(backend, timeout=tmo, **({} if rehearsal else {"device_id": dev}))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    t_start = time.perf_counter()
    progress("headline")
    if injected:      # control-flow rehearsal (tests): the same launch branch, rendezvous, timing, MAX-reduce and JSON line
        rend = load_injected_renderer(injected, args, world, rank)
        result, out = run_headline(rend, world, args.steps, args.warmup, dist, lambda: None, "f32", {"workload": "rehearsal: " + rend.describe()},
                                   lambda dt: None, device=dev, backend=backend)
        result["rehearsal"] = "control flow only (CPU, %s, injected renderer): not a measurement" % backend
    else:
        pkg = importlib.import_module("sahs-deformable-nerf_amd")
        result, out, rend = measure(pkg, dev, args.size, args.precision, args.steps, args.warmup, arch=args.arch, world=world, dist=dist)
        if one_gpu:
            result["collective_backend"] = backend
            result["rehearsal"] = "every rank on cuda:0, %s collectives: the N > 1 code path on one GPU, not a measurement" % backend
        if args.precision == "fp32" and world == 1 and args.arch == "audio":
            if not args.no_secondary:
                add_secondary_legs(result, pkg, dev, args)
            if not args.no_cpu_baseline:
                add_baselines(result, out, rend, pkg, dev)
    progress("done")
    result["bench_wall_s"] = time.perf_counter() - t_start
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
