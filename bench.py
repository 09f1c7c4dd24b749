#!/usr/bin/env python3
"""bench.py -- rendered rays/s of the deformable-NeRF hot path on MI355X.

Workload W512 (BASELINE.json configs[1]): one 512x512 frame = 262,144 rays, 64 coarse +
128 fine field evaluations per ray (num_coarse 64, num_fine 64 as in config/audio/person_2_auto.yml),
8x256 radiance MLP + 6x128 warp + 6x64 hyper-sheet MLPs, fp32, validation mode (perturb on, noise off),
background prior on, hash-filled HIGH-DYNAMIC-RANGE weights (weights.hash_state_dict(hdr=True): O(1) activations,
semi-transparent volume, so that accuracy figures mean something), synthetic audio/pose.  A "step" is one frame
through the drop-in driver's launch sequence (ray bundle, conditioning fold, and per 131,072-ray chunk: depths,
coarse field, composite, resample+sort, fine field, composite), every ray's 8-tuple written in place into one
(R, 36) row block.  Inputs are resident in HBM.  fp32: the fine field is two launches -- the deformation nets for the
64 NEW depths, then the radiance net for all 128 sorted depths -- because the deformed points of the 64 coarse depths
are kept from the coarse launch instead of being recomputed as the reference does (bit-identical outputs; the
roofline still counts the reference's algorithmic FLOPs, frac_executed the instructions actually issued).

N GPUs (torchrun, one process per GPU): the frame's rays are split into N contiguous blocks (no exchange while
rendering), then ONE all-gather of the 36 floats/ray rows (RCCL).  Total work is fixed => "scaling": "strong".

Prints ONE JSON line (rank 0).  roofline: the field kernel (99 % of the frame) against the fp32 MFMA peak,
achieved = algorithmic FLOPs (1,855,744 per sample evaluation) / time of the field launches
measured with HIP events on the launch stream inside the timed region; frac_executed prices the MACs the kernel
actually issues (padded tiles, per-frame constants folded away).  At N=1 the line also carries:
  cpu_baseline        the reference's CPU path: the torch-eager restatement (oracle/torch_eager.py, pinned to the reference
                      by tests/test_torch_eager_vs_golden.py) with torch.set_num_threads(all host cores) on the central 64x64
                      crop of the same frame, whose outputs are also checked against the GPU frame (configs[0]-size parity);
  torch_gpu_baseline  the same restatement in plain PyTorch-ROCm on this GPU, full frame: the denominator of the north star's
                      ">= 10x the reference single-GPU PyTorch path";
  bf16                configs[2], with the PSNR protocol of SURVEY.md section 8d on the high-dynamic-range network;
  nerface_fp32, num_fine128, train_T2048   the secondary workloads of SURVEY.md section 8d / 8f.
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
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "r2_pmc_summary.json")))
        return {"fp32": (d["f32_radiance"]["traffic_bytes"], "profiles/r2_pmc_summary.json:f32_radiance"),
                "bf16": (d["bf16_radiance"]["traffic_bytes"], "profiles/r2_pmc_summary.json:bf16_radiance")}
    except (OSError, KeyError, ValueError):
        return {}


TRAFFIC = _static_traffic()


# ---------------------------------------------------------------------------------------------------------------------------
# control flow shared by every leg and by tests/test_distributed_gloo.py (which runs it over gloo with the CPU oracle as renderer)
# ---------------------------------------------------------------------------------------------------------------------------
def frame_step(renderer, num_rays, world, rank, gather):
    """One step of one rank: render this rank's contiguous ray block [lo, hi) into (hi-lo, 36) rows, then ONE all-gather."""
    D = importlib.import_module("sahs-deformable-nerf_amd.distributed")
    lo, hi = D.shard_bounds(num_rays, world, rank)
    rows = renderer.render(lo, hi)
    return gather(rows, num_rays) if world > 1 else rows


def timed_steps(step, steps, warmup, dist=None, sync=lambda: None):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides; MAX over ranks."""
    def barrier():
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
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
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=out.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def headline_record(value, ms_per_step, world, steps, warmup, dtype, config, roofline, rccl_ranks):
    return {"metric": "rendered rays/sec (coarse64+fine128, 8x256 MLP)", "value": value, "unit": "rays/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic", "config": config, "roofline": roofline, "rccl_ranks": rccl_ranks}


def run_headline(renderer, num_rays, world, rank, steps, warmup, dist, gather, sync, dtype, config, roofline_fn):
    """The headline measurement -> the JSON record (every rank computes it; rank 0 prints)."""
    dt, out = timed_steps(lambda: frame_step(renderer, num_rays, world, rank, gather), steps, warmup, dist, sync)
    assert tuple(out.shape) == (num_rays, 36), out.shape
    ranks = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rec = headline_record(num_rays * steps / dt, dt / steps * 1e3, world, steps, warmup, dtype, config, roofline_fn(dt), ranks)
    return rec, out


# ---------------------------------------------------------------------------------------------------------------------------
# the HIP renderer: the drop-in driver's launch chain with HIP events around the field launches
# ---------------------------------------------------------------------------------------------------------------------------
class HipRenderer:
    def __init__(self, pkg, dev, size, precision="fp32", arch="audio", num_fine=None, weights=HDR, share_deformation=True):
        W = pkg.weights
        self.pkg, self.ops, self.dev, self.arch, self.precision_name = pkg, pkg.ops, dev, arch, precision
        rng = np.random.default_rng(42)
        if arch == "audio":
            self.cfg = pkg.default_config()
            self.fw = W.flatten_state_dict(W.hash_state_dict(**weights))
            self.model = pkg.AudioFaceModel(self.cfg, precision=precision).to(dev).load_flat(self.fw)
            self.audio = torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev)
            cam_z = 0.8
        else:   # expression-driven NeRFaceModel: driving = 76-d expression, near/far 0.2/0.8 (config/expression/person_2.yml:43-45)
            self.cfg = pkg.default_config("expression")
            # high-dynamic-range weights; the density logit placed so that rays spread their weight over many samples (mean w_bg ~0.2)
            self.fw = W.flatten_state_dict(W.hash_state_dict(0, -3.0, 10.0, model="nerface", hdr=True), model="nerface")
            self.model = pkg.NeRFaceModel(self.cfg, precision=precision).to(dev).load_flat(self.fw)
            self.audio = torch.from_numpy((rng.standard_normal(76) * 0.5).astype(np.float32)).to(dev)
            cam_z = 0.5
        self.pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam_z]]], axis=1).astype(np.float32)).to(dev)
        self.H = self.W = size
        self.R = size * size
        self.intr = np.array([1200.0 * size / 512, 1200.0 * size / 512, 0.5, 0.5], np.float32)
        bg = np.concatenate([rng.uniform(0, 1, (self.R, 3)), np.ones((self.R, 1)), np.zeros((self.R, 11))], axis=1).astype(np.float32)
        self.bg_all = torch.from_numpy(bg).to(dev)
        opt = self.cfg.nerf.validation
        self.nc, self.nf, self.chunk = int(opt.num_coarse), int(opt.num_fine if num_fine is None else num_fine), int(opt.chunksize)
        self.near, self.far = float(self.cfg.dataset.near), float(self.cfg.dataset.far)
        self.seed = int(self.cfg.experiment.randomseed)
        self.prec = self.model.precision
        self.packed, _ = self.model.packed()
        self.flop_per_sample = FLOP_PER_SAMPLE[arch]
        # the deformation nets are evaluated once per depth (sahs_model_field_forward_split), as the drop-in driver does
        self.split = share_deformation and precision in ("fp32", "bf16", "bf16x3") and arch != "nerface_static"
        self.mixed = self.ops.is_mixed(arch, self.prec)               # fp32 deformation launches + low-precision radiance launches
        ex = lambda part: 2 * self.ops.executed_macs_per_sample(arch, self.prec, part)
        self.exec_flop_per_sample = ex(0)
        # executed FLOPs per RAY: coarse = nc whole-network evaluations; fine = nf deformation + (nc + nf) radiance evaluations when split
        self.exec_flop_per_ray = (self.nc * ex(0) + self.nf * ex(1) + (self.nc + self.nf) * ex(2)) if self.split else (2 * self.nc + self.nf) * ex(0)
        self.ws = {}
        self.record = False
        self.field_events = []
        self.radiance_events = []     # fp32 split chain: the radiance-net launches alone (the dominant dispatch)

    def render(self, lo, hi):
        """Rays [lo, hi) of the frame -> (hi-lo, 36) rows; the same launches, in the same order, as sahs_model_render_rays_rows."""
        ops, dev, nc, nf = self.ops, self.dev, self.nc, self.nf
        ro, rd = self.pkg.get_ray_bundle(self.H, self.W, self.intr, self.pose)
        frame = self.model.frame(self.audio, self.pose)
        n = hi - lo
        rays = torch.cat([ro.view(-1, 3)[lo:hi], rd.view(-1, 3)[lo:hi], torch.full((n, 1), self.near, device=dev),
                          torch.full((n, 1), self.far, device=dev)], dim=1)
        rows = torch.empty(n, 36, dtype=torch.float32, device=dev)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        for s in range(0, n, self.chunk):
            rb = rays[s:s + self.chunk]
            N = rb.shape[0]
            bgb = self.bg_all[lo + s: lo + s + N]
            rw = rows[s:s + N]
            t_rand = ops.ray_uniforms(self.seed, 0, lo + s, N, nc, dev)     # keyed by GLOBAL ray index: the frame does not depend on N GPUs
            z_c = ops.stratified_depths(rb, nc, False, t_rand)
            e0, e1, e2, e3 = ev(), ev(), ev(), ev()
            split = self.split and nf > 0
            if split:     # the deformation nets once per depth (ops.render_rays_rows share_deformation): coarse launch also emits x', w
                xw, key = self.ws.get(("xw", N)), ("xw", N)
                if xw is None:
                    xw = self.ws[key] = torch.empty(N, nc + nf, 8, dtype=torch.float32, device=dev)
            e0.record()
            if split:
                raw = ops.field_forward_split(self.packed, frame, 0, ops.FIELD_ALL, rb, xw, z=z_c, out=self.ws.get(("raw", N, nc)), arch=self.arch, precision=self.prec)
            else:
                raw = ops.field_forward(self.packed, frame, 0, rb, z_c, precision=self.prec, out=self.ws.get(("raw", N, nc)), arch=self.arch)
            e1.record()
            self.ws[("raw", N, nc)] = raw
            wts = ops.composite_forward_rows(raw, z_c, rb, rw, False, bg=bgb, weights=self.ws.get(("w", N, nc)))
            self.ws[("w", N, nc)] = wts
            if nf > 0:
                u = ops.ray_uniforms(self.seed, 1, lo + s, N, nf, dev)
                if split:
                    z_f, z_new, src = ops.resample_merge(z_c, wts, nf, u=u)
                    e2.record()
                    ops.field_forward_split(self.packed, frame, 1, ops.FIELD_DEFORM, rb, xw, z=z_new, xw_col0=nc, arch=self.arch, precision=self.prec)
                    em = ev()
                    em.record()
                    raw_f = ops.field_forward_split(self.packed, frame, 1, ops.FIELD_RADIANCE, rb, xw, src=src, out=self.ws.get(("raw", N, nc + nf)), arch=self.arch, precision=self.prec)
                    if self.record:
                        self.radiance_events.append((em, e3, N * (nc + nf)))
                else:
                    z_f = ops.resample(z_c, wts, nf, u=u)
                    e2.record()
                    raw_f = ops.field_forward(self.packed, frame, 1, rb, z_f, precision=self.prec, out=self.ws.get(("raw", N, nc + nf)), arch=self.arch)
                e3.record()
                self.ws[("raw", N, nc + nf)] = raw_f
                self.ws[("w", N, nc + nf)] = ops.composite_forward_rows(raw_f, z_f, rb, rw, True, bg=bgb, weights=self.ws.get(("w", N, nc + nf)))
            if self.record:
                self.field_events.append((e0, e1, N * nc))
                if nf > 0:
                    self.field_events.append((e2, e3, N * (nc + nf)))
        return rows

    def roofline(self, dt):
        field_ms = sum(a.elapsed_time(b) for a, b, _ in self.field_events)
        samples = sum(p for _, _, p in self.field_events)
        achieved = samples * self.flop_per_sample / (field_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[self.precision_name]
        traffic, src = TRAFFIC.get(self.precision_name, (None, None))
        dominant = None
        if self.radiance_events:      # the one kernel that is 59 % of the frame, priced on ITS OWN algorithmic FLOPs (the radiance net's 757,760 MAC)
            rms = sum(a.elapsed_time(b) for a, b, _ in self.radiance_events)
            rfl = sum(p for _, _, p in self.radiance_events) * 2 * self.ops.executed_macs_per_sample(self.arch, self.prec, 2)
            alg = {"audio": 2 * 757_760}.get(self.arch)
            dominant = {"kernel": {"fp32": "field_forward_f32_kernel<false, 2>", "bf16": "field_forward_bf16w_kernel<2>",
                                   "bf16x3": "field_radiance_bf16x3_kernel"}[self.precision_name] + " (radiance net over the fine samples of a ray chunk)",
                        "avg_launch_ms": rms / len(self.radiance_events), "executed_tflops": rfl / (rms * 1e-3) / 1e12,
                        # bf16x3 issues three MFMAs per product: its executed FLOPs are priced against the pipe's 2.5 PFLOP/s
                        "frac_executed": rfl / (rms * 1e-3) / 1e12 / (PEAK_TFLOPS["bf16"] if self.precision_name == "bf16x3" else peak)}
            if alg is not None:
                dominant["achieved"] = sum(p for _, _, p in self.radiance_events) * alg / (rms * 1e-3) / 1e12
                dominant["frac"] = dominant["achieved"] / peak
        return {"bound": "mfma", "kernel": KERNEL[self.precision_name], "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_source": None if src is None else src + " (static: rocprofv3 --pmc "
                "passes of this command, per fine launch; bench.py cannot collect PMCs)", "launches": len(self.field_events),
                "avg_launch_ms": field_ms / max(1, len(self.field_events)), "flop_per_sample": self.flop_per_sample,
                "flop_per_sample_executed": self.exec_flop_per_sample,
                "shared_deformation": self.split, "dominant_kernel": dominant,
                "note": ("achieved counts the reference's algorithmic FLOPs (1,855,744 per sample evaluation) of the field launches over their time; with "
                         "shared_deformation the fine pass skips the reference's redundant second evaluation of the deformation nets at the coarse "
                         "depths, so frac can exceed frac_executed (the instructions actually issued) by more than the constant folding alone") if self.split else None,
                "frac_executed": (samples / (2 * self.nc + self.nf)) * self.exec_flop_per_ray / (field_ms * 1e-3) / 1e12 / peak,
                "field_time_share": field_ms * 1e-3 / dt,
                **({"mixed_precision": "fp32 deformation launches + low-precision radiance launches: 'achieved' / 'frac' price the whole chain against "
                                       "the low-precision peak and are NOT a kernel roofline here; dominant_kernel is the radiance launch alone"} if self.mixed else {})}


_T0 = time.perf_counter()


def progress(msg):
    """Leg-by-leg progress on stderr (stdout carries only the JSON line)."""
    print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def psnr(a, b):
    mse = float(torch.mean((a.clamp(0.0, 1.0) - b.clamp(0.0, 1.0)) ** 2))
    return 150.0 if mse == 0.0 else -10.0 * float(np.log10(mse))


def measure(pkg, dev, size, precision, steps, warmup, arch="audio", num_fine=None, world=1, rank=0, dist=None):
    """Timed frames of one workload -> (JSON record, last gathered (R,36) frame, renderer)."""
    r = HipRenderer(pkg, dev, size, precision, arch, num_fine)
    gather = pkg.distributed.all_gather_rows
    config = {"workload": "W%d: %dx%d rays, %d coarse + %d fine evaluations/ray, deform(6x128+6x64)+radiance(%s) MLPs, validation mode "
                          "(perturb on), bg prior, hash-filled high-dynamic-range weights" % (size, size, size, r.nc, r.nc + r.nf,
                                                                                               "8x256" if arch == "audio" else "4x256"),
              "rays_per_step": r.R, "ray_chunk": r.chunk, "parallelism": "rays x%d" % world, "precision": precision, "model": arch}
    r.record = True

    def roofline(dt):
        per_step = len(r.field_events) // (steps + warmup)
        r.field_events = r.field_events[-steps * per_step:]        # the timed region's launches only
        r.radiance_events = r.radiance_events[-steps * (len(r.radiance_events) // (steps + warmup)):] if r.radiance_events else []
        return r.roofline(dt)

    with torch.no_grad():
        rec, out = run_headline(r, r.R, world, rank, steps, warmup, dist, gather, torch.cuda.synchronize,
                                {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3 (hi + lo operands, f32 accumulate; deformation nets f32)"}[precision], config, roofline)
    assert bool(torch.isfinite(out).all())
    return rec, out, r


def rgb_fine(rows):
    return rows[:, 17:20]


def add_secondary_legs(result, pkg, dev, args):
    """configs[2] (bf16) with the PSNR protocol, NeRFaceModel, num_fine 128 and the T2048 training step, beside the fp32 headline."""
    size = args.size
    # BASELINE.json configs[2]: same workload through the bf16-MFMA field kernel (fp32 accumulate)
    progress("bf16 leg")
    rec16, out16, r16 = measure(pkg, dev, size, "bf16", max(args.steps, 5), 2)
    # PSNR protocol of SURVEY.md section 8d on the high-dynamic-range network: the fp32 frame is the reference image; the pseudo-target
    # T is the SAME network rendered under other random draws (another seed of the keyed uniforms), fp32
    progress("bf16 PSNR protocol")
    r32 = HipRenderer(pkg, dev, size, "fp32")
    with torch.no_grad():
        f32 = r32.render(0, r32.R)
        r32.seed += 1000
        tgt = r32.render(0, r32.R)
    p_b, p_f = psnr(rgb_fine(out16), rgb_fine(tgt)), psnr(rgb_fine(f32), rgb_fine(tgt))
    result["bf16"] = {"value": rec16["value"], "unit": "rays/s", "ms_per_step": rec16["ms_per_step"], "dtype": "bf16", "roofline": rec16["roofline"],
                      "psnr_bf16_vs_fp32_db": psnr(rgb_fine(out16), rgb_fine(f32)), "psnr_bf16_vs_target_db": p_b, "psnr_fp32_vs_target_db": p_f,
                      "delta_psnr_db": abs(p_b - p_f), "max_abs_rgb_diff": float((rgb_fine(out16) - rgb_fine(f32)).abs().max()),
                      "psnr_note": "rgb_fine of the same frame (same high-dynamic-range weights, rays and draws) by the bf16 and the fp32 kernel; "
                                   "target T = the same network under other draws (fp32); the north-star bound is delta_psnr <= 0.05 dB"}
    del r16, out16, f32, tgt
    # the expression-driven NeRFaceModel (config/expression/person_2.yml) on the same frame, fp32
    progress("nerface leg")
    rec, _, _ = measure(pkg, dev, size, "fp32", min(args.steps, 5), 1, arch="nerface")
    result["nerface_fp32"] = {"value": rec["value"], "unit": "rays/s", "ms_per_step": rec["ms_per_step"], "dtype": "f32",
                              "workload": rec["config"]["workload"], "roofline": rec["roofline"]}
    # the same model in mixed precision (fp32 deformation nets, bf16 radiance nets) and the section-8d PSNR protocol on it
    progress("nerface mixed-precision leg")
    recm, outm, rm = measure(pkg, dev, size, "bf16", min(args.steps, 5), 1, arch="nerface")
    rn = HipRenderer(pkg, dev, size, "fp32", arch="nerface")
    with torch.no_grad():
        f32n = rn.render(0, rn.R)
        rn.seed += 1000
        tgtn = rn.render(0, rn.R)
    p_b, p_f = psnr(rgb_fine(outm), rgb_fine(tgtn)), psnr(rgb_fine(f32n), rgb_fine(tgtn))
    result["nerface_mixed_bf16"] = {"value": recm["value"], "unit": "rays/s", "ms_per_step": recm["ms_per_step"], "dtype": "f32 deformation nets + bf16 radiance nets",
                                    "speedup_vs_nerface_fp32": recm["value"] / rec["value"], "roofline": recm["roofline"],
                                    "psnr_mixed_vs_fp32_db": psnr(rgb_fine(outm), rgb_fine(f32n)), "psnr_mixed_vs_target_db": p_b,
                                    "psnr_fp32_vs_target_db": p_f, "delta_psnr_db": abs(p_b - p_f), "w_bg_mean": float(f32n[:, 34].mean())}
    del rm, rn, outm, f32n, tgtn
    # near-fp32 on the bf16 pipe: fp32 deformation nets + radiance nets with bf16 hi/lo operands (3 MFMAs per product)
    progress("bf16x3 leg")
    recx, outx, rx = measure(pkg, dev, size, "bf16x3", min(args.steps, 5), 1)
    r32 = HipRenderer(pkg, dev, size, "fp32")
    with torch.no_grad():
        f32a = r32.render(0, r32.R)
    dx = (outx - f32a).abs()
    coarse_cols = list(range(0, 17))
    result["bf16x3"] = {"value": recx["value"], "unit": "rays/s", "ms_per_step": recx["ms_per_step"],
                        "dtype": "f32 deformation nets + bf16 hi/lo radiance nets (3 MFMAs per product, f32 accumulate)",
                        "speedup_vs_fp32": recx["value"] / result["value"], "roofline": recx["roofline"],
                        "psnr_vs_fp32_db": psnr(rgb_fine(outx), rgb_fine(f32a)), "max_abs_diff_coarse_outputs": float(dx[:, coarse_cols].max()),
                        "rays_within_4x_fp32_tolerance": float(((dx <= 4e-5 + 4e-4 * f32a.abs()).all(dim=1)).float().mean())}
    del rx, r32, outx, f32a, dx
    # the num_fine 128 reading (fine pass of 192 samples, 256 evaluations per ray)
    progress("num_fine128 leg")
    rec, _, _ = measure(pkg, dev, size, "fp32", 2, 1, num_fine=128)
    result["num_fine128"] = {"value": rec["value"], "unit": "rays/s", "ms_per_step": rec["ms_per_step"], "dtype": "f32",
                             "workload": rec["config"]["workload"], "roofline": rec["roofline"]}
    torch.cuda.empty_cache()
    progress("train_T2048 leg")
    result["train_T2048"] = train_leg(pkg, dev)
    progress("driver seam frame")
    # one timed frame through the drop-in driver seam itself (run_one_iter_of_nerf, keyed draws) next to the instrumented chain
    r = HipRenderer(pkg, dev, size, "fp32")
    ro, rd = pkg.get_ray_bundle(r.H, r.W, r.intr, r.pose)
    with torch.no_grad(), pkg.train_utils.partition_invariant_rng(r.seed):
        f = lambda: pkg.run_one_iter_of_nerf(r.H, r.W, r.intr, r.model, ro, rd, r.cfg, mode="validation", driving=r.audio, pose=r.pose,
                                             background_prior=r.bg_all)
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        result["driver_seam_ms_per_frame"] = (time.perf_counter() - t0) * 1e3


def train_leg(pkg, dev, rays=2048, steps=5, warmup=2):
    """BASELINE.json configs[4]: 2048 semantically-weighted rays, train mode (noise 0.1), forward + backward through the HIP autograd
    op and the reference's loss recipe (train_stage_rays_auto.py:437-499); no optimiser step (not part of the path)."""
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

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(loss))
    tflops = rays * 192 * FLOP_PER_SAMPLE["audio"] * 3 / dt / 1e12
    return {"workload": "T%d: %d semantically-weighted rays, 64+128 evaluations/ray, train mode (noise 0.1), forward + loss recipe + backward"
                        % (rays, rays), "ms_per_step": dt * 1e3, "value": rays / dt, "unit": "rays/s", "dtype": "f32", "steps": steps,
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s", "frac": tflops / PEAK_TFLOPS["fp32"],
                         "flop_rule": "3 x forward GEMM FLOPs per training ray (SURVEY.md section 8d)"}}


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
    with torch.no_grad():
        run_gpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eager = run_gpu()
        torch.cuda.synchronize()
        gdt = time.perf_counter() - t0
    names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    hip = pkg.train_utils.unpack_rows(out)
    result["torch_gpu_baseline"] = {"value": R / gdt, "unit": "rays/s", "ms_per_frame": gdt * 1e3, "dtype": "f32",
                                    "what": "plain PyTorch-ROCm eager restatement of the reference (same op sequence, chunksize 131072) on the same "
                                            "GPU, frame, weights and draws; one warm frame then one timed",
                                    "speedup_fp32": result["value"] / (R / gdt),
                                    "psnr_hip_vs_eager_db": psnr(hip[3][:, :3], eager[3][:, :3]),
                                    "max_abs_diff": {n: float((a.reshape(b.shape) - b).abs().max()) for n, a, b in zip(names, hip, eager)}}
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
        d = (cpu(a.reshape(R, -1)[idx]) - b.reshape(cs * cs, -1)).abs().max(dim=1).values
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="frame is size x size rays")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="fp32 = configs[1] (exact, headline); bf16 = configs[2] (bf16 MFMA operands, fp32 accumulate)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the torch_gpu_baseline / cpu_baseline legs")
    ap.add_argument("--no-secondary", action="store_true", help="headline only (no bf16 / NeRFace / num_fine128 / training legs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, "launch with torchrun --nproc-per-node == --gpus"

    pkg = importlib.import_module("sahs-deformable-nerf_amd")
    t_start = time.perf_counter()
    progress("headline")
    result, out, rend = measure(pkg, dev, args.size, args.precision, args.steps, args.warmup, world=world, rank=rank, dist=dist)
    if args.precision == "fp32" and world == 1:
        if not args.no_secondary:
            add_secondary_legs(result, pkg, dev, args)
        if not args.no_cpu_baseline:
            add_baselines(result, out, rend, pkg, dev)
    progress("done")
    result["bench_wall_s"] = time.perf_counter() - t_start
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
