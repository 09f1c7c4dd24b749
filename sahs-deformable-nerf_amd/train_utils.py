"""Drop-in for the render driver ``nerf/train_utils.py`` (seam B1 of SURVEY.md section 8b).

``run_one_iter_of_nerf`` has the reference's signature and return convention
(train_utils.py:209-321).  Differences are internal: points, encodings and hidden activations
are never materialised -- each ray chunk is six kernel launches (depths, coarse field,
composite, resample+sort, fine field, composite) -- and the per-frame conditioning is
computed once per call instead of once per 131072-point chunk.  Random draws are made with
torch.rand/torch.randn in the reference's order and shapes per ray chunk, so a
seeded generator on the same device gives the same stream.
"""
import contextlib

import torch

from . import ops
from .nerf_helpers import get_minibatches

_RAY_RNG = None   # None: torch.rand (the reference's stream) | (seed, first global ray index of the rays passed in)


@contextlib.contextmanager
def partition_invariant_rng(seed, ray_offset=0):
    """Inside the block the depth-perturbation and importance-sampling draws come from ``ops.ray_uniforms`` -- a pure function of
    (seed, global ray index, sample) -- instead of ``torch.rand`` over the chunk (train_utils.py:110, nerf_helpers.py:469), so
    a frame rendered in other chunk sizes or sharded over N GPUs (``ray_offset`` = first ray of this rank's block) is
    bit-identical.  The radiance noise of training mode stays on torch.randn."""
    global _RAY_RNG
    prev, _RAY_RNG = _RAY_RNG, (int(seed), int(ray_offset))
    try:
        yield
    finally:
        _RAY_RNG = prev


def unpack_rows(rows):
    """(n, 36) row block (ops.render_rays_rows) -> the reference's 8-tuple as column views (flat shapes)."""
    return [rows[:, 0:15], rows[:, 15], rows[:, 16], rows[:, 17:32], rows[:, 32], rows[:, 33], rows[:, 34], rows[:, 35]]


def run_network(level, network_fn, pts, ray_batch, chunksize, use_viewdirs, driving=None, pose=None, pose_c=None,
                latent_code=None, spatial_embeddings=None):
    """train_utils.py:9-50: evaluate the field at explicit points pts (N,S,3) -> (N,S,16)."""
    if not use_viewdirs:
        raise NotImplementedError("the shipped configs set nerf.use_viewdirs: True")
    flat = pts.reshape((-1, pts.shape[-1]))
    dirs = ray_batch[..., None, 3:6].expand(pts[..., :3].shape).reshape((-1, 3))
    x = torch.cat((flat, dirs), dim=-1)
    out = network_fn(level, x, driving, pose, pose_c, latent_code=latent_code)   # no python chunk loop: the kernel is grid-stride
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


def predict_and_render_radiance(ray_batch, model, options, mode="train", driving=None, pose=None, pose_c=None,
                                background_prior=None, latent_code=None, spatial_embeddings=None, ray_dirs_fake=None,
                                _frame=None, _workspace=None, _ray0=0, _rows=None, _loss=None):
    """train_utils.py:72-206 for one ray chunk -> the 8-tuple
    (rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine, weights_fine[:, -1], depth_fine).
    _rows (driver-internal): an (N, 36) row block to fill in place instead (ops.render_rays_rows); returns None then."""
    if latent_code is not None:
        raise NotImplementedError("latent codes are not used by the shipped audio configs")
    opt = getattr(options.nerf, mode)
    needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in model.parameters()) or (driving is not None and driving.requires_grad))
    rays = ray_batch.to(torch.float32)
    if not rays.is_contiguous():
        rays = rays.contiguous()
    N, dev = rays.shape[0], rays.device
    packed, _ = model.packed()
    frame = _frame if (_frame is not None or needs_grad) else model.frame(driving, pose)
    noise_std = float(opt.radiance_field_noise_std)
    nc, nf = int(opt.num_coarse), int(opt.num_fine)
    # draw order of the reference: rand(N,nc) [perturb] -> randn(N,nc) [noise] -> rand(N,nf) [perturb] -> randn(N,nc+nf) [noise]
    if _RAY_RNG is None:
        uniform = lambda stream, n: torch.rand((N, n), dtype=torch.float32, device=dev)
    else:
        uniform = lambda stream, n: ops.ray_uniforms(_RAY_RNG[0], stream, _RAY_RNG[1] + _ray0, N, n, dev)
    t_rand = uniform(0, nc) if opt.perturb else None
    noise_c = torch.randn((N, nc), dtype=torch.float32, device=dev) * noise_std if noise_std > 0.0 else None
    u = uniform(1, nf) if (nf > 0 and opt.perturb != 0.0) else None
    noise_f = torch.randn((N, nc + nf), dtype=torch.float32, device=dev) * noise_std if (nf > 0 and noise_std > 0.0) else None
    bg = None
    if background_prior is not None:
        if background_prior.shape[-1] != 15:
            raise NotImplementedError("background_prior must have 15 channels (rgb3 + seg12)")
        bg = background_prior.to(torch.float32)
    arch = getattr(model, "arch", "audio")
    if needs_grad:
        # training step (train_stage_rays_auto.py:437-499): differentiable op, fp32; its backward runs the HIP backward kernels
        if model.precision != ops.SAHS_F32:
            raise NotImplementedError("training runs the fp32 path; build the model with precision='fp32'")
        flat = model.flat_params(differentiable=True)
        # _loss = (target (N,>=3), mask (N,12), class weights (12,)): the op also returns (loss, stats) of the Stage-I objective and
        # forms that loss's gradient inside its composite backward (ops.RenderRaysFn)
        # ops.training_forward_precision("bf16x3"): the saving forward of a batch that keeps its activations runs on the split-operand kernels
        packed_x3 = None
        if ops.training_forward_precision() == "bf16x3" and arch == "audio" and nf > 0 and N <= ops.RenderRaysFn.BLOCK_RAYS:
            packed_x3, _ = model.packed(ops.SAHS_BF16X3)
        return ops.RenderRaysFn.apply(flat, driving.to(torch.float32), pose.to(torch.float32), rays.detach(), bg, t_rand, noise_c, u, noise_f,
                                      packed, nc, nf, bool(opt.lindisp), bool(opt.white_background), arch, *(_loss or (None, None, None)), packed_x3)
    if _rows is not None:
        ops.render_rays_rows(packed, frame, rays, nc, nf, _rows, precision=model.precision, lindisp=bool(opt.lindisp),
                             white_background=bool(opt.white_background), bg=bg, t_rand=t_rand, noise_c=noise_c, u=u, noise_f=noise_f,
                             workspace=_workspace, arch=arch)
        return None
    return ops.render_rays(packed, frame, rays, nc, nf, precision=model.precision, lindisp=bool(opt.lindisp),
                           white_background=bool(opt.white_background), bg=bg, t_rand=t_rand, noise_c=noise_c, u=u, noise_f=noise_f,
                           workspace=_workspace, arch=arch)


def run_one_iter_of_nerf(height, width, focal_length, model, ray_origins, ray_directions, options, mode="train", driving=None,
                         pose=None, pose_c=None, background_prior=None, latent_code=None, ray_directions_ablation=None,
                         spatial_embeddings=None, inHead=None, _loss=None, _shard=None):
    """train_utils.py:209-321.  height/width/focal_length are unused when dataset.no_ndc is True (as there).
    _loss (not in the reference): (target, mask, class_weights) of the Stage-I objective for a training batch that fits one ray
    chunk -> the 8-tuple is followed by (loss, stats), and the loss's gradient is formed inside the HIP backward
    (training.train_step uses it; stats layout: ops.LOSS_STATS_WORDS).
    _shard (not in the reference; no-grad calls): True or a torch.distributed process group -- every rank is handed the SAME full frame and
    renders only its contiguous block of the flattened ray list into its rows of the (R, 36) buffer, one in-place all-gather (RCCL on
    GPUs) completes the buffer on every rank (distributed.render_rows_sharded), and every rank returns the full frame.  Draws are then
    keyed by global ray index (partition_invariant_rng; entered with options.experiment.randomseed unless the caller already did), so
    the frame is bit-identical for any number of ranks.  Without an initialised process group it is the single-process render."""
    if options.dataset.no_ndc is False:
        raise NotImplementedError("NDC rays: the reference's own no_ndc=False branch is dead (NameError at train_utils.py:263)")
    ro = ray_origins.reshape((-1, 3))
    rd = ray_directions.reshape((-1, 3))
    near = options.dataset.near * torch.ones_like(rd[..., :1])
    far = options.dataset.far * torch.ones_like(rd[..., :1])
    parts = [ro, rd, near, far]
    if inHead is not None:
        parts.append(inHead.reshape((-1, inHead.shape[-1])).to(ro.dtype))   # carried for layout parity; the model never reads it (models.py:516)
    rays = torch.cat(parts, dim=-1)
    chunk = int(getattr(options.nerf, mode).chunksize)
    batches = get_minibatches(rays, chunksize=chunk)
    bgs = get_minibatches(background_prior, chunksize=chunk) if background_prior is not None else None
    needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in model.parameters()) or (driving is not None and driving.requires_grad))
    # once per call (the reference recomputes it per point-chunk); the differentiable op folds the conditioning itself
    frame = None if needs_grad else model.frame(driving, pose)
    workspace = {}
    if _shard is not None and _shard is not False and needs_grad:
        raise ValueError("_shard is the inference path's ray sharding; a differentiable call shards its batch (training.train_step)")
    if _loss is not None:
        if not needs_grad or len(batches) != 1:
            raise ValueError("_loss needs a differentiable call whose rays fit one chunk (nerf.%s.chunksize)" % mode)
        return tuple(predict_and_render_radiance(batches[0], model, options, mode, driving=driving, pose=pose, pose_c=pose_c,
                                                 background_prior=bgs[0] if bgs is not None else None, latent_code=latent_code,
                                                 _workspace=workspace, _loss=_loss))
    if needs_grad:
        pred = [predict_and_render_radiance(b, model, options, mode, driving=driving, pose=pose, pose_c=pose_c,
                                            background_prior=bgs[i] if bgs is not None else None, latent_code=latent_code,
                                            _frame=frame, _workspace=workspace if len(batches) == 1 or b.shape[0] == chunk else None,
                                            _ray0=i * chunk)
                for i, b in enumerate(batches)]
        images = [torch.cat(im, dim=0) if im[0] is not None else None for im in zip(*pred)]
    elif _shard is not None and _shard is not False:
        from . import distributed as D
        if float(getattr(options.nerf, mode).radiance_field_noise_std) > 0.0:
            raise NotImplementedError("ray sharding with radiance noise: the noise is torch.randn over a chunk (not keyed by ray); "
                                      "training shards the batch instead (training.train_step)")
        group = None if _shard is True else _shard

        def render_block(lo, hi, rows):      # this rank's rays [lo, hi) in chunks of nerf.<mode>.chunksize, written in place
            for s0 in range(lo, hi, chunk):
                e0 = min(hi, s0 + chunk)
                predict_and_render_radiance(rays[s0:e0], model, options, mode, driving=driving, pose=pose, pose_c=pose_c,
                                            background_prior=background_prior[s0:e0] if background_prior is not None else None,
                                            latent_code=latent_code, _frame=frame, _workspace=workspace if e0 - s0 == chunk or hi - lo <= chunk else None,
                                            _ray0=s0, _rows=rows[s0 - lo:e0 - lo])

        keyed = contextlib.nullcontext() if _RAY_RNG is not None else partition_invariant_rng(int(options.experiment.randomseed))
        with keyed:
            rows = D.render_rows_sharded(render_block, rays.shape[0], rays.device, group, ops.ROW_COLUMNS)
        images = [c.contiguous() for c in unpack_rows(rows)]
        if int(getattr(options.nerf, mode).num_fine) == 0:
            images[3] = images[4] = images[5] = None
    else:
        # every chunk writes its rays' 8-tuples straight into its row block of ONE (R, 36) buffer (the reference concatenates
        # eight lists of per-chunk tensors, train_utils.py:298-301)
        rows = torch.empty(rays.shape[0], ops.ROW_COLUMNS, dtype=torch.float32, device=rays.device)
        for i, b in enumerate(batches):
            s0 = i * chunk
            predict_and_render_radiance(b, model, options, mode, driving=driving, pose=pose, pose_c=pose_c,
                                        background_prior=bgs[i] if bgs is not None else None, latent_code=latent_code, _frame=frame,
                                        _workspace=workspace if len(batches) == 1 or b.shape[0] == chunk else None, _ray0=s0,
                                        _rows=rows[s0:s0 + b.shape[0]])
        images = [c.contiguous() for c in unpack_rows(rows)]
        if int(getattr(options.nerf, mode).num_fine) == 0:
            images[3] = images[4] = images[5] = None
    if mode == "validation":
        shape3, shape2 = tuple(ray_directions.shape), tuple(ray_directions.shape[:-1])
        shapes = [shape3, shape2, shape2]
        if hasattr(options.models, "fine"):
            shapes = shapes + shapes + [shape2, shape2]
        if options.models.mask.use_mask and images[0].shape[-1] == 15:
            shapes[0] = shape2 + (15,)
            if len(shapes) > 3:
                shapes[3] = shape2 + (15,)
        images = [im.reshape(s) if im is not None else None for im, s in zip(images, shapes)]
        if hasattr(options.models, "fine"):
            return tuple(images)
        return tuple(images + [None, None, None])
    return tuple(images)
