"""Stage-I training step around the HIP ops (BASELINE.json configs[4]; semantics of
``train_stage_rays_auto.py:390-509``, which is out of sync with ``nerf/`` as shipped -- SURVEY.md section 0.3 -- so this
mirrors what the step computes, not its tuple unpacking).

  1. semantic-weighted ray sampling: p(pixel) ~ sum_c sample_prob[c] * mask[pixel, c]            (:390-394, :417-419)
  2. 2048 rays without replacement; gather origins, directions, target colour, background prior, class mask (:421-432)
  3. run_one_iter_of_nerf(mode="train") -> coarse and fine 15-channel renders                               (:437-452)
  4. loss = sum over {coarse, fine} of  MSE + 0.02 * CE + 0.005 * (masked MSE + masked CE of classes 7:9)   (:455-465)
     [+ 0.005 * ||spatial_embeddings|| when regularize_spatial_embedding]                                  (:476-477, :490-491)
  5. sample_prob <- normalised sum of the four weighted per-class losses (feeds the next step's sampling)   (:466-468)
  6. backward, Adam step, lr = lr0 * decay_factor ** (i / (lr_decay * 1000))                               (:493-509)
"""
import torch

from .nerf_helpers import MaskCrossEntropyLoss, MaskMSELoss, get_ray_bundle, mse2psnr
from .train_utils import run_one_iter_of_nerf


def semantic_ray_probs(sample_prob, mask):
    """(12,), (H, W, 12) -> (H*W,) sampling distribution over pixels."""
    p = torch.sum(sample_prob.reshape(1, 1, -1) * mask, dim=-1).reshape(-1).double()
    return (p / p.sum()).float()


def sample_training_rays(probs, num_rays, generator=None):
    """Indices of ``num_rays`` distinct pixels drawn with probability ``probs`` (np.random.choice(replace=False, p=...))."""
    return torch.multinomial(probs, num_rays, replacement=False, generator=generator)


def sample_prob_weights(device=None, dtype=torch.float32):
    """train_stage_rays_auto.py:268-269: the per-class weights both loss modules are built with -- ones(12) with the mouth
    classes 7 and 8 doubled.  They do not enter the loss value, only the weighted per-class losses that become the next
    step's sampling distribution (:466-468)."""
    w = torch.ones(12, device=device, dtype=dtype)
    w[7:9] = 2
    return w


def stage1_loss(rgb_coarse, rgb_fine, target_rgb, mask, mse_loss=None, ce_loss=None):
    """train_stage_rays_auto.py:455-468 -> (loss, new_sample_prob, fine_mse).  rgb_*: (R,15) = [rgb3 | seg12]; target_rgb (R,3);
    mask (R,12) one-hot.  The loss modules default to the script's own (:270-271: weights ones(12), [7:9] = 2)."""
    if mse_loss is None:
        mse_loss = MaskMSELoss(sample_prob_weights(mask.device, rgb_coarse.dtype))
    if ce_loss is None:
        ce_loss = MaskCrossEntropyLoss(sample_prob_weights(mask.device, rgb_coarse.dtype))
    total, weighted = 0.0, []
    fine_mse = None
    for rgb in (rgb_coarse, rgb_fine):
        if rgb is None:
            continue
        l2, m_l2, m_l2_w = mse_loss(mask, rgb[..., :3], target_rgb[..., :3])
        ce, m_ce, m_ce_w = ce_loss(mask, rgb[..., 3:], mask)
        mouth = torch.sum(m_l2[7:9] + m_ce[7:9])
        total = total + l2 + 0.02 * ce + 0.005 * mouth
        weighted += [m_l2_w, m_ce_w]
        fine_mse = l2
    w = torch.stack([x.detach() for x in weighted]).sum(0)
    return total, w / w.sum(), fine_mse


def learning_rate(cfg, step):
    return cfg.optimizer.lr * (cfg.scheduler.lr_decay_factor ** (step / (cfg.scheduler.lr_decay * 1000)))


def train_step(model, optimizer, cfg, step, image, mask, pose, intrinsics, audio, background, sample_prob, generator=None,
               regularize_spatial_embedding=False, group=None, fused_loss=True):
    """One optimisation step on one frame.  image (H,W,3), mask (H,W,12) one-hot float, background (H,W,15).
    Returns dict(loss, psnr, sample_prob).

    Under torch.distributed (one process per GPU) the step is data-parallel over rays: every rank draws the same batch
    (``generator`` seeded identically), renders and back-propagates its contiguous slice, and the gradients -- and the
    per-class sampling feedback -- are averaged with one all-reduce each before the optimiser step, so the replicas stay
    identical.

    fused_loss: the objective is evaluated by one HIP launch on the rendered maps and its gradient is formed inside the composite
    backward kernels (ops.RenderRaysFn with loss operands) instead of ~40 small torch kernels and an (R,15) gradient round trip;
    same recipe, same class weights (stage1_loss is the unfused statement, kept for the seams and the tests)."""
    import torch.distributed as dist
    from . import distributed as D
    H, W = image.shape[:2]
    probs = semantic_ray_probs(sample_prob, mask)
    sel = sample_training_rays(probs, int(cfg.nerf.train.num_random_rays), generator)
    sel = sel[D.shard_batch(sel.shape[0], group)]
    ro, rd = get_ray_bundle(H, W, intrinsics, pose)
    ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
    target = image.reshape(-1, image.shape[-1])[sel]
    bg = background.reshape(-1, 15)[sel] if background is not None else None
    m = mask.reshape(-1, 12)[sel].float()
    if fused_loss and m.is_cuda and sel.shape[0] <= int(cfg.nerf.train.chunksize):
        outs = run_one_iter_of_nerf(H, W, intrinsics, model, ro, rd, cfg, mode="train", driving=audio, pose=pose, background_prior=bg, inHead=m,
                                    _loss=(target[..., :3].float(), m, sample_prob_weights(m.device)))
        loss, stats = outs[8], outs[9]
        new_prob, fine_mse = stats[2:14].clone(), stats[1].clone()
    else:
        outs = run_one_iter_of_nerf(H, W, intrinsics, model, ro, rd, cfg, mode="train", driving=audio, pose=pose, background_prior=bg, inHead=m)
        loss, new_prob, fine_mse = stage1_loss(outs[0], outs[3], target, m)
    if regularize_spatial_embedding:
        loss = loss + torch.norm(model.spatial_embeddings) * 0.0005 * 10
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        D.all_reduce_gradients(model.parameters(), group)
        stats = torch.cat([new_prob, loss.detach().reshape(1), fine_mse.detach().reshape(1)])
        dist.all_reduce(stats, group=group)
        stats /= dist.get_world_size(group)
        new_prob, loss, fine_mse = stats[:-2] / stats[:-2].sum(), stats[-2], stats[-1]
    optimizer.step()
    for param_group in optimizer.param_groups:
        param_group["lr"] = learning_rate(cfg, step)
    return dict(loss=float(loss.detach()), psnr=mse2psnr(float(fine_mse.detach())), sample_prob=new_prob)


def save_checkpoint(path, step, model, optimizer, loss, i_batch=0, background=None, latent_codes=None, pose_c=None, sample_prob=None):
    """The checkpoint dict of train_stage_rays_auto.py:698-722 (same keys), so files written here load in the reference and
    vice versa (its eval script reads model_state_dict / background / pose_c, eval_stage_rays.py:302-327)."""
    torch.save({"iter": int(step), "i_batch": int(i_batch), "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict(), "loss": float(loss),
                "background": None if background is None else background.detach().clone(),
                "latent_codes": None if latent_codes is None else latent_codes.detach().clone(),
                "pose_c": pose_c, "sample_prob": None if sample_prob is None else sample_prob.detach().clone()}, path)


def resume(path, model, optimizer, device):
    """train_stage_rays_auto.py:238-258: -> dict(start_iter, i_batch, background, latent_codes, sample_prob).  weights_only load."""
    ck = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ck["model_state_dict"])
    optimizer.load_state_dict(ck["optimizer_state_dict"])
    return dict(start_iter=int(ck["iter"]) + 1, i_batch=int(ck.get("i_batch", 0)), background=ck.get("background"),
                latent_codes=ck.get("latent_codes"), sample_prob=ck.get("sample_prob"), pose_c=ck.get("pose_c"))
