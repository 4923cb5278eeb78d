"""CPU ORACLE for the training step (test infrastructure only; see oracle/torch_ref.py).

torch autograd over the functional restatement in oracle/torch_ref.py with BatchNorm in training
mode = what the reference computes for ``net.train(); preds = net(imgs); loss.backward()``
(train.py:151,170,233) and, for ``train_step``, the losses / clipping / RMSprop of train.py:88,
181-237 and models/losses.py:6-41.
"""
import contextlib

import torch
import torch.nn.functional as F

from . import torch_ref


@contextlib.contextmanager
def bn_training():
    old = torch_ref.BN_TRAINING
    torch_ref.BN_TRAINING = True
    try:
        yield
    finally:
        torch_ref.BN_TRAINING = old


def leaf_state(sd):
    """Clone a state_dict: float parameters become autograd leaves, buffers plain clones."""
    out = {}
    for k, v in sd.items():
        v = v.detach().clone()
        leaf = k.rsplit(".", 1)[-1]
        if v.is_floating_point() and leaf not in ("running_mean", "running_var"):
            v.requires_grad_(True)
        out[k] = v
    return out


def forward_unet_train(x, sd, **kw):
    """forward_unet (models/reconstructor.py:132-158) under net.train()."""
    with bn_training():
        return torch_ref.forward_unet(x, sd, **kw)


def forward_train(x, sd, court_img, court_poi, **kw):
    """Reconstructor.forward (models/reconstructor.py:160-194) under net.train()."""
    with bn_training():
        return torch_ref.forward(x, sd, court_img, court_poi, **kw)


def reprojection_loss(inputs, targets, nonzeros, num_nonzero):
    """models/losses.py:6-19 (reduction='mean')."""
    dist = torch.sqrt(torch.sum(torch.pow(targets - inputs, 2), dim=2))
    return torch.mean(torch.sum(dist * nonzeros, dim=1) / num_nonzero)


def per_sample_weighted(loss_map, weights):
    """models/losses.py:33-41."""
    return torch.mean(torch.mean(loss_map, dim=(1, 2)) * weights)


def losses(preds, batch, mask_classes=4, lambdas=(1.0, 1.0, 1.0, 1.0), consistency=True):
    """train.py:181-224 with seg_loss='CE', rec_loss='SmoothL1', reproj_loss='RRMSE', consist_loss='CE'."""
    seg_l, rec_l, reproj_l, cons_l = lambdas
    out = {}
    out["seg"] = per_sample_weighted(F.cross_entropy(preds["logits"], batch["mask"], reduction="none"),
                                     batch["weight"]) * seg_l
    gt_f = batch["mask"].to(torch.float32) / float(mask_classes)
    out["rec"] = per_sample_weighted(F.smooth_l1_loss(preds["warp_mask"], gt_f, reduction="none"),
                                     batch["weight"]) * rec_l
    out["reproj"] = reprojection_loss(preds["poi"], batch["poi"], batch["nonzeros"], batch["num_nonzero"]) * reproj_l
    if consistency:
        rec_int = (preds["warp_mask"] * mask_classes).to(torch.long)
        out["consist"] = F.cross_entropy(preds["logits"], rec_int) * cons_l
    out["total"] = sum(out.values())
    return out


def focal_loss(logits, target, alpha=1.0, gamma=2.0, eps=1e-8):
    """kornia.losses.FocalLoss(alpha=1.0, gamma=2.0, reduction='none') as train.py:101,126 builds it.
    Kornia is not in this image (parity unpinned, see oracle/torch_ref.py); this follows the published
    0.5/0.6 implementation: softmax + eps, a one-hot target that carries +1e-6 on every class,
    focal = -alpha * (1 - p)^gamma * log(p), summed over the class axis."""
    p = F.softmax(logits, dim=1) + eps
    onehot = F.one_hot(target, logits.shape[1]).permute(0, 3, 1, 2).to(logits.dtype) + 1e-6
    focal = -alpha * torch.pow(1.0 - p, gamma) * torch.log(p)
    return torch.sum(onehot * focal, dim=1)
