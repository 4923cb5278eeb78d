"""CPU ORACLE - test infrastructure only, never part of the product path.

A PyTorch-CPU fp32 restatement of the reference's hot path, written as pure functions
of a flat ``state_dict`` (the reference's checkpoint layout) so that it shares no code
with the product (``sports-field-homography_amd/``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Each function cites the reference lines it follows (paths relative to /root/reference).

Pinning status
--------------
* UNet blocks and ResNetSTN (A1-A6): pinned - ``oracle/make_fixtures.py`` runs the
  reference's own ``unet/unet_parts.py`` and ``models/resnet.py`` classes (imported by
  file path) on seeded weights and commits their outputs under ``tests/golden/``;
  ``tests/test_oracle.py`` checks this restatement against them.
* Homography warp / POI projection (A7, A8): the arithmetic lives in the third-party
  dependency Kornia (requirements.txt:1, ``kornia>=0.5.0``), which is neither vendored
  in the reference nor installed here, and the reference has no test or golden vector
  for it -> **parity unpinned** with respect to Kornia itself.  The restatement follows
  the published Kornia 0.5/0.6 algorithm (create_meshgrid -> transform_points ->
  convert_points_from_homogeneous -> F.grid_sample); the sampling primitive
  ``torch.nn.functional.grid_sample`` IS available and ``oracle/warp_ref.py`` is checked
  against it.
"""
import torch
import torch.nn.functional as F

from . import warp_ref

# utils/postprocess.py:10-11 labels a pixel with argmax(softmax(logits)), not argmax(logits).  fp32 softmax is only weakly
# monotone: exp(b - a) rounds to 1.0 when the top-2 margin a - b is at most 2^-25 = 3.0e-8 (measured on torch 2.10 CPU:
# every such pair ties, no pair at 6e-8 or beyond does), the two probabilities are then EQUAL and argmax returns the lower
# index.  Distinct fp32 logits can be that close only below magnitude 0.5 (ulp(0.5) = 6e-8).  1.2e-7 = 2^-23 is the
# conservative band (one ulp of 1.0) inside which an argmax(logits) implementation may differ from the reference's rule.
SOFTMAX_TIE_MARGIN = 1.2e-7


def softmax_argmax_may_differ(logits):
    """bool (B,H,W): pixels whose top-2 logit margin is inside SOFTMAX_TIE_MARGIN - the only ones where
    argmax(softmax(logits)) (the reference) and argmax(logits) (the HIP epilogue) can disagree"""
    top2 = logits.topk(2, dim=1).values
    return (top2[:, 0] - top2[:, 1]) < SOFTMAX_TIE_MARGIN


BN_EPS = 1e-5  # nn.BatchNorm2d default, used unchanged by unet/unet_parts.py:16,19


BN_TRAINING = False  # set by oracle/train_ref.py: batch statistics + in-place running-stat update


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], BN_TRAINING, 0.1, BN_EPS)


def double_conv(x, sd, p):
    """unet/unet_parts.py:14-24 - (conv3x3 pad1 + bias, BN(eval), ReLU) twice."""
    q = p + ".double_conv"
    x = F.relu(_bn(F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], padding=1), sd, q + ".1"))
    x = F.relu(_bn(F.conv2d(x, sd[q + ".3.weight"], sd[q + ".3.bias"], padding=1), sd, q + ".4"))
    return x


def down(x, sd, p):
    """unet/unet_parts.py:32-38 - MaxPool2d(2) (floor) then DoubleConv."""
    return double_conv(F.max_pool2d(x, 2), sd, p + ".maxpool_conv.1")


def up(x1, x2, sd, p, bilinear=False):
    """unet/unet_parts.py:56-68 - upsample x1, pad to x2's size, cat([x2, x1]), DoubleConv."""
    if bilinear:
        x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        x1 = F.conv_transpose2d(x1, sd[p + ".up.weight"], sd[p + ".up.bias"], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(torch.cat([x2, x1], dim=1), sd, p + ".conv")


def out_conv(x, sd, p):
    """unet/unet_parts.py:74-77 - 1x1 conv + bias."""
    return F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"])


def forward_unet(x, sd, unet_size=(640, 360), target_size=(640, 360), bilinear=False):
    """models/reconstructor.py:132-158."""
    if x.shape[3] != unet_size[0] or x.shape[2] != unet_size[1]:
        x = F.interpolate(x, size=(unet_size[1], unet_size[0]), mode="bilinear", align_corners=False)
    x1 = double_conv(x, sd, "inc")
    x2 = down(x1, sd, "down1")
    x3 = down(x2, sd, "down2")
    x4 = down(x3, sd, "down3")
    x_top = down(x4, sd, "down4")
    y = up(x_top, x4, sd, "up1", bilinear)
    y = up(y, x3, sd, "up2", bilinear)
    y = up(y, x2, sd, "up3", bilinear)
    y = up(y, x1, sd, "up4", bilinear)
    logits = out_conv(y, sd, "outc")
    uv = out_conv(y, sd, "outuv") if "outuv.conv.weight" in sd else None
    if logits.shape[3] != target_size[0] or logits.shape[2] != target_size[1]:
        logits = F.interpolate(logits, size=(target_size[1], target_size[0]), mode="nearest")
        if uv is not None:
            uv = F.interpolate(uv, size=(target_size[1], target_size[0]), mode="nearest")
    return logits, x_top, uv


def _basic_block(x, sd, p, stride):
    """models/resnet.py:64-82."""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1), sd, p + ".bn1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, padding=1), sd, p + ".bn2")
    if p + ".downsample.0.weight" in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride), sd, p + ".downsample.1")
    return F.relu(out + x)


def _bottleneck(x, sd, p, stride):
    """models/resnet.py:120-140 (stride on the 3x3 conv)."""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], None, stride=stride, padding=1), sd, p + ".bn2"))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"], None), sd, p + ".bn3")
    if p + ".downsample.0.weight" in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride), sd, p + ".downsample.1")
    return F.relu(out + x)


def resnet_stn(y, sd, p="resnet_reg", layers=(3, 4, 6, 3)):
    """models/resnet.py:235-254 - stem 7x7 s2, maxpool 3 s2 p1, 4 stages, avgpool, Linear -> (B,1,3,3)."""
    x = F.relu(_bn(F.conv2d(y, sd[p + ".conv0.weight"], None, stride=2, padding=3), sd, p + ".bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, n in enumerate(layers, start=1):
        for bi in range(n):
            stride = 2 if (li > 1 and bi == 0) else 1
            bp = f"{p}.layer{li}.{bi}"
            x = (_bottleneck if bp + ".conv3.weight" in sd else _basic_block)(x, sd, bp, stride)
    x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
    x = F.linear(x, sd[p + ".reg.weight"], sd[p + ".reg.bias"])
    return x.view(-1, 1, 3, 3)


def warp(theta, court_img, warp_size=(640, 360), nearest=False):
    """models/reconstructor.py:109-118 -> Kornia HomographyWarper (restated in warp_ref)."""
    bs = theta.shape[0]
    return warp_ref.homography_warp(theta, court_img[0:bs], warp_size[1], warp_size[0],
                                    "nearest" if nearest else "bilinear")


def transform_poi(theta, court_poi, normalize=True):
    """models/reconstructor.py:120-130."""
    bs = theta.shape[0]
    poi = warp_ref.transform_points(torch.inverse(theta[:bs]), court_poi[:bs])
    return poi / 2.0 + 0.5 if normalize else poi


def predict(x, sd, court_img, court_poi, mask_classes=4, warp_size=(640, 360),
            unet_size=(640, 360), target_size=(640, 360), consistency=True, project_poi=False,
            use_warper=True, warp_with_nearest=True, layers=(3, 4, 6, 3)):
    """models/reconstructor.py:196-247 for resnet_input='img+mask'."""
    ret = {}
    ret["logits"], _, _ = forward_unet(x, sd, unet_size, target_size)
    theta = resnet_stn(torch.cat((ret["logits"], x), 1), sd, layers=layers)
    ret["theta"] = theta
    if use_warper:
        wm = warp(theta, court_img, warp_size, warp_with_nearest) * mask_classes
        if consistency:
            logits = ret["logits"]
            m = wm
            if logits.shape[2:4] != m.shape[1:3]:
                m = F.interpolate(m.unsqueeze(1), size=logits.shape[2:4], mode="nearest").squeeze(1)
            scores = F.cross_entropy(logits, m.type(torch.int64), reduction="none")
            ret["consist_score"] = torch.mean(scores, dim=(1, 2))
        ret["warp_mask"] = wm.type(torch.int32)
    if project_poi:
        ret["poi"] = transform_poi(theta, court_poi)
    return ret


def stn_input(x, logits, uv, resnet_input="img+mask"):
    """models/reconstructor.py:173-183."""
    if resnet_input == "img":
        return x
    if resnet_input == "mask":
        return logits
    if resnet_input == "img+mask":
        return torch.cat((logits, x), 1)
    if resnet_input == "img+mask+uv":
        return torch.cat((logits, x, uv), 1)
    raise NotImplementedError(resnet_input)


def forward(x, sd, court_img, court_poi, warp_size=(640, 360), unet_size=(640, 360),
            target_size=(640, 360), use_warper=True, warp_with_nearest=False, layers=(3, 4, 6, 3),
            resnet_input="img+mask", use_resnet=True, bilinear=False):
    """models/reconstructor.py:160-194 (BatchNorm mode per BN_TRAINING)."""
    ret = {}
    ret["logits"], _, uv = forward_unet(x, sd, unet_size, target_size, bilinear=bilinear)
    if uv is not None:
        ret["uv"] = uv
    if not use_resnet:
        return ret
    theta = resnet_stn(stn_input(x, ret["logits"], uv, resnet_input), sd, layers=layers)
    ret["theta"] = theta
    ret["poi"] = transform_poi(theta, court_poi)
    if use_warper:
        ret["warp_mask"] = warp(theta, court_img, warp_size, warp_with_nearest)
    return ret


def preds_to_masks(logits):
    """utils/postprocess.py:7-18 - softmax then argmax (class index per pixel) as uint8."""
    return torch.argmax(F.softmax(logits, dim=1), dim=1).to(torch.uint8)
