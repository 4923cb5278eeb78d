"""Does a freshly initialised model (torch default init, BatchNorm statistics at their defaults or after a few
training steps) stay inside the fp16 range of the f16x3 mode?  Prints range_fallbacks and the distance between the
f16x3 and bf16x6 results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

B, H, W = 4, 360, 640
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=3)).to(dev)
for seed in (0, 1, 2):
    torch.manual_seed(seed)
    net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
    net._init_like_reference() if hasattr(net, "_init_like_reference") else None
    net.to(dev).eval()
    outs = {}
    for prec in ("f16x3", "bf16x6"):
        net.precision = prec
        with torch.no_grad():
            outs[prec] = net.predict(x, consistency=False)
    torch.cuda.synchronize()
    dl = (outs["f16x3"]["logits"] - outs["bf16x6"]["logits"]).abs().max().item()
    dt = (outs["f16x3"]["theta"] - outs["bf16x6"]["theta"]).abs().max().item()
    print(f"seed {seed}: range_fallbacks {net.range_fallbacks}  max|dlogits| {dl:.2e} (|logits| max {outs['bf16x6']['logits'].abs().max().item():.2e})  max|dtheta| {dt:.2e}")
