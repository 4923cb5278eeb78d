"""Range of the pre-activation gradients dz of one training step (C3 shape, B given): what a two-plane fp16 copy of
dz would have to hold.  Prints per layer max|dz| and the 1 % quantile of |dz| over non-zero elements, and the same
for the weights of the backward-data convs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sfh_amd import synth, training  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net.to(dev).train()
g = torch.Generator().manual_seed(0)
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
batch = {"mask": torch.randint(0, 4, (B, H, W), generator=g).to(dev), "weight": torch.ones(B, device=dev),
         "poi": torch.rand(B, poi.shape[1], 2, generator=g).to(dev), "nonzeros": torch.ones(B, poi.shape[1], device=dev)}
batch["num_nonzero"] = batch["nonzeros"].sum(1)
ts = training.TrainStep(net, lr=1e-5, weight_decay=1e-8, seg_lambda=1.0, rec_lambda=1.0, reproj_lambda=1.0, consist_lambda=1.0)

rows = []
orig = training._bn_backward


def spy(lib, tape, dy, y, z, mi, bn, relu, want_dres, want_s3=False):
    out = orig(lib, tape, dy, y, z, mi, bn, relu, want_dres, want_s3)
    dz = out[0]
    a = dz.abs().flatten()
    nz = a[a > 0]
    q = torch.quantile(nz[:: max(1, nz.numel() // 1000000)].float(), torch.tensor([0.01, 0.5], device=a.device)) if nz.numel() else torch.zeros(2)
    rows.append((tuple(dz.shape), a.max().item(), q[0].item(), q[1].item(), dy.abs().max().item()))
    return out


training._bn_backward = spy
ts.step(x, batch)
torch.cuda.synchronize()
print(f"B={B}: 1/(B*H*W) = {1.0 / (B * H * W):.3e}")
for shp, mx, q1, q50, dymax in rows:
    print(f"dz {str(shp):24s} max {mx:.3e}  median {q50:.3e}  1% {q1:.3e}  ratio max/1% {mx / max(q1, 1e-300):.1e}  max|dy| {dymax:.3e}")
mxs = [r[1] for r in rows]
print(f"max over layers {max(mxs):.3e}, min over layers of the layer max {min(mxs):.3e}, spread 2^{torch.log2(torch.tensor(max(mxs) / min(mxs))).item():.1f}")
