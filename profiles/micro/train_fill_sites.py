"""Which Python call sites of a training step (BASELINE config 3, TrainStep) issue torch fill / copy launches: torch.zeros,
zeros_like, zero_, fill_, copy_, clone, contiguous on non-contiguous, .to() - counted per (file:line) over ONE step.
usage: python profiles/micro/train_fill_sites.py"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import synth, training
from sfh_amd.reconstructor import Reconstructor

dev = torch.device("cuda", 0)
B, W, H = 16, 640, 360
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
for _ in range(2):
    ts.step(x, batch)
torch.cuda.synchronize()

from torch.utils._python_dispatch import TorchDispatchMode
sites = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        st = traceback.extract_stack()
        where = "?"
        for fr in reversed(st[:-1]):
            if "sports-field-homography_amd" in fr.filename or "sfh_amd" in fr.filename:
                where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                break
        sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    ts.step(x, batch)
torch.cuda.synchronize()
tot = collections.Counter()
for (n, w), c in sites.items():
    tot[n] += c
print("aten ops of one step:", sum(sites.values()))
for n, c in tot.most_common(25):
    print(f"{c:5d} {n}")
print()
for (n, w), c in sites.most_common(70):
    print(f"{c:5d} {n:40s} {w}")
