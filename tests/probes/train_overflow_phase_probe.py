"""Where do the f16x3 range fallbacks of a long training run come from - the forward pass (an activation beyond the H2 range at the
fixed activation exponent) or the backward pass (a gradient at the step's power-of-two scale)?  300 steps of BASELINE config 3 on one
repeated synthetic batch; the overflow word is read after the forward pass and after the backward pass of every f16x3 attempt.
usage: python tests/probes/train_overflow_phase_probe.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import synth, training as T
from sfh_amd.reconstructor import Reconstructor

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
ts = T.TrainStep(net, lr=1e-5, weight_decay=1e-8, seg_lambda=1.0, rec_lambda=1.0, reproj_lambda=1.0, consist_lambda=1.0)
events = []
real_bwd = T.run_backward
state = {"step": 0}
def spy(net_, tape, f, dheads, dtheta, unscale=True):
    fwd = int(tape.overflow.item()) if tape.overflow is not None else 0
    if fwd:
        events.append((state["step"], "forward", tape.fmt))
    try:
        return real_bwd(net_, tape, f, dheads, dtheta, unscale)
    except T.FP16RangeError:
        if not fwd:
            events.append((state["step"], "backward", tape.fmt, f"gscale 2^{int(torch.log2(torch.tensor(float(tape.gscale))).item())}"))
        raise
T.run_backward = spy
for i in range(steps):
    state["step"] = i
    losses = ts.step(x, batch)
torch.cuda.synchronize()
print("steps", steps, "fallbacks", ts.range_fallbacks, "final losses", [round(float(v), 4) for v in losses])
for e in events:
    print(e)
