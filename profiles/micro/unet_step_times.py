"""Per-launch time of the UNet engine's launches IN SEQUENCE, timed with HIP events around every recorded step (no profiler):
do the encoder's deep layers (d2.3 .. d4.3) really run a third slower than the decoder's equal shapes, as the rocprofv3
--kernel-trace tables since round 5 say (profiles/r05b_layer_table.txt, r06_layer_table.txt), or is that the profiler?
usage: python profiles/micro/unet_step_times.py [passes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

B, W, H = 16, 640, 360
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net.to(dev).eval()
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
with torch.no_grad():
    for _ in range(3):
        net.predict(x, consistency=False)
    un, rn = net._get_engines()
    steps = un.steps
    tot = [0.0] * len(steps)
    for p in range(passes):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(steps) + 1)]
        evs[0].record()
        for i, (_, fn) in enumerate(steps):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize()
        for i in range(len(steps)):
            tot[i] += evs[i].elapsed_time(evs[i + 1])
    for i, (outs, _) in enumerate(steps):
        print(f"{i:3d} {','.join(outs) or '-':28s} {tot[i] / passes * 1e3:8.1f} us")
    print(f"sum {sum(tot) / passes:.3f} ms")
