"""Every launch of the UNet pass (640x360 x 16, real activations of the synthetic checkpoint), timed (a) once in sequence, as
predict() issues them, and (b) repeated 20x in place (same buffers: inputs and weights hot).  A launch that is much slower in
sequence than repeated pays for its place in the pass (cold operands, the clock the previous launch left behind), not for its
own work.   usage: python profiles/micro/step_repeat_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import synth, engine as E
from sfh_amd.reconstructor import Reconstructor

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
B, W, H = 16, 640, 360
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net.to(dev).eval()
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
with torch.no_grad():
    for _ in range(3):
        net.predict(x)
    torch.cuda.synchronize()
    un, rn = net._get_engines()
    steps = un.steps
    names = [",".join(o) if o else "-" for o, _ in steps]
    # (a) in sequence, events around every launch, 5 passes
    seq = [0.0] * len(steps)
    for _ in range(5):
        evs = []
        with E._stream_scope():
            for _, fn in steps:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream()); fn(); e1.record(torch.cuda.current_stream())
                evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(evs):
            seq[i] += e0.elapsed_time(e1) / 5
    # (b) each launch repeated in place
    rep = []
    for _, fn in steps:
        with E._stream_scope():
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            for _ in range(reps):
                fn()
            e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        rep.append(e0.elapsed_time(e1) / reps)
    print(f"{'launch (H2 tensors written)':40s} {'in sequence ms':>15s} {'repeated ms':>12s} {'ratio':>6s}")
    for n, a, b in zip(names, seq, rep):
        print(f"{n:40s} {a:15.3f} {b:12.3f} {a / b:6.2f}")
    print(f"{'sum':40s} {sum(seq):15.3f} {sum(rep):12.3f} {sum(seq) / sum(rep):6.2f}")
