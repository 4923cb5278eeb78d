"""predict() at 640x360 over batch sizes (latency of one batch, frames/s), default f16x3 arithmetic; with
SFH_SPLITK=0 the small-batch split-K of the ResNet layers is off.  GPU box only.
usage: python profiles/batch_sweep.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]
W, H = 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, max(sizes)).to(dev)
poi = synth.load_court_poi("pitch", max(sizes)).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net.to(dev).eval()
for B in sizes:
    x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
    with torch.no_grad():
        res = {}
        for name, fn in (("predict", net.predict), ("predict_replay", net.predict_replay)):
            for _ in range(4):
                fn(x, consistency=False)
            torch.cuda.synchronize()
            n = max(5, 160 // B)
            t0 = time.perf_counter()
            for _ in range(n):
                fn(x, consistency=False)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n
            # host time of a call with the GPU idle at entry: how long the caller's thread is held
            t0 = time.perf_counter()
            for _ in range(n):
                fn(x, consistency=False)
                held = time.perf_counter()
                torch.cuda.synchronize()
            res[name + "_sync_each"] = (time.perf_counter() - t0) / n
        # two batches in flight (predict_async): the ResNet-STN / warp of batch k under the UNet of batch k + 1
        def piped(m):
            prev = None
            for _ in range(m):
                h = net.predict_async(x, consistency=False)
                if prev is not None:
                    prev.result()
                prev = h
            prev.result()
        piped(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        piped(n)
        torch.cuda.synchronize()
        res["async"] = (time.perf_counter() - t0) / n
        net.pipeline_splitk = False           # the ResNet-STN launches unsplit beside the other batch's UNet (opt-in: other bits)
        piped(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        piped(n)
        torch.cuda.synchronize()
        res["async_nosplit"] = (time.perf_counter() - t0) / n
        net.pipeline_splitk = True
    print(f"B={B:3d}  predict_async() {B / res['async']:8.1f} frames/s {res['async'] * 1e3:7.3f} ms per batch   "
          f"(pipeline_splitk=False: {B / res['async_nosplit']:8.1f} frames/s {res['async_nosplit'] * 1e3:7.3f} ms)   "
          f"predict() {B / res['predict']:8.1f} frames/s {res['predict'] * 1e3:7.3f} ms per batch   "
          f"predict_replay() {B / res['predict_replay']:8.1f} frames/s {res['predict_replay'] * 1e3:7.3f} ms per batch   "
          f"(one batch at a time, synchronised: {res['predict_sync_each'] * 1e3:.3f} / {res['predict_replay_sync_each'] * 1e3:.3f} ms)", flush=True)
