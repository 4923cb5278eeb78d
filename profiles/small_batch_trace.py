"""predict() at small batches under `rocprofv3 --kernel-trace`: is a batch of 1 / 2 / 8 frames bound by the host (enqueue) or
by the chain of dependent launches on the GPU?  Prints wall time per batch; profiles/small_batch_gaps.py reads the trace.
    B=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_sb1 -- python3 profiles/small_batch_trace.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

B, W, H = int(os.environ.get("B", "1")), 640, 360
N = int(os.environ.get("N", "20"))
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev)
poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
net.to(dev).eval()
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0)).to(dev)
mode = os.environ.get("MODE", "predict")
with torch.no_grad():
    fn = (lambda: net.predict(x, consistency=False)) if mode == "predict" else (lambda: net.predict_replay(x, consistency=False))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
        torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / N
print(f"B={B} mode={mode} {el * 1e3:.3f} ms per batch (each batch synchronised), {B / el:.1f} frames/s", flush=True)
