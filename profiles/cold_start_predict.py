"""Cold start of the drop-in path under `rocprofv3 --kernel-trace --stats`: build the model, load a checkpoint, move it
to the GPU, ONE predict() (weight packing, BatchNorm folding, exponent choice happen here) and two more (steady state).
Which kernels run that are not this library's?  (no at::native::* kernel on the predict path.)
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_cold -- python3 profiles/cold_start_predict.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sfh_amd import synth  # noqa: E402
from sfh_amd.reconstructor import Reconstructor  # noqa: E402

B, W, H = int(os.environ.get("B", "16")), 640, 360
mode = os.environ.get("INPUT", "img+mask")
court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
poi = synth.load_court_poi("pitch", B)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True,
                    resnet_input=mode, unet_uv=(mode == "img+mask+uv"))
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0))
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=0))
net.court_img, net.court_poi = court.cuda(), poi.cuda()
net.cuda().eval()
x = x.cuda()
torch.cuda.synchronize()
print("MARK cold predict", flush=True)
with torch.no_grad():
    for k in range(3):
        out = net.predict(x, consistency=True, project_poi=True)
        torch.cuda.synchronize()
        print("MARK predict", k, "done", flush=True)
print(float(abs(out["theta"].cpu().numpy()).sum()))      # (host arithmetic: no torch kernel of the probe's own in the trace)
