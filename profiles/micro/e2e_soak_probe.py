import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from sfh_amd import synth, engine as E, outputs as O
from sfh_amd.pipeline import FramePipeline
from sfh_amd.reconstructor import Reconstructor
B, W, H = 16, 640, 360
dev = torch.device("cuda", 0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B).to(dev); poi = synth.load_court_poi("pitch", B).to(dev)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H), warp_with_nearest=True)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), 0)); net.to(dev).eval()
N = 400
host = [torch.from_numpy(synth.synth_frames_u8(B, H, W, seed=900 + k)).pin_memory() for k in range(7)]
req = ("theta", "warp_mask", "segm_mask", "poi")
pipe = FramePipeline(net, B, (H, W), req_outputs=req, consistency=True)
want = []
with torch.no_grad():
    for k in range(7):
        want.append(O.transfer_gpu_to_cpu(net.predict(E.frames_u8_to_input(host[k].cuda()), consistency=True, project_poi=True), set(req), 4))
    t0 = time.perf_counter()
    bad = 0
    for i, res in enumerate(pipe.run(host[k % 7] for k in range(N))):
        w = want[i % 7]
        for key in res:
            if not np.array_equal(res[key], w[key]):
                bad += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(f"{N} batches through FramePipeline (7 distinct batches cycling), every output compared with predict(): {bad} mismatching arrays; "
      f"{B * N / el:.1f} frames/s incl. the host-side comparisons; range raises {net.range_raises} rescales {net.range_rescales}")
