"""The single-kernel Up block against the two-launch form over a range of frame sizes (odd widths / heights at several levels,
sizes below one 16 x 32 tile, non-multiples of the tile): logits and theta must be identical bit for bit at every level set.
usage: python tests/probes/upfused_sizes_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sfh_amd  # noqa
from sfh_amd import synth
from sfh_amd.reconstructor import Reconstructor

bad = 0
for (w, h) in ((100, 60), (136, 90), (72, 56), (200, 120), (330, 180), (64, 48), (98, 74), (258, 130), (640, 368), (34, 34)):
    B = 2
    court = synth.load_court_template("ncaa_nc4_640x360", 4, B)
    court = torch.nn.functional.interpolate(court, size=(h, w), mode="nearest").contiguous().cuda()
    poi = synth.load_court_poi("pitch", B).cuda()
    net = Reconstructor(court, poi, target_size=(w, h), unet_size=(w, h), warp_size=(w, h), warp_with_nearest=True)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), 5))
    net.cuda().eval()
    x = synth.smooth_frames(B, h, w, seed=3).cuda()
    outs = {}
    for single in ((), (3, 4), (1, 2, 3, 4)):
        net.invalidate_engines()
        un, _ = net._get_engines()
        un.up_single = set(single)
        with torch.no_grad():
            o = net.predict(x, consistency=False)
        outs[single] = (o["logits"].clone(), o["theta"].clone())
    ok = all(torch.equal(outs[s][0], outs[()][0]) and torch.equal(outs[s][1], outs[()][1]) for s in ((3, 4), (1, 2, 3, 4)))
    fin = bool(torch.isfinite(outs[()][0]).all())
    print(f"{w}x{h}: {'identical' if ok else 'DIFFERENT'}; finite {fin}", flush=True)
    bad += (not ok) or (not fin)
print("mismatching sizes:", bad)
sys.exit(1 if bad else 0)
