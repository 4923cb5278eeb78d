"""(test tooling: imports the CPU oracle) Gradient error of the HIP training path and of the fp32 CPU oracle, both against an fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfh_amd import synth, training as T
from sfh_amd.reconstructor import Reconstructor
from oracle import train_ref

H, W, B = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (38, 48, 2)))
net = Reconstructor(None, None, use_warper=False, use_resnet=False, target_size=(W, H), unet_size=(W, H))
sd = synth.synth_state_dict(net.state_dict(), 41)
net.load_state_dict(sd)
x = synth.smooth_frames(B, H, W, seed=41) if os.environ.get("SMOOTH") else synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=41))
g = torch.Generator().manual_seed(7)
dl = torch.randn(B, 4, H, W, generator=g) / (H * W)

def run(dtype):
    ref = train_ref.leaf_state({k: (v.to(dtype) if v.dtype == torch.float32 else v) for k, v in sd.items()})
    for k, v in ref.items():
        if v.dtype == dtype and not v.requires_grad and v.dim() > 0:
            pass
    lg, _, _ = train_ref.forward_unet_train(x.to(dtype), ref, unet_size=(W, H), target_size=(W, H))
    lg.backward(dl.to(dtype))
    return lg.detach(), {k: v.grad for k, v in ref.items() if v.requires_grad}

lg64, g64 = run(torch.float64)
lg32, g32 = run(torch.float32)
net.cuda().train()
tape = T.Tape()
u = T.UNetTrainer(net).forward(tape, x.cuda())
lg = u["logits"]
u["heads"][0][1](dl.cuda()); tape.backward(); torch.cuda.synchronize()
rel = lambda a, b: ((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30)).item()
print("logits: gpu %.2e cpu32 %.2e" % (rel(lg, lg64), rel(lg32, lg64)))
rows = [(k, rel(tape.param_grads[k], g64[k]), rel(g32[k], g64[k]), g64[k].abs().max().item()) for k in g64]
rows = [r for r in rows if r[3] > 1e-10]
rows.sort(key=lambda r: -r[1])
for r in rows[:12]:
    print("%-50s gpu %.2e cpu32 %.2e |g|max %.2e" % r)
