"""(test tooling: imports the CPU oracle) Loss trajectories of the same training run on the HIP TrainStep and
on the CPU restatement (oracle forward + losses + autograd + clip_grad_value_ + torch.optim.RMSprop): same
initial weights, same batch, train.py's defaults (lr 1e-4, weight decay 1e-8, momentum 0.9, lambdas 2/2/8/1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfh_amd import synth, training as T
from sfh_amd.reconstructor import Reconstructor
from oracle import train_ref

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H, W = 4, 96, 128
lam = (2.0, 2.0, 8.0, 1.0)
court = synth.load_court_template("ncaa_nc4_640x360", 4, B)[:, :, :H, :W].contiguous()
poi = synth.load_court_poi("pitch", B)
net = Reconstructor(court, poi, target_size=(W, H), unet_size=(W, H), warp_size=(W, H))
sd = synth.synth_state_dict(net.state_dict(), 61)
net.load_state_dict(sd)
x = synth.frames_to_float(synth.synth_frames_u8(B, H, W, seed=61))
g = torch.Generator().manual_seed(61)
batch = {"mask": torch.randint(0, 4, (B, H, W), generator=g), "weight": torch.rand(B, generator=g) + 0.5,
         "poi": torch.rand(B, poi.shape[1], 2, generator=g), "nonzeros": (torch.rand(B, poi.shape[1], generator=g) > 0.3).float()}
batch["num_nonzero"] = batch["nonzeros"].sum(1).clamp(min=1.0)

torch.set_num_threads(min(16, os.cpu_count() or 1))
ref = train_ref.leaf_state(sd)
params = [v for v in ref.values() if v.requires_grad]
opt = torch.optim.RMSprop(params, lr=1e-4, weight_decay=1e-8, momentum=0.9)
cpu = []
for it in range(steps):
    pr = train_ref.forward_train(x, ref, court, poi, warp_size=(W, H), unet_size=(W, H), target_size=(W, H))
    l = train_ref.losses(pr, batch, lambdas=lam)
    opt.zero_grad()
    l["total"].backward()
    torch.nn.utils.clip_grad_value_(params, 0.1)
    opt.step()
    cpu.append([l[k].item() for k in ("seg", "rec", "consist", "reproj")])

net.court_img, net.court_poi = court.cuda(), poi.cuda()
net.cuda().train()
ts = T.TrainStep(net, lr=1e-4, weight_decay=1e-8)
xb, bb = x.cuda(), {k: v.cuda() for k, v in batch.items()}
hip = [ts.step(xb, bb).cpu().tolist() for _ in range(steps)]
print("# step | HIP TrainStep: seg rec consist reproj total | CPU restatement: seg rec consist reproj total | rel. diff of total")
for it in range(steps):
    th, tc = sum(hip[it]), sum(cpu[it])
    print("%3d | %s %.5f | %s %.5f | %.2e" % (it, " ".join("%.5f" % v for v in hip[it]), th,
                                               " ".join("%.5f" % v for v in cpu[it]), tc, abs(th - tc) / tc))
