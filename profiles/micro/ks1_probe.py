import sys, torch
sys.path.insert(0, '/root/repo')
from sfh_amd import engine as E
torch.manual_seed(0)
for cin in (32, 64, 128):
  for tile in (0,1,2):
    w = torch.randn(64, cin, 1, 1, device='cuda')
    x = torch.randn(1, 10, 9, cin, device='cuda')
    a = E.PackedConv(w, None, None, 1, cin, relu=False, s3=True)
    b = E.PackedConv(w, None, None, 1, cin, relu=False)
    ya = torch.empty(1,10,9,64, device='cuda'); yb = torch.empty_like(ya)
    a.run(E.f32_to_s3(x), 1, 10, 9, ya, tile=tile); b.run(x, 1, 10, 9, yb, tile=tile)
    torch.cuda.synchronize()
    ref = torch.einsum('bhwc,oc->bhwo', x.double(), w[:, :, 0, 0].double())
    print(cin, tile, 's3 err', (ya.double()-ref).abs().max().item(), 'fp32 err', (yb.double()-ref).abs().max().item())
