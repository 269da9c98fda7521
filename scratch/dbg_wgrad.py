import sys; sys.path.insert(0,'tests')
import conftest, torch, ops_harness as oh
from engine import hiplib
lib = hiplib.load()
for code in (hiplib.YH_F32, hiplib.YH_F16):
    dt = oh.tdtype(code)
    g = torch.Generator().manual_seed(1)
    N,H,W,cin,cout = 1,4,8,16,16
    x = torch.randn(N,H,W,cin,generator=g).to(dt); dz = torch.randn(N,H,W,cout,generator=g).to(dt)
    got = oh.wgrad(lib, code, x.cuda(), dz.cuda(), cin, cout, 1, 1, 0, splits=1).cpu()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0,3,1,2), (cout,cin,1,1), dz.float().permute(0,3,1,2))
    print('code', code, 'err', (got-ref).abs().max().item(), ref.abs().max().item())
    print(got[:4,:6,0,0]); print(ref[:4,:6,0,0])
    # one-hot probes: x = delta at pixel p, channel c ; dz = delta at pixel p, channel k -> dw[k][c] = 1
    for (p, c, k) in ((0,0,0),(1,2,3),(5,7,9),(17,3,1),(31,15,15)):
        x = torch.zeros(N,H,W,cin).to(dt); dz = torch.zeros(N,H,W,cout).to(dt)
        x.view(-1,cin)[p,c] = 1; dz.view(-1,cout)[p,k] = 1
        got = oh.wgrad(lib, code, x.cuda(), dz.cuda(), cin, cout, 1, 1, 0, splits=1).cpu()[:,:,0,0]
        nz = got.nonzero().tolist()
        print('probe', (p,c,k), '->', nz[:6], [got[i,j].item() for i,j in nz[:6]])
