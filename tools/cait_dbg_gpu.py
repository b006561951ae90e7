import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden"); sys.path.insert(0, "/root/repo/tests")
from cases import CAIT_CASES, CAIT_PARAM_NAMES, make_cait_inputs
from efficient_probing_amd.poolings.cait import CAPooling
case = CAIT_CASES[0]
inp = make_cait_inputs(case)
pool = CAPooling(embed_dim=case.D).cuda()
with torch.no_grad():
    for n, p in zip(CAIT_PARAM_NAMES, pool._tensors()):
        p.copy_(torch.from_numpy(inp[n]))
x = torch.from_numpy(inp["x_buf"]).cuda()
torch.manual_seed(0)
dout = torch.randn(case.B, case.D)
out = pool(x)
(out * dout.cuda()).sum().backward()
np.savez("/root/repo/gpurun_out/cait_dbg.npz", out=out.detach().cpu().numpy(), dout=dout.numpy(),
         **{f"g_{n}": p.grad.cpu().numpy() for n, p in zip(CAIT_PARAM_NAMES, pool._tensors())})
print("saved")
