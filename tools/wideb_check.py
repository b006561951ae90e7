"""bf16-stored wide rows (D = 4096): the hybrid pass (csrc/ep_pool_wideb.hip, EP_POOL_WIDEB=1) against float64 on the stored values,
and its time at the benchmark shape.  Run once per EP_POOL_WIDEB setting (read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficient_probing_amd import functional as F_, _native as N  # noqa: E402

dev = "cuda:0"
lib = N.load()
print("EP_POOL_WIDEB =", os.environ.get("EP_POOL_WIDEB", "(unset)"))
for (B, Nn, D, Q, amp) in [(9, 40, 4096, 8, 1.0), (5, 17, 4096, 3, 3.0), (3, 196, 4096, 8, 3.0), (300, 31, 4096, 8, 1.0), (4, 8, 4096, 8, 30.0), (2, 1, 4096, 1, 1.0)]:
    g = torch.Generator(device=dev).manual_seed(B + Nn + Q)
    buf = torch.randn(B, Nn + 1, D, device=dev, generator=g).to(torch.bfloat16)
    x = buf[:, 1:]
    cls = torch.randn(Q, D, device=dev, generator=g) * amp / D ** 0.5
    P, S, ML = F_.pool_forward(x, cls, 1.0)
    xs = x.double()
    s = torch.matmul(cls.double(), xs.transpose(1, 2))
    Pr = torch.matmul(torch.softmax(s, -1), xs)
    lse = (ML[..., 0].double() + ML[..., 1].double().log())
    eP = float((P.double() - Pr).abs().max() / Pr.abs().max())
    eS = float((S.double() - s).abs().max())
    eL = float((lse - torch.logsumexp(s, -1)).abs().max())
    # second pass: dcls = scale * sum_b sum_n dS x with dS = A (dP . x - delta), delta = sum_d dP P  (ML[..., 2])
    dP = torch.randn(B, Q, D, device=dev, generator=g)
    ML2 = ML.clone()
    ML2[..., 2] = (dP * P).sum(-1)
    dcls = F_.pool_backward(x, S, ML2, dP, 1.0)
    A = torch.softmax(s, -1)
    dA = torch.matmul(dP.double(), xs.transpose(1, 2))
    dS = A * (dA - (dP.double() * Pr).sum(-1, keepdim=True))
    dcr = torch.einsum("bqn,bnd->qd", dS, xs)
    eG = float((dcls.double() - dcr).abs().max() / dcr.abs().max())
    flag = "" if (eP < 5e-6 and eS < 2e-5 and eL < 1e-5 and eG < 2e-5) else "   <-- BAD"
    print(f"{B:4d} x {Nn:3d} x {D} q{Q} amp {amp}: P {eP:.2e}  S {eS:.2e}  lse {eL:.2e}  dcls {eG:.2e}{flag}", flush=True)
B, Nn, D, Q = 1024, 196, 4096, 8
xs_ = [torch.randn(B, Nn, D, device=dev).to(torch.bfloat16) for _ in range(3)]
cls = torch.randn(Q, D, device=dev) / D ** 0.5
for i in range(3):
    F_.pool_forward(xs_[i % 3], cls, 1.0)
ts = []
for i in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); F_.pool_forward(xs_[i % 3], cls, 1.0); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"forward pass 1024 x 196 x 4096 bf16: {ts[len(ts) // 2]:.1f} us (incl. output allocation) = {B * Nn * D * 2 / ts[len(ts) // 2] / 1e6:.2f} TB/s")
P, S, ML = F_.pool_forward(xs_[0], cls, 1.0)
dP = torch.randn(B, Q, D, device=dev)
dcls = torch.empty(Q, D, device=dev)
for i in range(3):
    F_.pool_backward(xs_[i % 3], S, ML, dP, 1.0, dcls=dcls)
ts = []
for i in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); F_.pool_backward(xs_[i % 3], S, ML, dP, 1.0, dcls=dcls); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"backward pass 1024 x 196 x 4096 bf16: {ts[len(ts) // 2]:.1f} us (incl. workspace allocation and the reduction) = {B * Nn * D * 2 / ts[len(ts) // 2] / 1e6:.2f} TB/s")
