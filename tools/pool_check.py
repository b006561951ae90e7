import sys, torch
sys.path.insert(0, "/root/repo")
from efficient_probing_amd import functional as F_, _native as N
lib = N.load()
torch.manual_seed(0)
dev = "cuda:0"
for (B, Nn, D, Q, amp) in [(5, 256, 1152, 16, 1.0), (5, 256, 1152, 16, 30.0), (64, 256, 1152, 16, 1.0), (5, 256, 1152, 12, 1.0), (5, 256, 1152, 8, 30.0), (5, 197, 768, 16, 30.0), (5, 256, 1024, 16, 30.0), (5, 256, 1280, 16, 1.0)]:
    x = torch.randn(B, Nn, D, device=dev)
    cls = torch.randn(Q, D, device=dev) * amp / D ** 0.5
    P, S, ML = F_.pool_forward(x, cls, 1.0)
    s = torch.einsum("qd,bnd->bqn", cls.double(), x.double())
    A = torch.softmax(s, -1)
    Pr = torch.einsum("bqn,bnd->bqd", A, x.double())
    err = (P.double() - Pr).abs().max().item()
    print(B, Nn, D, Q, amp, lib.ep_pool_kernel_name(B, Nn, D, Q, 0).decode(), f"err {err:.2e}")
