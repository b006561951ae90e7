"""CPU baseline for bench.py -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the
product package).

An op-for-op restatement, on stock PyTorch CPU ops, of what the reference executes for one
iteration of the EP head: project-then-pool forward exactly as reference poolings/ep.py:28-47
writes it (Linear over every token, reshape/permute, q @ k^T, softmax, attn @ v),
BatchNorm1d(affine=False, eps=1e-6) and Linear (probe_heads.py:106-110,76), CrossEntropyLoss
(main_linprobe.py:589), autograd backward, and the LARS update of util/lars.py:13-37.  It has
the reference's FLOP count and memory behaviour (that is the point of timing it); its numerical
equality with the real reference is pinned by tests/test_oracle_golden.py::test_torch_port_*.
"""
from __future__ import annotations

import time

import torch
from torch import nn


class EPPort(nn.Module):
    def __init__(self, dim, num_queries, d_out=1):
        super().__init__()
        self.scale = dim ** -0.5
        self.num_queries, self.d_out = num_queries, d_out
        self.v = nn.Linear(dim, dim // d_out, bias=False)
        self.cls_token = nn.Parameter(torch.randn(1, num_queries, dim) * 0.02)

    def forward(self, x):
        B, N, C = x.shape
        Q = self.num_queries
        q = self.cls_token.expand(B, -1, -1).reshape(B, Q, 1, C).permute(0, 2, 1, 3) * self.scale
        k = x.reshape(B, N, 1, C).permute(0, 2, 1, 3)
        v = self.v(x).reshape(B, N, Q, C // (self.d_out * Q)).permute(0, 2, 1, 3)
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        out = torch.matmul(attn.squeeze(1).unsqueeze(2), v)
        return out.view(B, C // self.d_out)


def make_head(dim, num_queries, nb_classes, d_out=1):
    pool = EPPort(dim, num_queries, d_out)
    return nn.Sequential(pool, nn.BatchNorm1d(dim // d_out, affine=False, eps=1e-6),
                         nn.Linear(dim // d_out, nb_classes))


@torch.no_grad()
def lars_update(params, mus, lr, weight_decay=0.0, momentum=0.9, tc=0.001):
    for p, mu in zip(params, mus):
        dp = p.grad
        if dp is None:                       # util/lars.py:18-19: parameters without a gradient are skipped
            continue
        if p.ndim > 1:
            dp = dp.add(p, alpha=weight_decay)
            pn, un = torch.norm(p), torch.norm(dp)
            one = torch.ones_like(pn)
            q = torch.where(pn > 0., torch.where(un > 0, tc * pn / un, one), one)
            dp = dp.mul(q)
        mu.mul_(momentum).add_(dp)
        p.add_(mu, alpha=-lr)


def train_step(head, mus, x, targets, lr):
    for p in head.parameters():
        p.grad = None
    loss = nn.functional.cross_entropy(head(x), targets)
    loss.backward()
    lars_update(list(head.parameters()), mus, lr)
    return loss


def time_train_steps(B, N, D, Q, C, budget_s=15.0, threads=None, min_steps=2, make=None):
    """Runs train steps for about ``budget_s`` seconds on the host cores; returns a dict with
    images/s and what was run.  ``make``: factory of another head port (e.g. coca_oracle.make_head)."""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    head = (make() if make is not None else make_head(D, Q, C)).train()
    mus = [torch.zeros_like(p) for p in head.parameters()]
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, N, D, generator=g)
    t = torch.randint(0, C, (B,), generator=g)
    train_step(head, mus, x, t, 1.6)                      # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        train_step(head, mus, x, t, 1.6)
        n += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and n >= min_steps) or n >= 200:
            break
    return dict(value=B * n / el, steps=n, seconds=el, batch=B, threads=torch.get_num_threads())
