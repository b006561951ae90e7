"""SimPool heads (simpool / esimpool) on the CPU: pin the oracle (oracle/simpool_oracle.py) against golden vectors produced by
the real reference (tests/golden/make_golden.py -> simpool_*.npz, esimpool_*.npz) and check the host side of the native
modules.  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import (ESIMPOOL_CASES, SIMPOOL_CASES, SIMPOOL_INIT_DIMS, SIMPOOL_SMALL, STEP_LRS, make_simpool_inputs,
                   siglip_sub, simpool_param_names)
from oracle import simpool_oracle as SO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALL = SIMPOOL_CASES + ESIMPOOL_CASES
IDS = [f"{c.family}-{c.name}" for c in ALL]


def load(case):
    return np.load(os.path.join(GOLD, f"{case.family}_{case.name}.npz"))


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", ALL, ids=IDS)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_simpool_inputs(case)
    names = simpool_param_names(case)
    head = SO.make_head(case.D, case.C, case.linears)
    params = SO.head_params(head)
    with torch.no_grad():
        for n, p in zip(names, params):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(xb[:, 1:] if case.strided else xb)
    mus = [torch.zeros_like(p) for p in params]
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for p in params:
            p.grad = None
        pooled = head[0](x)
        logits = head[2](head[1](pooled))
        loss = torch.nn.functional.cross_entropy(logits, t)
        loss.backward()
        if step == 0:
            np.testing.assert_allclose(pooled.detach().numpy(), g["pooled"], rtol=2e-5,
                                       atol=5e-6 * max(1.0, float(np.abs(g["pooled"]).max())))
            with torch.no_grad():
                np.testing.assert_allclose(head[0].attention(x)[0][:, :, 0].numpy(), g["attn"], rtol=1e-4, atol=1e-7)
            np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=2e-5)
            for n, p in zip(names, params):
                gr = p.grad.numpy()
                close(n, gr if n in SIMPOOL_SMALL else keep(gr), g[f"grad_{n}"])
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p in zip(names, params):
            small = n in SIMPOOL_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6)
    head.eval()
    with torch.no_grad():
        np.testing.assert_allclose(head(view(inp["x_buf"])).numpy(), g["eval_logits"], rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("case", [SIMPOOL_CASES[0], SIMPOOL_CASES[1], ESIMPOOL_CASES[0], ESIMPOOL_CASES[1]],
                         ids=lambda c: f"{c.family}-{c.name}")
def test_derived_query_algebra_equals_the_reference_association(case):
    """What the HIP path computes (csrc/ep_simpool.hip header): per-image query rows on the NORMALISED tokens without their
    affine part; simpool pools xhat and applies the affine part afterwards, esimpool pools the raw head slices."""
    g, inp = load(case), make_simpool_inputs(case)
    xb = inp["x_buf"][:, 1:] if case.strided else inp["x_buf"]
    x = torch.from_numpy(np.ascontiguousarray(xb)).double()
    B, N, D = x.shape
    H, dh = case.heads, D // case.heads
    gam, beta = torch.from_numpy(inp["norm_w"]).double(), torch.from_numpy(inp["norm_b"]).double()
    mu = x.mean(-1, keepdim=True); var = x.var(-1, unbiased=False, keepdim=True)
    xhat = (x - mu) / torch.sqrt(var + 1e-6)
    gap = x.mean(1)
    scale = dh ** -0.5
    if case.linears:
        Wq, Wk = torch.from_numpy(inp["wq"]).double(), torch.from_numpy(inp["wk"]).double()
        t = (gap @ Wq.t()) @ Wk                                        # (B, D)
        u = scale * gam * t
        A = torch.softmax(torch.einsum("bd,bnd->bn", u, xhat), -1)     # the constant scale t.beta cancels
        out = gam * torch.einsum("bn,bnd->bd", A, xhat) + beta
        attn = A[:, None]
    else:
        gm = gap.mean(-1, keepdim=True); gv = gap.var(-1, unbiased=False, keepdim=True)
        q = gam * (gap - gm) / torch.sqrt(gv + 1e-6) + beta
        u = scale * q * gam
        out = torch.empty(B, D, dtype=torch.float64)
        attn = torch.empty(B, H, N, dtype=torch.float64)
        for h in range(H):
            sl = slice(h * dh, (h + 1) * dh)
            A = torch.softmax(torch.einsum("bd,bnd->bn", u[:, sl], xhat[:, :, sl]), -1)
            out[:, sl] = torch.einsum("bn,bnd->bd", A, x[:, :, sl])
            attn[:, h] = A
    np.testing.assert_allclose(out.numpy(), g["pooled"], rtol=2e-5, atol=5e-6 * max(1.0, float(np.abs(g["pooled"]).max())))
    np.testing.assert_allclose(attn.numpy(), g["attn"], rtol=2e-4, atol=1e-7)


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


class _Encoder(torch.nn.Module):
    def __init__(self, dim, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, C)


@pytest.mark.parametrize("fam", ["simpool", "esimpool"])
@pytest.mark.parametrize("dim,C", SIMPOOL_INIT_DIMS)
def test_native_head_initialises_like_the_reference(fam, dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["simpool_init"][f"{fam}_d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features=fam, nb_classes=C))
    head = enc.head
    assert probe_heads.is_native_simpool_head(head) and head[2] is own and head[0].num_heads == fx["num_heads"]
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd.poolings.simpool import SimPool, SimPool_nolinears
    for kw in (dict(qkv_bias=True), dict(qk_scale=0.5), dict(gamma=1.25), dict(num_heads=2)):
        with pytest.raises(NotImplementedError):
            SimPool(dim=64, **kw)
    with pytest.raises(NotImplementedError):
        SimPool_nolinears(dim=384, num_heads=12, gamma=2.0)
    with pytest.raises(ValueError):
        SimPool_nolinears(dim=1024, num_heads=12)            # 1024 / 12: the reference's reshape fails too
    m = SimPool(dim=64)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 5, 64), cls=torch.zeros(2, 1, 64))
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 64, 4, 4))
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 5, 64))
