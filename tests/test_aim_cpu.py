"""AIM attention-pooling head on the CPU: pin the oracle (oracle/aim_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py -> aim_*.npz) and check the host side of the native module.  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import AIM_CASES, AIM_INIT_DIMS, AIM_PARAM_NAMES, AIM_SMALL, STEP_LRS, make_aim_inputs, siglip_sub
from oracle import aim_oracle as AO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(case):
    return np.load(os.path.join(GOLD, f"aim_{case.name}.npz"))


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


def oracle_head(case, inp):
    head = AO.make_head(case.D, case.C, case.heads)
    params = AO.head_params(head)
    with torch.no_grad():
        for n, p in zip(AIM_PARAM_NAMES, params):
            p.copy_(torch.from_numpy(inp[n]))
        head[0].bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        head[0].bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
    return head, params


@pytest.mark.parametrize("case", AIM_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_aim_inputs(case)
    head, params = oracle_head(case, inp)
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
            np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=2e-5)
            for n, p in zip(AIM_PARAM_NAMES, params):
                gr = p.grad.numpy()
                close(n, gr if n in AIM_SMALL else keep(gr), g[f"grad_{n}"])
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p in zip(AIM_PARAM_NAMES, params):
            small = n in AIM_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6)
        np.testing.assert_allclose(head[0].bn.running_mean.numpy(), g[f"{tag}_tok_running_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(head[0].bn.running_var.numpy(), g[f"{tag}_tok_running_var"], rtol=1e-5, atol=1e-6)
        assert int(head[0].bn.num_batches_tracked) == int(g[f"{tag}_tok_nbt"]) == step + 1
    head.eval()
    with torch.no_grad():
        np.testing.assert_allclose(head(view(inp["x_buf"])).numpy(), g["eval_logits"], rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("case", AIM_CASES[:2], ids=lambda c: c.name)
def test_derived_query_algebra_equals_the_reference_association(case):
    """What the HIP path computes (csrc/ep_aim.hip header): scores from the derived rows r * (scale Wk_h^T q_h) on the RAW
    tokens, pooled raw tokens, projection with Wv diag(r) and bias -Wv (mu r)."""
    g, inp = load(case), make_aim_inputs(case)
    xb = inp["x_buf"][:, 1:] if case.strided else inp["x_buf"]
    x = torch.from_numpy(np.ascontiguousarray(xb)).double()
    B, N, D = x.shape
    H, dh = case.heads, D // case.heads
    q = torch.from_numpy(inp["cls_token"]).double().reshape(D)
    Wk, Wv = torch.from_numpy(inp["k_w"]).double(), torch.from_numpy(inp["v_w"]).double()
    mu = x.mean(dim=(0, 1)); var = x.var(dim=(0, 1), unbiased=False)
    r = 1.0 / torch.sqrt(var + 1e-6)
    scale = dh ** -0.5
    out = torch.empty(B, D, dtype=torch.float64)
    attn = torch.empty(B, H, N, dtype=torch.float64)
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        w = r * (scale * (Wk[sl].t() @ q[sl]))                     # (D,)
        A = torch.softmax(x @ w, dim=-1)                           # (B, N): the constant -w.mu cancels
        P = torch.einsum("bn,bnd->bd", A, x)
        out[:, sl] = P @ (Wv[sl] * r).t() - Wv[sl] @ (mu * r)
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


@pytest.mark.parametrize("dim,C", AIM_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["aim_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="aim", nb_classes=C, num_heads=fx["num_heads"]))
    head = enc.head
    assert probe_heads.is_native_aim_head(head) and head[2] is own and head[0].num_heads == fx["num_heads"]
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd.poolings.aim import AttentionPoolingClassifier
    for kw in (dict(qkv_bias=True), dict(qk_scale=0.5), dict(num_queries=2)):
        with pytest.raises(NotImplementedError):
            AttentionPoolingClassifier(dim=64, num_heads=4, **kw)
    with pytest.raises(ValueError):
        AttentionPoolingClassifier(dim=64, num_heads=12)
    m = AttentionPoolingClassifier(dim=64, num_heads=4)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 5, 64), cls=torch.zeros(2, 1, 64))
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 5, 64))
