"""CAE attentive-block head on the CPU: pin the oracle (oracle/cae_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py -> cae_*.npz) and check the host side of the native module.  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import CAE_CASES, CAE_INIT_DIMS, CAE_PARAM_NAMES, CAE_SMALL, STEP_LRS, make_cae_inputs, siglip_sub
from oracle import cae_oracle as CO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# exactly-zero gradients hold rounding noise in the reference: a key-side shift (norm1_k.bias) cancels in the softmax, and
# a constant added to the head's output (proj.bias, and norm1_v.bias through Wv and proj) is removed again by BatchNorm
NOISE = {"nk_b": 2e-6, "proj_b": 2e-5, "nv_b": 2e-6}
UNUSED = ("n2_w", "n2_b")


def load(case):
    return np.load(os.path.join(GOLD, f"cae_{case.name}.npz"))


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", CAE_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_cae_inputs(case)
    head = CO.make_head(case.D, case.C)
    params = CO.head_params(head)
    with torch.no_grad():
        for n, p in zip(CAE_PARAM_NAMES, params):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    keep = (lambda a: a) if case.full else siglip_sub
    live = [(n, p) for n, p in zip(CAE_PARAM_NAMES, params) if n not in UNUSED]
    mus = [torch.zeros_like(p) for _, p in live]
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        x = torch.from_numpy(xb[:, 1:] if case.strided else xb)
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
            for n, p in zip(CAE_PARAM_NAMES, params):
                if n in UNUSED:
                    assert p.grad is None and int(g[f"grad_{n}_is_none"]) == 1
                    continue
                gr = p.grad.numpy()
                close(n, gr if n in CAE_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=NOISE.get(n, 1e-7))
        lars_update([p for _, p in live], mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p in zip(CAE_PARAM_NAMES, params):
            small = n in CAE_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6,
                  abs_floor=NOISE.get(n, 1e-7))
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        np.testing.assert_allclose(head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).numpy(), g["eval_logits"],
                                   rtol=2e-4, atol=5e-5)


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


class _Encoder(torch.nn.Module):
    def __init__(self, dim, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, C)


@pytest.mark.parametrize("dim,C", CAE_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["cae_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="cae", nb_classes=C))
    head = enc.head
    assert probe_heads.is_native_cae_head(head) and head[2] is own
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd.poolings.cae import CAEAttentiveBlock
    for kw in (dict(qkv_bias=True), dict(drop=0.1), dict(attn_head_dim=32), dict(qk_scale=0.5)):
        with pytest.raises(NotImplementedError):
            CAEAttentiveBlock(dim=64, **kw)
    m = CAEAttentiveBlock(dim=64)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 5, 64), pos_k=torch.zeros(5, 64))
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 5, 64))
