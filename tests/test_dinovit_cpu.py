"""DINOv2-block head on the CPU: pin the oracle (oracle/dinovit_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py -> dinovit_*.npz) and check the host side of the native module.  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import (DINOVIT_ATTN_ROWS, DINOVIT_CASES, DINOVIT_INIT_DIMS, DINOVIT_PARAM_NAMES, DINOVIT_SMALL, STEP_LRS,
                   make_dinovit_inputs, siglip_sub)
from oracle import dinovit_oracle as AO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(case):
    return np.load(os.path.join(GOLD, f"dinovit_{case.name}.npz"))


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


def oracle_head(case, inp):
    head = AO.make_head(case.D, case.C)
    params = AO.head_params(head)
    with torch.no_grad():
        for n, p in zip(DINOVIT_PARAM_NAMES, params):
            p.copy_(torch.from_numpy(inp[n]))
    return head, params


@pytest.mark.parametrize("case", DINOVIT_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_dinovit_inputs(case)
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
        pooled, attn = head[0](x, return_attention=True)
        logits = head[2](head[1](pooled))
        loss = torch.nn.functional.cross_entropy(logits, t)
        loss.backward()
        if step == 0:
            a = attn.detach().numpy()
            np.testing.assert_allclose(a if case.full else a[:, :, ::DINOVIT_ATTN_ROWS], g["attn"], rtol=1e-4, atol=1e-7)
            np.testing.assert_allclose(pooled.detach().numpy(), g["pooled"], rtol=2e-5,
                                       atol=5e-6 * max(1.0, float(np.abs(g["pooled"]).max())))
            np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=2e-5)
            for n, p in zip(DINOVIT_PARAM_NAMES, params):
                gr = p.grad.numpy()
                close(n, gr if n in DINOVIT_SMALL else keep(gr), g[f"grad_{n}"])
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p in zip(DINOVIT_PARAM_NAMES, params):
            small = n in DINOVIT_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6)
    head.eval()
    with torch.no_grad():
        np.testing.assert_allclose(head(view(inp["x_buf"])).numpy(), g["eval_logits"], rtol=2e-4, atol=5e-5)


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


class _Encoder(torch.nn.Module):
    def __init__(self, dim, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, C)


@pytest.mark.parametrize("dim,C", DINOVIT_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["dinovit_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="dinovit", nb_classes=C))
    head = enc.head
    assert probe_heads.is_native_dinovit_head(head) and head[2] is own
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd.poolings.dinovit import DinoBlock, DinoViTBlockPooling
    for kw in (dict(qkv_bias=True), dict(proj_bias=False), dict(ffn_bias=False), dict(drop=0.1), dict(attn_drop=0.1),
               dict(init_values=1e-5), dict(drop_path=0.1)):
        with pytest.raises(NotImplementedError):
            DinoBlock(dim=64, num_heads=8, **kw)
    with pytest.raises(AssertionError):
        DinoViTBlockPooling(d_model=60, num_heads=8)
    with pytest.raises(ValueError, match="multiple of 4"):
        DinoViTBlockPooling(d_model=48, num_heads=8)
    m = DinoViTBlockPooling(d_model=64)
    with pytest.raises(ValueError, match="expected tokens"):
        m(torch.zeros(2, 16, 32))
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 16, 64))
