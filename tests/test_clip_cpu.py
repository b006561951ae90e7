"""CLIP attention-pooling head on the CPU: pin the oracle (oracle/clip_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py -> clip_*.npz) and check the host side of the native module.  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import CLIP_CASES, CLIP_INIT_DIMS, CLIP_PARAM_NAMES, CLIP_SMALL, STEP_LRS, make_clip_inputs, siglip_sub
from oracle import clip_oracle as AO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# d loss / d qkv.bias[D:2D] (the key bias) is exactly zero in exact arithmetic: it shifts the N + 1 scores of a head equally and
# cancels in the softmax -- rounding noise on the reference's side, zeros on the native side
NOISE = {"qkv_b": 2e-6}


def load(case):
    return np.load(os.path.join(GOLD, f"clip_{case.name}.npz"))


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


def oracle_head(case, inp):
    head = AO.make_head(case.D, case.C, case.N)
    params = AO.head_params(head)
    with torch.no_grad():
        for n, p in zip(CLIP_PARAM_NAMES, params):
            p.copy_(torch.from_numpy(inp[n]))
    return head, params


@pytest.mark.parametrize("case", CLIP_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_clip_inputs(case)
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
            for n, p in zip(CLIP_PARAM_NAMES, params):
                gr = p.grad.numpy()
                close(n, gr if n in CLIP_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=NOISE.get(n, 1e-7))
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p in zip(CLIP_PARAM_NAMES, params):
            small = n in CLIP_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6,
                  abs_floor=NOISE.get(n, 1e-7))
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


@pytest.mark.parametrize("dim,C", CLIP_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["clip_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="clip", nb_classes=C, model="vit_base_patch16"))
    head = enc.head
    assert probe_heads.is_native_clip_head(head) and head[2] is own and head[0].num_heads == 4
    assert head[0].pos_embed.shape[0] == 197
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.clip import AttentionPool2d
    for kw in (dict(out_features=32), dict(embed_dim=128), dict(qkv_bias=False)):
        with pytest.raises(NotImplementedError):
            AttentionPool2d(in_features=64, feat_size=14, **kw)
    m = AttentionPool2d(in_features=64, feat_size=14)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 196, 64), cls=torch.zeros(2, 1, 64))
    with pytest.raises(ValueError, match="token count"):
        m(torch.zeros(2, 100, 64))                              # the reference's broadcast with pos_embed fails too
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 196, 64))
    # CAPI ViT-L/14 is the one encoder with a 16 x 16 token grid (reference probe_heads.py:54-57)
    enc = _Encoder(64, 10)
    probe_heads.build_probe_head(enc, Namespace(cls_features="clip", nb_classes=10, model="capi_vitl14_in1k"))
    assert enc.head[0].pos_embed.shape[0] == 257
