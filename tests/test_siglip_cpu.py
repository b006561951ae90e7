"""SigLIP attention-pool head on the CPU: pin the oracle (oracle/siglip_oracle.py) against golden vectors produced by
the real reference (tests/golden/make_golden.py -> siglip_*.npz) and check the host side of the native module
(initialisation parity incl. the truncated-normal latent, state-dict keys, registry wiring).  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import SIGLIP_CASES, SIGLIP_INIT_DIMS, SIGLIP_PARAM_NAMES, SIGLIP_SMALL, STEP_LRS, make_siglip_inputs, siglip_sub
from oracle import siglip_oracle as SO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(case):
    return np.load(os.path.join(GOLD, f"siglip_{case.name}.npz"))


# Gradients that are (partly) sums of cancelling terms: out = z1 + fc2(.) + b2 is shifted uniformly by proj.bias and
# fc2.bias, BatchNorm removes the shift again, so d fc2.bias is exactly zero and d proj.bias keeps only its MLP path;
# d kv.bias[:D] is exactly zero (softmax shift invariance).  Their error is set by the size of the cancelling terms.
NOISE = {"fc2_b": 1e-5, "proj_b": 2e-5, "kv_b": 5e-6}


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", SIGLIP_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_siglip_inputs(case)
    head = SO.make_head(case.D, case.C)
    params = SO.head_params(head)
    with torch.no_grad():
        for n, p in zip(SIGLIP_PARAM_NAMES, params):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    mus = [torch.zeros_like(p) for p in params]
    keep = (lambda a: a) if case.full else siglip_sub
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
            np.testing.assert_allclose(head[0].attention(x)[0][:, :, 0].detach().numpy(), g["attn"], rtol=5e-5, atol=1e-7)
            np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=1e-5)
            for n, p in zip(SIGLIP_PARAM_NAMES, params):
                gr = p.grad.numpy()
                # d kv.bias[:D] is exactly zero in exact arithmetic (softmax shift invariance): rounding noise on both sides
                close(n, gr if n in SIGLIP_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=NOISE.get(n, 1e-7))
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p, mu in zip(SIGLIP_PARAM_NAMES, params, mus):
            small = n in SIGLIP_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6,
                  abs_floor=NOISE.get(n, 1e-7))
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        np.testing.assert_allclose(head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).numpy(), g["eval_logits"],
                                   rtol=2e-4, atol=2e-5)


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


class _Encoder(torch.nn.Module):
    def __init__(self, dim, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, C)


@pytest.mark.parametrize("dim,C", SIGLIP_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["siglip_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="siglip", nb_classes=C))
    head = enc.head
    assert probe_heads.is_native_siglip_head(head) and head[2] is own
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd.poolings.siglip import AttentionPoolLatent
    for kw in (dict(qk_norm=True), dict(latent_len=2), dict(pos_embed="abs"), dict(pool_type="avg"), dict(qkv_bias=False),
               dict(norm_layer=torch.nn.LayerNorm), dict(embed_dim=32)):
        with pytest.raises(NotImplementedError):
            AttentionPoolLatent(in_features=64, **kw)
    m = AttentionPoolLatent(in_features=64, num_heads=4, mlp_ratio=2.0)
    assert m.mlp.fc1.out_features == 128 and m.head_dim == 16
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 5, 64))
