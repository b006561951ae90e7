"""CBAM pooling head on the CPU: pin the oracle (oracle/cbam_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py -> cbam_*.npz) and check the host side of the native module.  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import CBAM_CASES, CBAM_INIT_DIMS, CBAM_PARAM_NAMES, CBAM_SMALL, STEP_LRS, make_cbam_inputs, siglip_sub
from oracle import cbam_oracle as AO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NOISE = {}


def load(case):
    return np.load(os.path.join(GOLD, f"cbam_{case.name}.npz"))


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


def oracle_head(case, inp):
    head = AO.make_head(case.D, case.C)
    params = AO.head_params(head)
    with torch.no_grad():
        for n, p in zip(CBAM_PARAM_NAMES, params):
            p.copy_(torch.from_numpy(inp[n]))
        head[0].bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        head[0].bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
    return head, params


@pytest.mark.parametrize("case", CBAM_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_cbam_inputs(case)
    head, params = oracle_head(case, inp)
    head.train()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(np.ascontiguousarray(xb[:, 1:] if case.strided else xb))
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
            for n, p in zip(CBAM_PARAM_NAMES, params):
                gr = p.grad.numpy()
                close(n, gr if n in CBAM_SMALL else keep(gr), g[f"grad_{n}"], abs_floor=NOISE.get(n, 1e-7))
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p in zip(CBAM_PARAM_NAMES, params):
            small = n in CBAM_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6,
                  abs_floor=NOISE.get(n, 1e-7))
        np.testing.assert_allclose(head[0].bn.running_mean.numpy(), g[f"{tag}_tok_running_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(head[0].bn.running_var.numpy(), g[f"{tag}_tok_running_var"], rtol=1e-5, atol=1e-6)
        assert int(head[0].bn.num_batches_tracked) == int(g[f"{tag}_tok_nbt"]) == step + 1
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


@pytest.mark.parametrize("dim,C", CBAM_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["cbam_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="cbam", nb_classes=C))
    head = enc.head
    assert probe_heads.is_native_cbam_head(head) and head[2] is own and head[0].rd == dim // 16
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd.poolings.cbam import CbamPooling
    for kw in (dict(rd_channels=8), dict(mlp_bias=True), dict(gate_layer="hard_sigmoid"), dict(output_size=2)):
        with pytest.raises(NotImplementedError):
            CbamPooling(channels=64, **kw)
    m = CbamPooling(channels=64)
    with pytest.raises(ValueError, match="perfect square"):
        m(torch.zeros(2, 15, 64))
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 16, 64))


def test_relu_of_gated_sum_factorises():
    """out = mean_n relu(x gc gs + x) = R0 + gc mean_n gs relu(x): the identity the streaming passes rest on (gates in (0, 1))."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 16, 8, generator=g, dtype=torch.float64)
    gc, gs = torch.rand(3, 1, 8, generator=g, dtype=torch.float64), torch.rand(3, 16, 1, generator=g, dtype=torch.float64)
    lhs = torch.relu(x * gc * gs + x).mean(1)
    rhs = torch.relu(x).mean(1) + gc[:, 0] * (gs * torch.relu(x)).mean(1)
    assert torch.allclose(lhs, rhs, rtol=1e-12, atol=1e-14)
