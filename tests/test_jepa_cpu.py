"""V-JEPA attentive pooler on the CPU: pin the oracle (oracle/jepa_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py -> jepa_*.npz) and check the host side of the native module (bit-identical
initialisation incl. truncated-normal weights and residual rescaling, registry wiring).  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import JEPA_CASES, JEPA_INIT_DIMS, JEPA_PARAM_NAMES, JEPA_SMALL, STEP_LRS, make_jepa_inputs, siglip_sub
from oracle import jepa_oracle as JO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# exactly-zero / cancelling gradients: the key half of kv.bias (softmax shift invariance); a constant added to the head's
# output (fc2.bias, and proj.bias / query / norm1.bias through the residual) is removed again by BatchNorm
NOISE = {"kv_b": 5e-6, "fc2_b": 1e-4, "proj_b": 1e-4, "query": 1e-4, "n1_b": 1e-5}


def load(case):
    return np.load(os.path.join(GOLD, f"jepa_{case.name}.npz"))


def close(name, got, want, rtol=1e-4, floor=3e-5, abs_floor=1e-7):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(abs_floor, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", JEPA_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_jepa_inputs(case)
    head = JO.make_head(case.D, case.C, case.heads)
    params = JO.head_params(head)
    with torch.no_grad():
        for n, p in zip(JEPA_PARAM_NAMES, params):
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
            np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=5e-5)
            for n, p in zip(JEPA_PARAM_NAMES, params):
                gr = p.grad.numpy()
                # floor 3e-4 of the tensor's scale: with six images BatchNorm's backward amplifies the upstream
                # gradients, and every entry here is the end of a chain of D- and 4D-long fp32 contractions
                close(n, gr if n in JEPA_SMALL else keep(gr), g[f"grad_{n}"], floor=3e-4, abs_floor=NOISE.get(n, 1e-7))
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p in zip(JEPA_PARAM_NAMES, params):
            small = n in JEPA_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=5e-6,
                  abs_floor=NOISE.get(n, 1e-7))
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        np.testing.assert_allclose(head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).numpy(), g["eval_logits"],
                                   rtol=2e-4, atol=1e-4)


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


class _Encoder(torch.nn.Module):
    def __init__(self, dim, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, C)


@pytest.mark.parametrize("dim,C,heads", JEPA_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C, heads):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["jepa_init"][f"d{dim}_c{C}_h{heads}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="jepa", nb_classes=C, num_heads=heads))
    head = enc.head
    assert probe_heads.is_native_jepa_head(head) and head[2] is own
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_options_outside_the_registry_configuration_raise():
    from efficient_probing_amd.poolings.jepa import AttentivePooler
    for kw in (dict(num_queries=2), dict(depth=2), dict(complete_block=False), dict(qkv_bias=False)):
        with pytest.raises(NotImplementedError):
            AttentivePooler(embed_dim=64, num_heads=4, **kw)
    m = AttentivePooler(embed_dim=64, num_heads=4)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 5, 64))
