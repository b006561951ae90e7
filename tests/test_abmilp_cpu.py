"""AbMILP head on the CPU: pin the oracle (oracle/abmilp_oracle.py) against the golden vectors produced by the real
reference (tests/golden/make_golden.py -> abmilp_*.npz), and check the host side of the native module (initialisation
parity, state-dict keys, registry wiring, option validation).  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import ABMILP_CASES, ABMILP_INIT_DIMS, ABMILP_PARAM_NAMES, ABMILP_SMALL, STEP_LRS, make_abmilp_inputs, sub
from oracle import abmilp_oracle as AO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(case):
    return np.load(os.path.join(GOLD, f"abmilp_{case.name}.npz"))


def oracle_head(case, inp):
    head = AO.make_head(case.D, case.C, case.content)
    with torch.no_grad():
        for n, p in zip(ABMILP_PARAM_NAMES, AO.head_params(head)):
            p.copy_(torch.from_numpy(inp[n]))
    return head.train()


def close(name, got, want, rtol=1e-4, floor=3e-5):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(1e-7, floor * scale), err_msg=name)


@pytest.mark.parametrize("case", ABMILP_CASES, ids=lambda c: c.name)
def test_oracle_forward_grads_and_steps(case):
    g, inp = load(case), make_abmilp_inputs(case)
    head = oracle_head(case, inp)
    params = AO.head_params(head)
    mus = [torch.zeros_like(p) for p in params]
    keep = (lambda a: a) if case.full else sub
    for step in range(case.steps):
        x = torch.from_numpy(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for p in params:
            p.grad = None
        pooled, amap = head[0].forward_with_attn_map(x)
        logits = head[2](head[1](pooled))
        loss = torch.nn.functional.cross_entropy(logits, t)
        loss.backward()
        if step == 0:
            np.testing.assert_allclose(pooled.detach().numpy(), g["pooled"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(amap.detach().numpy(), g["attn_map"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=1e-5)
            for n, p in zip(ABMILP_PARAM_NAMES, params):
                gr = p.grad.numpy()
                close(n, gr if n in ABMILP_SMALL else keep(gr), g[f"grad_{n}"])
                assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=1e-4, abs=1e-9)
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p, mu in zip(ABMILP_PARAM_NAMES, params, mus):
            small = n in ABMILP_SMALL
            close(f"{tag} {n}", p.detach().numpy() if small else keep(p.detach().numpy()), g[f"{tag}_{n}"], rtol=2e-4, floor=2e-6)
            close(f"{tag} mu {n}", mu.numpy() if small else keep(mu.numpy()), g[f"{tag}_mu_{n}"], rtol=5e-4, floor=5e-5)
    head.eval()
    with torch.no_grad():
        np.testing.assert_allclose(head(torch.from_numpy(inp["x_buf"])).numpy(), g["eval_logits"], rtol=2e-4, atol=2e-5)


def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


class _Encoder(torch.nn.Module):
    def __init__(self, dim, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, C)


def _args(**kw):
    a = Namespace(cls_features="abmilp", ep_queries=32, d_out=1, nb_classes=1000, num_heads=16, abmilp_sa="both",
                  abmilp_act="tanh", abmilp_depth=2, abmilp_cond=None, abmilp_content="all", model="vit_base_patch16")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("dim,C", ABMILP_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["abmilp_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own = enc.head
    probe_heads.build_probe_head(enc, _args(nb_classes=C))
    head = enc.head
    assert probe_heads.is_native_abmilp_head(head) and head[2] is own
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]
    if dim == 1152:
        assert fx["n_trainable"] == 7_791_977                   # SURVEY.md section 8 (a14), SO400M width


def test_options_outside_the_defaults_raise():
    from efficient_probing_amd.poolings.abmilp import ABMILPHead
    for kw in (dict(self_attention_apply_to="map"), dict(self_attention_apply_to="both", activation="relu"),
               dict(self_attention_apply_to="both", depth=3), dict(self_attention_apply_to="both", cond="pe")):
        with pytest.raises(NotImplementedError):
            ABMILPHead(dim=64, **kw)
    m = ABMILPHead(dim=64, self_attention_apply_to="both")
    with pytest.raises(RuntimeError, match="GPU"):                # no CPU path: fails loudly
        m(torch.zeros(2, 5, 64))
