"""CoCa attentional-pooler head on the CPU: pin the oracle (oracle/coca_oracle.py) against the golden vectors
produced by the real reference (tests/golden/make_golden.py -> coca_*.npz), and check the host side of the native
module (initialisation parity, state-dict keys, registry wiring).  No GPU, no kernels."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import COCA_CASES, COCA_INIT_DIMS, COCA_PARAM_NAMES, STEP_LRS, make_coca_inputs, sub
from oracle import coca_oracle as CO
from oracle.torch_port import lars_update

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FWD = dict(rtol=1e-5, atol=1e-6)


def load(case):
    return np.load(os.path.join(GOLD, f"coca_{case.name}.npz"))


def oracle_head(case, inp):
    head = CO.make_head(case.D, case.C, dim_head=case.dim_head, num_img_queries=case.M, heads=case.heads)
    with torch.no_grad():
        for n, p in zip(COCA_PARAM_NAMES, CO.head_params(head)):
            p.copy_(torch.from_numpy(inp[n]))
    return head.train()


def tokens(case, buf):
    return torch.from_numpy(buf[:, 1:] if case.strided else buf)


def check_grad(name, got, want, full):
    scale = max(float(np.abs(want).max()), 1e-12)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=max(1e-6, 3e-5 * scale), err_msg=name)


@pytest.mark.parametrize("case", COCA_CASES, ids=lambda c: c.name)
def test_oracle_forward_and_grads(case):
    g, inp = load(case), make_coca_inputs(case)
    head = oracle_head(case, inp)
    x, t = tokens(case, inp["x_buf"]), torch.from_numpy(inp["targets"])
    pooled = head[0](x)
    z = head[1](pooled)
    logits = head[2](z)
    loss = torch.nn.functional.cross_entropy(logits, t)
    loss.backward()
    np.testing.assert_allclose(pooled.detach().numpy(), g["pooled"], rtol=1e-5, atol=1e-6)
    attn, _ = head[0].attention(x)
    np.testing.assert_allclose(attn[:, :, 0].detach().numpy(), g["attn0"], **FWD)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=1e-5)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-5)
    keep = (lambda a: a) if case.full else sub
    for n, p in zip(COCA_PARAM_NAMES, CO.head_params(head)):
        gr = p.grad.numpy()
        if n == "img_queries":
            check_grad(n, gr[0], g["grad_img_queries_row0"], True)
            assert float(np.abs(gr[1:]).max() if gr.shape[0] > 1 else 0.0) == 0.0 == float(g["grad_img_queries_rest_absmax"])
        else:
            check_grad(n, gr if n in ("gamma", "fc_bias") else keep(gr), g[f"grad_{n}"], case.full)
        assert float(p.grad.double().norm()) == pytest.approx(float(g[f"gradnorm_{n}"]), rel=1e-4, abs=1e-9)


@pytest.mark.parametrize("case", COCA_CASES, ids=lambda c: c.name)
def test_oracle_lars_steps(case):
    """The oracle head trained with the LARS restatement (oracle/torch_port.py, pinned on its own against
    reference util/lars.py) reproduces the reference's parameters / momentum after each recorded step."""
    g, inp = load(case), make_coca_inputs(case)
    head = oracle_head(case, inp)
    params = CO.head_params(head)
    mus = [torch.zeros_like(p) for p in params]
    keep = (lambda a: a) if case.full else sub
    for step in range(case.steps):
        x = tokens(case, inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(head(x), t)
        loss.backward()
        lars_update(params, mus, STEP_LRS[step % len(STEP_LRS)], weight_decay=case.weight_decay)
        tag = f"lars{step + 1}"
        assert loss.item() == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-5)
        for n, p, mu in zip(COCA_PARAM_NAMES, params, mus):
            small = n in ("gamma", "fc_bias")
            got_p = p.detach().numpy() if small else keep(p.detach().numpy())
            got_mu = mu.numpy() if small else keep(mu.numpy())
            np.testing.assert_allclose(got_p, g[f"{tag}_{n}"], rtol=2e-4, atol=2e-6, err_msg=f"{tag} {n}")
            sc = max(float(np.abs(g[f"{tag}_mu_{n}"]).max()), 1e-12)
            np.testing.assert_allclose(got_mu, g[f"{tag}_mu_{n}"], rtol=5e-4, atol=5e-5 * sc, err_msg=f"{tag} mu {n}")
        np.testing.assert_allclose(head[1].running_mean.numpy(), g[f"{tag}_running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(head[1].running_var.numpy(), g[f"{tag}_running_var"], rtol=1e-4, atol=1e-6)
    head.eval()
    with torch.no_grad():
        np.testing.assert_allclose(head(tokens(case, inp["x_buf"])).numpy(), g["eval_logits"], rtol=2e-4, atol=2e-5)


# ---- host side of the native module ---------------------------------------------------------------------------
def _sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


class _Encoder(torch.nn.Module):
    def __init__(self, dim, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, C)


def _args(**kw):
    a = Namespace(cls_features="coca", ep_queries=32, d_out=1, nb_classes=1000, num_heads=16, abmilp_sa="both",
                  abmilp_act="tanh", abmilp_depth=2, abmilp_cond=None, abmilp_content="all", model="vit_base_patch16")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("dim,C", COCA_INIT_DIMS)
def test_native_head_initialises_like_the_reference(dim, C):
    """Same RNG draws in the same order (reference probe_heads.py:14-16,104): the encoder's Linear first, then
    img_queries, to_q, to_kv, to_out -- bit-identical state dict under torch.manual_seed(0)."""
    from efficient_probing_amd import probe_heads
    fx = json.load(open(os.path.join(GOLD, "host_fixtures.json")))["coca_init"][f"d{dim}_c{C}"]
    torch.manual_seed(0)
    enc = _Encoder(dim, C)
    own_head = enc.head
    probe_heads.build_probe_head(enc, _args(nb_classes=C))
    head = enc.head
    assert probe_heads.is_native_coca_head(head) and not probe_heads.is_native_ep_head(head)
    assert head[2] is own_head                                  # the encoder's classifier object is kept (:105)
    sd = head.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == fx["keys"]
    assert list(sd.keys()) == list(fx["keys"].keys()) or sorted(sd.keys()) == sorted(fx["keys"].keys())
    for k, v in sd.items():
        assert _sha(v) == fx["sha256"][k], k
    assert sum(p.numel() for p in head.parameters()) == fx["n_trainable"]


def test_coca_all_variant_and_unsupported_options():
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.poolings.coca import CrossAttention
    enc = _Encoder(64, 10)
    probe_heads.build_probe_head(enc, _args(cls_features="coca_all", nb_classes=10))
    assert probe_heads.is_native_coca_head(enc.head)
    with pytest.raises(NotImplementedError):
        CrossAttention(dim=64, parallel_ff=True)
    with pytest.raises(NotImplementedError):
        CrossAttention(dim=64, norm_context=True)
    m = CrossAttention(dim=64, num_img_queries=3)
    with pytest.raises(RuntimeError):                           # no CPU path: fails loudly
        m(torch.zeros(2, 5, 64))


def test_so400m_parameter_count():
    """2,707,048 trainable parameters for the CoCa probe at D = 1152, C = 1000
    ... of which the pooler holds everything but the classifier."""
    from efficient_probing_amd.poolings.coca import CrossAttention
    pool = CrossAttention(dim=1152)
    n_pool = sum(p.numel() for p in pool.parameters())
    assert n_pool == 1152 + 196 * 1152 + 512 * 1152 + 128 * 1152 + 1152 * 512
    assert n_pool + 1152 * 1000 + 1000 == 2_707_048
