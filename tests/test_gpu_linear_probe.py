"""Plain linear probing (BatchNorm1d + Linear on one feature vector per image, reference probe_heads.py:96-99) through
the fused engine (ep_lp_train_step) against stock PyTorch modules run in fp64-free fp32 on the CPU with the reference's
LARS restatement (oracle/torch_port.py).  Needs an MI355X (pytest -m gpu)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle.torch_port import lars_update

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Enc(torch.nn.Module):
    def __init__(self, D, C):
        super().__init__()
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(D, C)


@pytest.mark.parametrize("shape", [(32, 64, 10), (64, 768, 1000), (50, 384, 100)], ids=["tiny", "vitb", "vits"])
def test_linear_probe_engine_matches_torch(shape):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import LinearProbeEngine, make_engine
    B, D, Cc = shape
    torch.manual_seed(0)
    enc = Enc(D, Cc)
    own = enc.head
    probe_heads.build_probe_head(enc, Namespace(cls_features="cls", nb_classes=Cc))
    assert probe_heads.is_native_lp_head(enc.head) and enc.head[1] is own
    ref = torch.nn.Sequential(torch.nn.BatchNorm1d(D, affine=False, eps=1e-6), torch.nn.Linear(D, Cc)).train()
    ref[1].load_state_dict(own.state_dict())
    eng = make_engine(enc.head.to(DEV).train(), optimizer="lars", weight_decay=1e-4)
    assert isinstance(eng, LinearProbeEngine)
    g = torch.Generator().manual_seed(1)
    mus = [torch.zeros_like(p) for p in ref.parameters()]
    for step in range(4):
        x = torch.randn(B, D, generator=g) * 2 + 0.5
        t = torch.randint(0, Cc, (B,), generator=g)
        for p in ref.parameters():
            p.grad = None
        loss = torch.nn.functional.cross_entropy(ref(x), t)
        loss.backward()
        lars_update(list(ref.parameters()), mus, 0.3, weight_decay=1e-4)
        eng.train_step(x.to(DEV), t.to(DEV), lr=0.3)
        assert eng.read_stats()[0] == pytest.approx(loss.item(), rel=2e-5)
    np.testing.assert_allclose(eng.fc.weight.detach().cpu().numpy(), ref[1].weight.detach().numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(eng.fc.bias.detach().cpu().numpy(), ref[1].bias.detach().numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(eng.bn.running_var.cpu().numpy(), ref[0].running_var.numpy(), rtol=1e-5, atol=1e-6)
    ref.eval()
    x = torch.randn(B, D, generator=g)
    with torch.no_grad():
        np.testing.assert_allclose(eng.eval_logits(x.to(DEV)).cpu().numpy(), ref(x).numpy(), rtol=3e-4, atol=3e-5)
    # (B, N, D) tokens are mean-pooled by the token pass first ("pos" / gap features)
    tok = torch.randn(B, 20, D, generator=g)
    with torch.no_grad():
        np.testing.assert_allclose(eng.eval_logits(tok.to(DEV)).cpu().numpy(), ref(tok.mean(1)).numpy(), rtol=3e-4, atol=5e-5)
