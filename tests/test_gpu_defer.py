"""The deferred large update of the EP head step (include/ep_hip.h: phases bits 4 / 5; engine.ProbeHeadEngine.defer_update):
cls_token is updated on the step's stream by the one-launch small-segment optimizer kernel, v.weight / fc.weight / fc.bias
on the aux stream beside the NEXT step's first token pass.  Same arithmetic per tensor as the undeferred step, so after a
flush() everything is bit-equal to an engine that never deferred.  Needs an MI355X (pytest -m gpu)."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _defer_on(monkeypatch):
    monkeypatch.setenv("EP_DEFER_OPT", "1")      # opt-in (slower than the plain step on this stack: engine._can_defer)


def make_engine(seed, opt="lars", D=768, Q=8, C=1000, **kw):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(D, C)
    torch.manual_seed(seed)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C))
    return ProbeHeadEngine(enc.head.to(DEV).train(), optimizer=opt, lr=0.3, weight_decay=1e-4, **kw)


@pytest.mark.parametrize("opt", ["lars", "sgd", "adamw"])
def test_deferred_update_is_bit_equal_after_flush(opt):
    a, b = make_engine(3, opt), make_engine(3, opt)
    a.defer_update = True
    assert a._can_defer() and not b._can_defer()
    g = torch.Generator().manual_seed(17)
    for s in range(5):
        B = 1024 if s != 3 else 512                       # (a batch-size change flushes the update that is in flight)
        x = torch.randn(B, 40, 768, generator=g).to(DEV)
        t = torch.randint(0, 1000, (B,), generator=g).to(DEV)
        lr = 0.3 * (1.0 - 0.1 * s)
        a.train_step(x, t, lr=lr)
        b.train_step(x, t, lr=lr)
        assert a._deferred
        if s == 2:                                         # a reader in the middle: eval_logits() flushes by itself
            assert torch.equal(a.eval_logits(x[:64]), b.eval_logits(x[:64]))
            assert not a._deferred
    a.flush()
    torch.cuda.synchronize()
    assert torch.equal(a.flat_p, b.flat_p)
    assert torch.equal(a.flat_g, b.flat_g)
    for sa, sb in zip(a.state, b.state):
        assert torch.equal(sa, sb)
    assert a.read_stats() == b.read_stats()
    assert a.opt_step == b.opt_step == 5


def test_train_one_epoch_defers_and_leaves_flushed_parameters():
    from efficient_probing_amd import engine_finetune as EF, probe_heads
    from efficient_probing_amd.util.lars import LARS

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(256, 50)
    outs = []
    for defer in (True, False):
        torch.manual_seed(0)
        m = Enc()
        probe_heads.build_probe_head(m, Namespace(cls_features="ep", ep_queries=8, d_out=1, nb_classes=50))
        m.to(DEV)
        g = torch.Generator().manual_seed(2)
        loader = [(torch.randn(64, 30, 256, generator=g), torch.randint(0, 50, (64,), generator=g)) for _ in range(6)]
        opt = LARS(m.head.parameters(), lr=0.0, weight_decay=1e-4)
        args = Namespace(accum_iter=1, amp="none", lr=0.4, min_lr=0.0, warmup_epochs=1, epochs=3)
        import os
        os.environ["EP_DEFER_OPT"] = "1" if defer else "0"
        st = EF.train_one_epoch(m, torch.nn.CrossEntropyLoss(), loader, opt, torch.device(DEV), 1, None, args=args)
        eng = m._ep_engine
        assert not eng._deferred and not eng.defer_update           # flushed and switched off again on the way out
        outs.append((st["loss"], torch.cat([p.detach().flatten() for p in m.head.parameters()]).cpu()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
