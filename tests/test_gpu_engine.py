"""engine_finetune on the GPU: the fused path and the module (autograd) path of train_one_epoch
produce the same head, gradient accumulation matches one big step's gradients, and evaluate() agrees
with the oracle.  Needs an MI355X (pytest -m gpu)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from cases import Case, make_inputs
from oracle import ep_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Enc(torch.nn.Module):
    def __init__(self, dim, classes):
        super().__init__()
        self.head = torch.nn.Linear(dim, classes)

    def forward(self, tokens):
        return self.head(tokens)


def make_model(case, inp):
    from efficient_probing_amd import probe_heads
    torch.manual_seed(0)
    m = Enc(case.D, case.C)
    probe_heads.build_probe_head(m, Namespace(cls_features="ep", ep_queries=case.Q, d_out=1, nb_classes=case.C))
    with torch.no_grad():
        m.head[0].cls_token.copy_(torch.from_numpy(inp["cls_token"]))
        m.head[0].v.weight.copy_(torch.from_numpy(inp["v_weight"]))
        m.head[2].weight.copy_(torch.from_numpy(inp["fc_weight"]))
        m.head[2].bias.copy_(torch.from_numpy(inp["fc_bias"]))
    return m.to(DEV)


def loader_of(case, inp, n_batches):
    xs = [inp["x_buf"], inp["x_buf2"]]
    ts = [inp["targets"], inp["targets2"]]
    return [(torch.from_numpy(xs[i % 2]), torch.from_numpy(ts[i % 2])) for i in range(n_batches)]


ARGS = Namespace(accum_iter=1, amp="none", lr=0.4, min_lr=0.0, warmup_epochs=1, epochs=3)


def run(case, inp, fused, accum=1, n_batches=4):
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd.util.lars import LARS
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    from efficient_probing_amd import functional as F_
    model = make_model(case, inp)
    opt = LARS(model.head.parameters(), lr=0.0, weight_decay=1e-4)
    args = Namespace(**{**vars(ARGS), "accum_iter": accum})
    loader = loader_of(case, inp, n_batches)
    if fused:
        stats = EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), loader, opt, torch.device(DEV), 1,
                                   NativeScalerWithGradNormCount(), args=args)
    else:
        class Wrapped(torch.nn.Module):           # hides the head from the fused-path detection
            def __init__(self, m):
                super().__init__()
                self.inner = m

            def forward(self, x):
                return self.inner(x)
        crit = lambda out, t: F_.cross_entropy_loss(out, t)[0]
        stats = EF.train_one_epoch(Wrapped(model), crit, loader, opt, torch.device(DEV), 1,
                                   NativeScalerWithGradNormCount(), args=args)
    return model, opt, stats


def test_fused_and_module_paths_agree():
    case = Case("ef", B=16, N=40, D=256, Q=8, C=20, seed=5)
    inp = make_inputs(case)
    m1, o1, s1 = run(case, inp, fused=True)
    m2, o2, s2 = run(case, inp, fused=False)
    for (n1, p1), (n2, p2) in zip(m1.head.named_parameters(), m2.head.named_parameters()):
        np.testing.assert_allclose(p1.detach().cpu().numpy(), p2.detach().cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=n1)
    assert s1["loss"] == pytest.approx(s2["loss"], rel=1e-4)
    assert s1["acc1"] == pytest.approx(s2["acc1"]) and s1["lr"] == pytest.approx(s2["lr"])
    np.testing.assert_allclose(m1.head[1].running_var.cpu().numpy(), m2.head[1].running_var.cpu().numpy(), rtol=1e-5)
    # optimizer state is exposed in the reference's format on both paths
    for p in m1.head.parameters():
        assert "mu" in o1.state[p] and o1.state[p]["mu"].shape == p.shape
    sd = o1.state_dict()
    assert all("mu" in st for st in sd["state"].values())


def test_gradient_accumulation_matches_reference_semantics():
    """accum_iter = 2: gradients of the two micro-batches (each loss / 2) add up before one LARS step
    (reference engine_finetune.py:72-77); BatchNorm sees each micro-batch separately."""
    case = Case("acc", B=8, N=24, D=128, Q=4, C=12, seed=9)
    inp = make_inputs(case)
    m, opt, _ = run(case, inp, fused=True, accum=2, n_batches=2)
    st = O.HeadState(cls_token=inp["cls_token"].copy(), v_weight=inp["v_weight"].copy(),
                     fc_weight=inp["fc_weight"].copy(), fc_bias=inp["fc_bias"].copy(),
                     running_mean=np.zeros(case.D, np.float32), running_var=np.ones(case.D, np.float32),
                     num_queries=case.Q, d_out=1)
    grads = None
    for xb, tg in ((inp["x_buf"], inp["targets"]), (inp["x_buf2"], inp["targets2"])):
        out, cache = O.head_forward_train(st, xb, tg)
        g = O.head_backward(st, cache, loss_scale=0.5)
        gl = [g[k] for k in O.PARAM_ORDER]
        grads = gl if grads is None else [a + b for a, b in zip(grads, gl)]
        st.running_mean, st.running_var, st.num_batches_tracked = cache["new_bn"]
    from efficient_probing_amd.util.lr_sched import lr_at
    lr = lr_at(1.0, 0.4, 0.0, 1, 3)                 # schedule point of micro-step 0 of epoch 1
    ps, _ = O.lars_step(st.params(), grads, [None] * 4, lr=lr, weight_decay=1e-4)
    for got, want, n in zip(m.head.parameters(), ps, O.PARAM_ORDER):
        np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=2e-4, atol=2e-6, err_msg=n)
    assert int(m.head[1].num_batches_tracked) == 2


def test_evaluate_matches_oracle():
    from efficient_probing_amd import engine_finetune as EF
    case = Case("ev", B=16, N=33, D=256, Q=8, C=20, seed=2)
    inp = make_inputs(case)
    model = make_model(case, inp)
    with torch.no_grad():
        model.head[1].running_mean.normal_(0, 0.1)
        model.head[1].running_var.uniform_(0.5, 1.5)
    st = O.HeadState(cls_token=inp["cls_token"], v_weight=inp["v_weight"], fc_weight=inp["fc_weight"],
                     fc_bias=inp["fc_bias"], running_mean=model.head[1].running_mean.cpu().numpy(),
                     running_var=model.head[1].running_var.cpu().numpy(), num_queries=case.Q, d_out=1)
    loader = loader_of(case, inp, 2)
    stats = EF.evaluate(loader, model, torch.device(DEV), return_targets_and_preds=True, precision="fp32")
    accs, losses = [], []
    for xb, tg in ((inp["x_buf"], inp["targets"]), (inp["x_buf2"], inp["targets2"])):
        logits = O.head_forward_eval(st, xb)
        loss, _ = O.cross_entropy(logits, tg)
        accs.append(O.accuracy(logits, tg)); losses.append(float(loss))
    assert stats["loss"] == pytest.approx(np.mean(losses), rel=1e-4)
    assert stats["acc1"] == pytest.approx(np.mean([a[0] for a in accs]))
    assert stats["acc5"] == pytest.approx(np.mean([a[1] for a in accs]))
    assert stats["preds"].shape == (2 * case.B,)
    # the reference's evaluation precision (fp16 autocast, engine_finetune.py:131) through the same call: the fused path
    # (engine emulation) and the module path (torch autocast around the native modules) agree to fp16 resolution, and
    # sit within fp16 resolution of the fp32 evaluation
    s16 = EF.evaluate(loader, model, torch.device(DEV), precision="fp16_autocast")
    sdef = EF.evaluate(loader, model, torch.device(DEV))              # the default IS the reference's mode (fp16 autocast)
    assert sdef["loss"] == s16["loss"] and sdef["acc1"] == s16["acc1"]
    assert s16["loss"] == pytest.approx(stats["loss"], rel=2e-3)
    assert abs(s16["acc1"] - stats["acc1"]) <= 100.0 / case.B + 1e-9
    with pytest.raises(ValueError):
        EF.evaluate(loader, model, torch.device(DEV), precision="bf16")


def _adamw_run(case, inp, epochs, resume_after=None, tmp_path=None):
    """`epochs` epochs of the fused path under torch.optim.AdamW; with `resume_after`, the run is cut after that many
    epochs, saved with checkpoint.save_model, and continued in a FRESH model / optimizer through checkpoint.load_model."""
    from efficient_probing_amd import engine_finetune as EF, checkpoint as CK
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    args = Namespace(**{**vars(ARGS), "epochs": epochs + 1, "output_dir": str(tmp_path) if tmp_path else "", "suffix": "t",
                        "resume": "", "start_epoch": 0})

    def fresh():
        m = make_model(case, inp)
        return m, torch.optim.AdamW(m.head.parameters(), lr=0.0, weight_decay=0.05, betas=(0.9, 0.95))
    model, opt = fresh()
    loader = loader_of(case, inp, 3)
    for ep in range(epochs):
        if resume_after is not None and ep == resume_after:
            path = CK.save_model(args, ep - 1, model, model.head, opt, NativeScalerWithGradNormCount(), {})
            model, opt = fresh()
            args.resume = str(path)
            CK.load_model(args, model, opt, NativeScalerWithGradNormCount())
            assert args.start_epoch == ep
        EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), loader, opt, torch.device(DEV), ep,
                           NativeScalerWithGradNormCount(), args=args)
    return model, opt


def test_adamw_save_and_resume_continues_the_trajectory(tmp_path):
    """with AdamW the fused path kept exp_avg / exp_avg_sq / step only inside the engine, so a --resume
    restarted the moments and the bias correction.  Now optimizer.state holds views of the engine's buffers (and the
    step), so the checkpoint carries them and a resumed run equals the uninterrupted one bit for bit."""
    case = Case("aw", B=16, N=24, D=128, Q=4, C=12, seed=4)
    inp = make_inputs(case)
    m_full, o_full = _adamw_run(case, inp, epochs=3)
    m_res, o_res = _adamw_run(case, inp, epochs=3, resume_after=2, tmp_path=tmp_path)
    for (n1, p1), (n2, p2) in zip(m_full.head.named_parameters(), m_res.head.named_parameters()):
        assert torch.equal(p1, p2), n1
    sd = o_res.state_dict()["state"]
    assert len(sd) == 4 and all({"step", "exp_avg", "exp_avg_sq"} <= set(st) for st in sd.values())
    assert all(float(st["step"]) == 9.0 for st in sd.values())                 # 3 epochs x 3 steps
    for a, b in zip(o_full.state_dict()["state"].values(), sd.values()):
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])
    # and the moments are what torch's own AdamW computes from the same gradients (one step, module path)
    from efficient_probing_amd import functional as F_
    m_t = make_model(case, inp)
    o_t = torch.optim.AdamW(m_t.head.parameters(), lr=0.01, weight_decay=0.05, betas=(0.9, 0.95))
    x, t = torch.from_numpy(inp["x_buf"]).to(DEV), torch.from_numpy(inp["targets"]).to(DEV)
    F_.cross_entropy_loss(m_t.head(x), t)[0].backward()
    o_t.step()
    from efficient_probing_amd.engine import make_engine
    m_e = make_model(case, inp)
    eng = make_engine(m_e.head, optimizer="adamw", lr=0.01, weight_decay=0.05, betas=(0.9, 0.95))
    eng.train_step(x, t)
    # Adam's first step moves every element by lr * g / (|g| + eps): an element whose gradient is of the order of eps
    # (1e-8) turns the two paths' last-bit gradient difference into a visible fraction of lr = 0.01 -- hence the atol
    for p_t, p_e in zip(m_t.head.parameters(), m_e.head.parameters()):
        np.testing.assert_allclose(p_e.detach().cpu().numpy(), p_t.detach().cpu().numpy(), rtol=2e-5, atol=1e-4)
        assert float((p_e.detach() - p_t.detach()).abs().mean()) < 1e-7


def test_evaluate_before_training_does_not_fix_the_optimizer():
    """evaluate() used to create (and cache) a LARS engine with weight_decay 0; a later train_one_epoch with
    SGD / another weight decay silently reused it."""
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    case = Case("evtr", B=16, N=24, D=128, Q=4, C=12, seed=6)
    inp = make_inputs(case)
    loader = loader_of(case, inp, 2)

    def train(evaluate_first):
        model = make_model(case, inp)
        if evaluate_first:
            EF.evaluate(loader, model, torch.device(DEV))
            assert EF.get_engine(model)._eval_only
        opt = torch.optim.SGD(model.head.parameters(), lr=0.0, weight_decay=0.01)
        EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), loader, opt, torch.device(DEV), 1,
                           NativeScalerWithGradNormCount(), args=ARGS)
        eng = EF.get_engine(model)
        assert eng.optimizer_name == "sgd" and eng.weight_decay == 0.01 and not eng._eval_only
        return model
    a, b = train(False), train(True)
    for p, q in zip(a.head.parameters(), b.head.parameters()):
        assert torch.equal(p, q)
    # another optimizer class on the same model is an error, not a silent reuse
    from efficient_probing_amd.util.lars import LARS
    with pytest.raises(RuntimeError, match="built for sgd"):
        EF.get_engine(a, LARS(a.head.parameters(), lr=0.1))


def test_clipping_and_other_criteria_take_the_module_path():
    """The fused step computes plain mean cross-entropy and does not clip: max_norm, label smoothing or mixup must
    not be dropped silently -- those calls run the reference's module loop."""
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd.util.lars import LARS
    from efficient_probing_amd.util.misc import NativeScalerWithGradNormCount
    case = Case("clip", B=16, N=24, D=128, Q=4, C=12, seed=7)
    inp = make_inputs(case)
    loader = loader_of(case, inp, 2)
    for kw, crit in ((dict(max_norm=0.5), torch.nn.CrossEntropyLoss()), (dict(), torch.nn.CrossEntropyLoss(label_smoothing=0.1))):
        model = make_model(case, inp)
        opt = LARS(model.head.parameters(), lr=0.0)
        EF.train_one_epoch(model, crit, loader, opt, torch.device(DEV), 1, NativeScalerWithGradNormCount(), args=ARGS, **kw)
        assert getattr(model, "_ep_engine", None) is None
    assert EF._plain_cross_entropy(torch.nn.CrossEntropyLoss()) and not EF._plain_cross_entropy(torch.nn.MSELoss())


def test_store_batches_through_train_one_epoch_and_evaluate():
    """``(store, image_index, targets)`` batches of a resident token store (token_store.ResidentTokenStore.loader) go
    through train_one_epoch / evaluate in place and give what the same images give as gathered token tensors."""
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd.token_store import ResidentTokenStore
    from efficient_probing_amd.util.lars import LARS
    case = Case("store", B=16, N=40, D=256, Q=8, C=20, seed=7)
    inp = make_inputs(case)
    tokens = torch.from_numpy(np.concatenate([inp["x_buf"], inp["x_buf2"]], 0)).to(DEV)      # 2 B images
    labels = torch.from_numpy(np.concatenate([inp["targets"], inp["targets2"]], 0)).to(DEV)
    store = ResidentTokenStore.from_tensors(tokens, labels, seed=3)
    assert len(store.loader(case.B)) == 2
    outs = []
    for use_store in (True, False):
        model = make_model(case, inp)
        opt = LARS(model.head.parameters(), lr=0.0, weight_decay=1e-4)
        loader = store.loader(case.B, epoch=1)
        if not use_store:                         # the same batches, gathered into dense token tensors
            loader = [(tok[idx.long()].contiguous(), tgt) for tok, idx, tgt in loader]
        stats = EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), loader, opt, torch.device(DEV), 1, None, args=ARGS)
        ev = EF.evaluate(store.loader(case.B, shuffle=False) if use_store else
                         [(tok[idx.long()].contiguous(), tgt) for tok, idx, tgt in store.loader(case.B, shuffle=False)],
                         model, torch.device(DEV))
        outs.append((stats, ev, torch.cat([p.detach().flatten() for p in model.head.parameters()]).cpu()))
    (s1, e1, p1), (s2, e2, p2) = outs
    assert s1["loss"] == pytest.approx(s2["loss"], rel=1e-6)
    assert torch.equal(p1, p2)                    # same kernels on the same images: bit-equal parameters
    assert e1["acc1"] == e2["acc1"] and e1["loss"] == pytest.approx(e2["loss"], rel=1e-6)


def test_three_element_batches_are_read_like_the_reference_and_partial_last_batch():
    """A loader that yields ``(tokens, extra, target)`` is read as the reference reads it -- ``batch[0]`` and ``batch[-1]``
    (reference engine_finetune.py:40-41,125-126) -- even when ``extra`` looks like an index vector; only a
    ``token_store.StoreBatch`` is dereferenced in place.  The epoch ends on a PARTIAL batch: the accuracy meters divide every
    window by the images it really held."""
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd.token_store import StoreBatch
    from efficient_probing_amd.util.lars import LARS
    case = Case("three", B=16, N=24, D=128, Q=4, C=12, seed=5)
    inp = make_inputs(case)
    x = torch.from_numpy(inp["x_buf"]).to(DEV); t = torch.from_numpy(inp["targets"]).to(DEV)
    extra = torch.arange(case.B, device=DEV, dtype=torch.int32).flip(0)        # would permute the batch if it were used as an index
    part = 5                                                                    # last batch: 5 of 16 images
    two = [(x, t)] * 22 + [(x[:part].contiguous(), t[:part])]
    three = [(x, extra, t)] * 22 + [(x[:part].contiguous(), extra[:part].contiguous(), t[:part])]
    outs = []
    for loader in (two, three):
        model = make_model(case, inp)
        opt = LARS(model.head.parameters(), lr=0.0)
        st = EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), loader, opt, torch.device(DEV), 1, None, args=ARGS)
        ev = EF.evaluate(loader, model, torch.device(DEV))
        outs.append((st, ev, torch.cat([p.detach().flatten() for p in model.head.parameters()]).cpu()))
    (s2, e2, p2), (s3, e3, p3) = outs
    assert torch.equal(p2, p3) and s2["loss"] == s3["loss"] and e2["acc1"] == e3["acc1"]
    assert 0.0 <= s2["acc1"] <= 100.0 and 0.0 <= s2["acc5"] <= 100.0
    # the same images as a store batch ARE dereferenced: flipped order, flipped targets -> the same loss as the dense flipped batch
    model = make_model(case, inp)
    opt = LARS(model.head.parameters(), lr=0.0)
    sa = EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), [StoreBatch(x, extra, t.flip(0))], opt, torch.device(DEV), 1, None, args=ARGS)
    model = make_model(case, inp)
    opt = LARS(model.head.parameters(), lr=0.0)
    sb = EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), [(x.flip(0).contiguous(), t.flip(0))], opt, torch.device(DEV), 1, None, args=ARGS)
    assert sa["loss"] == pytest.approx(sb["loss"], rel=1e-6)


def test_epoch_meters_read_one_window_behind_account_every_step():
    """train_one_epoch reads a window's statistics when the next window's read-back is enqueued (no queue drain): over an
    epoch of 2 1/2 print windows the averaged loss still equals the mean of the per-step losses of a plain loop, and the
    asynchronous read-back itself returns what the blocking one does."""
    from efficient_probing_amd import engine_finetune as EF
    from efficient_probing_amd.util.lars import LARS
    case = Case("meters", B=8, N=12, D=64, Q=4, C=10, seed=11)
    inp = make_inputs(case)
    x = torch.from_numpy(inp["x_buf"]).to(DEV); t = torch.from_numpy(inp["targets"]).to(DEV)
    n_it = 50                                                     # print_freq = 20: windows of 20, 20 and 10 steps
    per_step = []
    model = make_model(case, inp)
    opt = LARS(model.head.parameters(), lr=0.05, weight_decay=1e-4)
    eng = EF.get_engine(model, opt, ARGS)
    for _ in range(n_it):
        eng.train_step(x, t, lr=0.05)
        h = eng.read_stats_async()
        per_step.append(eng.wait_stats(h)[0])
    assert eng.read_stats()[0] == 0.0                             # the asynchronous read cleared the counters behind it
    model = make_model(case, inp)
    opt = LARS(model.head.parameters(), lr=0.05, weight_decay=1e-4)
    args = Namespace(**{**vars(ARGS), "lr": 0.05, "min_lr": 0.05, "warmup_epochs": 0, "epochs": 1})
    stats = EF.train_one_epoch(model, torch.nn.CrossEntropyLoss(), [(x, t)] * n_it, opt, torch.device(DEV), 0, None, args=args)
    assert stats["loss"] == pytest.approx(float(np.mean(per_step)), rel=1e-5)


def test_weight_planes_follow_torch_side_parameter_writes():
    """The bf16 planes of fc.weight (and v.weight at D >= 2048) live in the engine's workspace and are rewritten by the
    optimizer's update kernel; ``planes_valid`` tells a step to trust them.  A parameter write from the torch side --
    load_state_dict, an in-place op on the Parameter -- bumps the Parameter's version counter: the engine sees it and the
    next step splits the planes again.  Checked against a FRESH engine on the same weights (bit-equal parameters after the
    step); a write through ``p.data`` is invisible to torch's counters and needs ``invalidate_planes()``."""
    from efficient_probing_amd.engine import ProbeHeadEngine
    for D in (256, 2048):
        case = Case("planes", B=128, N=12, D=D, Q=8, C=40, seed=3)
        inp = make_inputs(case)
        x = torch.from_numpy(inp["x_buf"]).to(DEV); t = torch.from_numpy(inp["targets"]).to(DEV)
        model = make_model(case, inp)
        eng = ProbeHeadEngine(model.head, optimizer="lars", lr=0.3)
        eng.train_step(x, t); eng.train_step(x, t)
        assert eng._planes_current()                                   # written by the update kernel, nobody touched the weights
        sd = {k: v.clone() for k, v in model.head.state_dict().items()}
        sd["2.weight"] = sd["2.weight"] * 1.5 + 0.01
        sd["0.v.weight"] = sd["0.v.weight"] * 0.5
        for how in ("load_state_dict", "inplace", "data"):
            if how == "load_state_dict":
                model.head.load_state_dict(sd)
            elif how == "inplace":
                with torch.no_grad():
                    model.head[2].weight.mul_(0.9)
            else:
                model.head[2].weight.data.mul_(1.1)                      # invisible to version counters ...
                assert eng._planes_current()
                eng.invalidate_planes()                                  # ... so the caller says so
            assert not eng._planes_current(), how
            # a fresh engine on a copy of the same head state
            ref_model = make_model(case, inp)
            ref_model.head.load_state_dict(model.head.state_dict())
            ref = ProbeHeadEngine(ref_model.head, optimizer="lars", lr=0.3)
            ref.state[0].copy_(eng.state[0])
            eng.train_step(x, t); ref.train_step(x, t)
            torch.cuda.synchronize()
            assert torch.equal(eng.flat_p, ref.flat_p), (D, how)
            assert eng._planes_current()
