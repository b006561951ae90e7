"""Accuracy parity on a LEARNABLE task (BASELINE metric: "ImageNet-1k EP top-1", north_star "within +-0.1 %").

ImageNet and the backbones cannot travel to the GPU box, so the claim is checked where it can be: the native engine and
the op-for-op torch-CPU port of the reference step (oracle/torch_port.py, pinned on the reference's goldens) are trained
from the same initialisation on the same synthetic token classification task -- class-dependent token content planted at
random token positions among noise tokens, so the attentive pooling has to find it -- for 300 LARS steps under the
reference's schedule (linear warm-up then half cosine, util/lr_sched.py:3-15; lr = blr * B / 256, main_linprobe.py:572-573),
then scored on 8192 held-out samples.  The task is tuned to end at ~67 % top-1 (not saturated: a prediction flip shows).

Asserted, for fp32 tokens and for bf16-STORED tokens (the port then sees the same rounded values):
  * held-out top-1 within +-0.1 % absolute (8 of 8192 samples) of the port's,
  * the training-loss curves agree to 1e-3 at every one of the 300 steps (measured: see the printed maxima).
Needs an MI355X (pytest -m gpu); the CPU port takes ~10 s per run."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

N_TOK, DIM, Q, CLASSES, BATCH, STEPS = 64, 256, 8, 100, 256, 300
BLR, WARMUP, EPOCHS, STEPS_PER_EPOCH = 2.0, 2, 10, 30          # 10 "epochs" of 30 steps
STRENGTH, K_SIG = 0.14, 6


def make_task(n, seed):
    """n images of N_TOK noise tokens; K_SIG random positions additionally carry the class mean (x STRENGTH) plus a
    fixed 'this token is informative' direction the queries can learn to attend to."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(CLASSES, DIM, generator=torch.Generator().manual_seed(999))
    key = torch.randn(DIM, generator=torch.Generator().manual_seed(998))
    t = torch.randint(0, CLASSES, (n,), generator=g)
    x = torch.randn(n, N_TOK, DIM, generator=g)
    pos = torch.stack([torch.randperm(N_TOK, generator=g)[:K_SIG] for _ in range(n)])
    sig = STRENGTH * mu[t][:, None, :] + 1.5 * key[None, None, :] / DIM ** 0.5 * 8
    x.scatter_add_(1, pos[:, :, None].expand(-1, -1, DIM), sig.expand(-1, K_SIG, -1).contiguous())
    return x, t


def heads():
    from efficient_probing_amd import probe_heads
    from oracle import torch_port

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(DIM, CLASSES)
    torch.manual_seed(0)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=CLASSES))
    port = torch_port.make_head(DIM, Q, CLASSES)
    with torch.no_grad():
        port[0].cls_token.copy_(enc.head[0].cls_token); port[0].v.weight.copy_(enc.head[0].v.weight)
        port[2].weight.copy_(enc.head[2].weight); port[2].bias.copy_(enc.head[2].bias)
    return enc.head.to(DEV).train(), port.train()


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_trained_head_matches_the_reference_port_on_held_out_accuracy(storage):
    from efficient_probing_amd.engine import ProbeHeadEngine
    from efficient_probing_amd.util.lr_sched import lr_at, absolute_lr
    from oracle import torch_port
    xtr, ttr = make_task(BATCH * 40, 1)
    xte, tte = make_task(8192, 2)
    if storage == "bf16":                       # the stored values ARE the data: the port trains on the same rounded tokens
        xtr, xte = xtr.to(torch.bfloat16), xte.to(torch.bfloat16)
    xtr_d, ttr_d, xte_d = xtr.to(DEV), ttr.to(DEV), xte.to(DEV)
    xtr_c, xte_c = xtr.float(), xte.float()
    head, port = heads()
    base_lr = absolute_lr(BLR, BATCH)
    eng = ProbeHeadEngine(head, optimizer="lars", lr=0.0, weight_decay=0.0)
    mus = [torch.zeros_like(p) for p in port.parameters()]
    gpu_loss, cpu_loss = [], []
    for s in range(STEPS):
        lr = lr_at(s / STEPS_PER_EPOCH, base_lr, 0.0, WARMUP, EPOCHS)
        i = (s % 40) * BATCH
        eng.train_step(xtr_d[i:i + BATCH], ttr_d[i:i + BATCH], lr=lr)
        gpu_loss.append(eng.read_stats()[0])
        cpu_loss.append(float(torch_port.train_step(port, mus, xtr_c[i:i + BATCH], ttr[i:i + BATCH], lr).detach()))
    gl, cl = torch.tensor(gpu_loss), torch.tensor(cpu_loss)
    # held-out evaluation: running BatchNorm statistics, eval forward
    pred_gpu = torch.cat([eng.eval_logits(xte_d[j:j + 1024]).argmax(1).cpu() for j in range(0, len(xte), 1024)])
    port.eval()
    with torch.no_grad():
        pred_cpu = port(xte_c).argmax(1)
    acc_gpu = (pred_gpu == tte).float().mean().item() * 100
    acc_cpu = (pred_cpu == tte).float().mean().item() * 100
    dmax = float((gl - cl).abs().max())
    print(f"[{storage}] held-out top-1: native {acc_gpu:.3f} %  port {acc_cpu:.3f} %  |  prediction flips "
          f"{int((pred_gpu != pred_cpu).sum())} / {len(tte)}  |  max |loss difference| over {STEPS} steps {dmax:.2e}  "
          f"(loss {cl[0]:.3f} -> {cl[-1]:.3f})")
    assert 40.0 < acc_cpu < 95.0                      # the task is learnable and not saturated
    assert cl[-1] < 0.5 * cl[0]                       # ... and training made progress
    assert abs(acc_gpu - acc_cpu) <= 0.1              # north_star: top-1 within +-0.1 % absolute
    assert dmax <= 1e-3                               # loss curves agree at every step
    for a, b in zip(head.parameters(), [dict(port.named_parameters())[k] for k in ("0.cls_token", "0.v.weight", "2.weight", "2.bias")]):
        assert torch.allclose(a.detach().cpu(), b.detach().reshape(a.shape), rtol=1e-3, atol=1e-5)
