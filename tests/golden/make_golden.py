"""Generate tests/golden/*.npz by running the REAL reference (read-only at /root/reference)
on the deterministic inputs of ``cases.py``.

Runs only in the build container (the reference does not travel to the GPU box); the
committed ``.npz`` files are data: inputs are regenerated from seeds, expected outputs are
stored.  Usage:  python tests/golden/make_golden.py

What is imported from the reference: ``probe_heads`` (registry + build_probe_head, through
a two-package import stub for the absent ``timm``/``torchvision``), ``poolings.ep``,
``util.lars``, ``util.lr_sched``.  BatchNorm1d / Linear / CrossEntropyLoss / SGD /
GradScaler are stock torch, exactly as the reference uses them.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = os.environ.get("EP_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from cases import (CASES, EPCLS_CASES, make_epcls_inputs, INIT_DIMS, LR_POINTS, STEP_LRS, make_inputs, view_tokens, sub, keeper,   # noqa: E402
                   COCA_CASES, COCA_INIT_DIMS, COCA_PARAM_NAMES, make_coca_inputs,
                   ABMILP_CASES, ABMILP_INIT_DIMS, ABMILP_PARAM_NAMES, ABMILP_SMALL, make_abmilp_inputs,
                   KNN_CASES, KNN_GRID, make_knn_inputs,
                   SIGLIP_CASES, SIGLIP_INIT_DIMS, SIGLIP_PARAM_NAMES, SIGLIP_SMALL, make_siglip_inputs, siglip_sub,
                   CAE_CASES, CAE_INIT_DIMS, CAE_PARAM_NAMES, CAE_SMALL, make_cae_inputs,
                   JEPA_CASES, JEPA_INIT_DIMS, JEPA_PARAM_NAMES, JEPA_SMALL, make_jepa_inputs,
                   AIM_CASES, AIM_INIT_DIMS, AIM_PARAM_NAMES, AIM_SMALL, make_aim_inputs,
                   CBAM_CASES, CBAM_INIT_DIMS, CBAM_PARAM_NAMES, CBAM_SMALL, make_cbam_inputs,
                   DINOVIT_CASES, DINOVIT_INIT_DIMS, DINOVIT_PARAM_NAMES, DINOVIT_SMALL, DINOVIT_ATTN_ROWS, make_dinovit_inputs,
                   DOLG_CASES, DOLG_INIT_DIMS, DOLG_PARAM_NAMES, DOLG_SMALL, make_dolg_inputs,
                   CLIP_CASES, CLIP_INIT_DIMS, CLIP_PARAM_NAMES, CLIP_SMALL, make_clip_inputs,
                   CAIT_CASES, CAIT_INIT_DIMS, CAIT_PARAM_NAMES, CAIT_SMALL, make_cait_inputs,
                   SIMPOOL_CASES, ESIMPOOL_CASES, SIMPOOL_INIT_DIMS, SIMPOOL_SMALL, simpool_param_names, make_simpool_inputs)


def _stub_missing_packages():
    """timm / torchvision are absent here; probe_heads only needs the names to exist
    (SURVEY.md section 8c)."""
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    timm = mod("timm"); models = mod("timm.models"); vt = mod("timm.models.vision_transformer")
    layers = mod("timm.models.layers")
    timm.models = models; models.vision_transformer = vt; models.layers = layers
    vt.VisionTransformer = type("VisionTransformer", (torch.nn.Module,), {})
    vt.Mlp = type("Mlp", (torch.nn.Module,), {})
    layers.drop_path = lambda x, *a, **k: x
    layers.DropPath = type("DropPath", (torch.nn.Module,), {})
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    tv = mod("torchvision"); ops = mod("torchvision.ops"); misc = mod("torchvision.ops.misc")
    tv.ops = ops; ops.misc = misc
    misc.FrozenBatchNorm2d = type("FrozenBatchNorm2d", (torch.nn.Module,), {})


_stub_missing_packages()
import probe_heads                                   # noqa: E402  (reference)
from poolings.ep import EfficientProbing             # noqa: E402  (reference)
from util.lars import LARS                           # noqa: E402  (reference)
from util.lr_sched import adjust_learning_rate       # noqa: E402  (reference)


class StubEncoder(torch.nn.Module):
    """What build_probe_head needs from an encoder (tools/inv_heads.py:53-60)."""
    def __init__(self, dim, nb_classes):
        super().__init__()
        self.embed_dim = dim
        self.patch_embed = Namespace(num_patches=196)
        self.head = torch.nn.Linear(dim, nb_classes)


def ref_args(case_or=None, **kw):
    a = Namespace(cls_features="ep", ep_queries=32, d_out=1, nb_classes=1000, num_heads=16,
                  abmilp_sa="both", abmilp_act="tanh", abmilp_depth=2, abmilp_cond=None,
                  abmilp_content="all", model="vit_base_patch16")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def build_ref_head(dim, Q, d_out, C, cls_features="ep", **kw):
    enc = StubEncoder(dim, C)
    probe_heads.build_probe_head(enc, ref_args(cls_features=cls_features, ep_queries=Q,
                                               d_out=d_out, nb_classes=C, **kw))
    return enc.head


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def load_params(head, inp):
    with torch.no_grad():
        head[0].cls_token.copy_(torch.from_numpy(inp["cls_token"]))
        head[0].v.weight.copy_(torch.from_numpy(inp["v_weight"]))
        head[2].weight.copy_(torch.from_numpy(inp["fc_weight"]))
        head[2].bias.copy_(torch.from_numpy(inp["fc_bias"]))


def topk_acc(output, target, topk=(1, 5)):
    """timm.utils.accuracy (timm 0.9.16), restated because timm is absent:
    maxk = min(max(topk), C); topk -> compare -> correct[:k].sum() * 100 / B."""
    maxk = min(max(topk), output.size(1))
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [float(correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / target.size(0)) for k in topk]


def run_case(case, optimizer_name="lars"):
    inp = make_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, case.Q, case.d_out, case.C)
    load_params(head, inp)
    head.train()
    if optimizer_name == "lars":
        opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    else:
        opt = torch.optim.SGD(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = keeper(case)
    names = ["cls_token", "v_weight", "fc_weight", "fc_bias"]
    plist = [head[0].cls_token, head[0].v.weight, head[2].weight, head[2].bias]
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        tg = inp["targets"] if step % 2 == 0 else inp["targets2"]
        x = torch.from_numpy(view_tokens(case, xb))          # non-contiguous when strided
        t = torch.from_numpy(tg)
        lr = STEP_LRS[step % len(STEP_LRS)]
        for g in opt.param_groups:
            g["lr"] = lr
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0 and optimizer_name == "lars":
            # the published runs train under --amp bfloat16 (README.md:639-645; engine_finetune.py:52-55): the same head, same
            # inputs, train mode, under CPU bf16 autocast (Linear / matmul operands and results bf16; softmax, BatchNorm
            # statistics and the loss fp32) -- a FIDELITY fixture: how far the fp32 head is from the published protocol's head
            with torch.no_grad():
                rm, rv, nb = head[1].running_mean.clone(), head[1].running_var.clone(), head[1].num_batches_tracked.clone()
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    lg16 = head[2](head[1](head[0](x)))
                    ls16 = crit(lg16, t)
                head[1].running_mean.copy_(rm); head[1].running_var.copy_(rv); head[1].num_batches_tracked.copy_(nb)
            out["logits_bf16_autocast"] = lg16.float().numpy()
            out["loss_bf16_autocast"] = np.float32(ls16.float().item())
            a1, a5 = topk_acc(logits, t)
            attn = ((head[0].cls_token * case.D ** -0.5) @ x.transpose(1, 2)).softmax(-1)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(),
                       logits=logits.detach().numpy(), loss=np.float32(loss.item()),
                       acc1=np.float32(a1), acc5=np.float32(a5),
                       attn=attn.detach().numpy())
            for n, p in zip(names, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in ("cls_token", "fc_bias") else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        tag = f"{optimizer_name}{step + 1}"
        if optimizer_name == "lars":
            # the trust ratio util/lars.py:21-29 is about to use, twice: with torch-CPU's float32 norms (what the reference
            # computes: its naive accumulation over 1e5..1e7 elements is off by up to ~1e-3) and in float64 (the exact
            # value of the same formula).  Tests pin the GPU's ratio on the float64 one; the float32 one explains the
            # common factor between the GPU's momentum and the reference's.
            for n, p in zip(names, plist):
                if p.ndim > 1:
                    dp = p.grad.add(p.detach(), alpha=case.weight_decay)
                    pn32, un32 = torch.norm(p.detach()), torch.norm(dp)
                    pn64, un64 = torch.norm(p.detach().double()), torch.norm(dp.double())
                    out[f"{tag}_q32_{n}"] = np.float64((0.001 * pn32 / un32).item() if pn32 > 0 and un32 > 0 else 1.0)
                    out[f"{tag}_q64_{n}"] = np.float64((0.001 * pn64 / un64).item() if pn64 > 0 and un64 > 0 else 1.0)
        opt.step()
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(names, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in ("cls_token", "fc_bias") else keep(a)
            if optimizer_name == "lars":
                mu = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu if n in ("cls_token", "fc_bias") else keep(mu)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
        out[f"{tag}_nbt"] = np.int64(head[1].num_batches_tracked.item())
    if optimizer_name == "lars":
        head.eval()
        with torch.no_grad():
            x = torch.from_numpy(view_tokens(case, inp["x_buf"]))
            out["eval_logits"] = head(x).numpy()
            # the reference evaluates under torch.cuda.amp.autocast() (engine_finetune.py:131: fp16); the CPU autocast
            # of the same modules places the same roundings (Linear / matmul operands and results fp16, softmax and
            # BatchNorm statistics fp32)
            with torch.autocast("cpu", dtype=torch.float16):
                out["eval_logits_fp16_autocast"] = head(x).float().numpy()
    return out



def bf16_autocast_forward(head, x, t, crit, pool_fn=None):
    """The same head, same inputs, train mode, under CPU bf16 autocast (the published runs' --amp bfloat16, README.md:639-645;
    engine_finetune.py:52-55) -- BatchNorm buffers restored afterwards.  Returns (logits fp32 view, loss)."""
    with torch.no_grad():
        bn = head[1]
        rm, rv, nb = bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            pooled = pool_fn(x) if pool_fn is not None else head[0](x)
            lg16 = head[2](head[1](pooled))
            ls16 = crit(lg16, t)
        bn.running_mean.copy_(rm); bn.running_var.copy_(rv); bn.num_batches_tracked.copy_(nb)
    return lg16.float().numpy(), np.float32(ls16.float().item())


def run_epcls_case(case):
    """EfficientProbing.forward(x, cls=...) of the real reference (poolings/ep.py:32-33): pooled vector, and the gradients
    of the per-image queries and of v.weight under the upstream gradient ``dy`` (the learned cls_token takes none)."""
    inp = make_epcls_inputs(case)
    pool = EfficientProbing(case.D, num_queries=case.Q, d_out=case.d_out)
    with torch.no_grad():
        pool.cls_token.copy_(torch.from_numpy(inp["cls_token"]))
        pool.v.weight.copy_(torch.from_numpy(inp["v_weight"]))
    x = torch.from_numpy(view_tokens(case, inp["x_buf"]))
    cls = torch.from_numpy(inp["cls"]).requires_grad_(True)
    pooled = pool(x, cls=cls)
    pooled.backward(torch.from_numpy(inp["dy"]))
    assert pool.cls_token.grad is None
    return dict(pooled=pooled.detach().numpy(), grad_cls=cls.grad.numpy(), grad_v_weight=keeper(case)(pool.v.weight.grad.numpy()))


def coca_ref_params(head):
    p = head[0]
    return [p.norm.gamma, p.img_queries, p.to_q.weight, p.to_kv.weight, p.to_out.weight, head[2].weight, head[2].bias]


def build_coca_ref_head(case_or_dim, C, M=196):
    """Sequential(CrossAttention, BN, encoder.head) built by the reference registry for 'coca'
    (probe_heads.py:78,102-106); ``M != 196`` swaps in a CrossAttention with that many image queries."""
    dim = case_or_dim
    head = build_ref_head(dim, 32, 1, C, cls_features="coca")
    if M != 196:
        from poolings.coca_pytorch import CrossAttention
        head[0] = CrossAttention(dim=dim, num_img_queries=M)
    return head


def run_coca_case(case):
    inp = make_coca_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_coca_ref_head(case.D, case.C, case.M)
    plist = coca_ref_params(head)
    with torch.no_grad():
        for n, p in zip(COCA_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = keeper(case)
    small = ("gamma", "fc_bias")
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        tg = inp["targets"] if step % 2 == 0 else inp["targets2"]
        x = torch.from_numpy(xb[:, 1:] if case.strided else xb)
        t = torch.from_numpy(tg)
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            # round 6: the head under bf16 autocast -- the fidelity fixture of the AMP-bf16 arithmetic mode (as for EP)
            out["logits_bf16_autocast"], out["loss_bf16_autocast"] = bf16_autocast_forward(head, x, t, crit)
            a1, a5 = topk_acc(logits, t)
            # attention of image query 0, restated from coca_pytorch.py:307-330
            pool = head[0]
            with torch.no_grad():
                q = pool.to_q(pool.norm(pool.img_queries[:1])).reshape(1, pool.heads, -1) * pool.scale   # (1,H,dh)
                k = pool.to_kv(x)[..., :q.shape[-1]]                                                       # (B,N,dh)
                attn0 = torch.einsum("hd,bnd->bhn", q[0], k).softmax(-1)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5), attn0=attn0.numpy())
            for n, p in zip(COCA_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                if n == "img_queries":
                    out["grad_img_queries_row0"] = g[0].copy()
                    out["grad_img_queries_rest_absmax"] = np.float32(np.abs(g[1:]).max() if g.shape[0] > 1 else 0.0)
                else:
                    out[f"grad_{n}"] = g if n in small else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(COCA_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            mu = opt.state[p]["mu"].numpy().copy()
            out[f"{tag}_{n}"] = a if n in small else keep(a)
            out[f"{tag}_mu_{n}"] = mu if n in small else keep(mu)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        out["eval_logits"] = head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).numpy()
        # the reference's evaluation mode (torch.cuda.amp.autocast(), engine_finetune.py:131: fp16), placed by CPU autocast
        with torch.autocast("cpu", dtype=torch.float16):
            out["eval_logits_fp16_autocast"] = head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).float().numpy()
    return out


def jepa_ref_params(head):
    p = head[0]
    b = p.cross_attention_block
    return [p.query_tokens, b.norm1.weight, b.norm1.bias, b.xattn.q.weight, b.xattn.q.bias, b.xattn.kv.weight, b.xattn.kv.bias,
            b.xattn.proj.weight, b.xattn.proj.bias, b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias,
            b.mlp.fc2.weight, b.mlp.fc2.bias, head[2].weight, head[2].bias]


def build_jepa_ref_head(dim, C, heads):
    enc = StubEncoder(dim, C)
    probe_heads.build_probe_head(enc, ref_args(cls_features="jepa", nb_classes=C, num_heads=heads))
    return enc.head


def run_jepa_case(case):
    inp = make_jepa_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_jepa_ref_head(case.D, case.C, case.heads)
    plist = jepa_ref_params(head)
    with torch.no_grad():
        for n, p in zip(JEPA_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        x = torch.from_numpy(xb[:, 1:] if case.strided else xb)
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(JEPA_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in JEPA_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(JEPA_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in JEPA_SMALL else keep(a)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        out["eval_logits"] = head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).numpy()
    return out


def jepa_init_fixture():
    rec = {}
    for dim, C, heads in JEPA_INIT_DIMS:
        torch.manual_seed(0)
        head = build_jepa_ref_head(dim, C, heads)
        sd = head.state_dict()
        rec[f"d{dim}_c{C}_h{heads}"] = {"keys": {k: list(v.shape) for k, v in sd.items()},
                                        "sha256": {k: sha(v) for k, v in sd.items()},
                                        "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


def cae_ref_params(head):
    p = head[0]
    c = p.cross_attn
    return [p.query_token, p.norm1_q.weight, p.norm1_q.bias, p.norm1_k.weight, p.norm1_k.bias, p.norm1_v.weight, p.norm1_v.bias,
            p.norm2_cross.weight, p.norm2_cross.bias, c.q.weight, c.k.weight, c.v.weight, c.proj.weight, c.proj.bias,
            head[2].weight, head[2].bias]


def run_cae_case(case):
    inp = make_cae_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="cae")
    plist = cae_ref_params(head)
    with torch.no_grad():
        for n, p in zip(CAE_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        x = torch.from_numpy(xb[:, 1:] if case.strided else xb)
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(CAE_PARAM_NAMES, plist):
                if p.grad is None:                                  # norm2_cross: never used by the forward
                    out[f"grad_{n}_is_none"] = np.int32(1)
                    continue
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in CAE_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(CAE_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in CAE_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu if n in CAE_SMALL else keep(mu)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        out["eval_logits"] = head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).numpy()
    return out


def cae_init_fixture():
    rec = {}
    for dim, C in CAE_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="cae")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


def aim_ref_params(head):
    p = head[0]
    return [p.cls_token, p.k.weight, p.v.weight, head[2].weight, head[2].bias]


def run_aim_case(case):
    """--cls_features aim: the REAL AttentionPoolingClassifier (poolings/aim.py:337-392) behind BatchNorm1d + Linear."""
    inp = make_aim_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="aim", num_heads=case.heads)
    assert head[0].num_heads == case.heads
    plist = aim_ref_params(head)
    with torch.no_grad():
        for n, p in zip(AIM_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
        head[0].bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        head[0].bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(xb[:, 1:] if case.strided else xb)
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            # attention of the query token: recomputed with the batch statistics this forward used
            with torch.no_grad():
                xb = x.transpose(-2, -1)
                mu = xb.mean(dim=(0, 2), keepdim=True); var = xb.var(dim=(0, 2), unbiased=False, keepdim=True)
                xn = ((xb - mu) / torch.sqrt(var + 1e-6)).transpose(-2, -1)
                B, N, C = xn.shape
                H = case.heads
                q = head[0].cls_token.expand(B, -1, -1).reshape(B, 1, H, C // H).permute(0, 2, 1, 3) * head[0].scale
                k = head[0].k(xn).reshape(B, N, H, C // H).permute(0, 2, 1, 3)
                attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)[:, :, 0]
            out.update(pooled=pooled.detach().numpy(), attn=attn.numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(AIM_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in AIM_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(AIM_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in AIM_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu_ = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu_ if n in AIM_SMALL else keep(mu_)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
        out[f"{tag}_tok_running_mean"] = head[0].bn.running_mean.numpy().copy()
        out[f"{tag}_tok_running_var"] = head[0].bn.running_var.numpy().copy()
        out[f"{tag}_tok_nbt"] = np.int64(head[0].bn.num_batches_tracked.item())
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(view(inp["x_buf"])).numpy()
    return out


def aim_init_fixture():
    rec = {}
    for dim, C in AIM_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="aim")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters())),
                               "num_heads": int(head[0].num_heads)}
    return rec


def cbam_ref_params(head):
    p = head[0]
    return [p.channel.fc1.weight, p.channel.fc2.weight, p.spatial.conv.conv.weight, p.spatial.conv.bn.weight,
            p.spatial.conv.bn.bias, head[2].weight, head[2].bias]


def run_cbam_case(case):
    """--cls_features cbam: the REAL CbamPooling (poolings/cbam.py:104-139) behind BatchNorm1d + Linear."""
    inp = make_cbam_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="cbam")
    plist = cbam_ref_params(head)
    tbn = head[0].spatial.conv.bn
    with torch.no_grad():
        for n, p in zip(CBAM_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
        tbn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        tbn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(np.ascontiguousarray(xb[:, 1:] if case.strided else xb))    # (the reference uses .view)
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(CBAM_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in CBAM_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(CBAM_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in CBAM_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu_ = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu_ if n in CBAM_SMALL else keep(mu_)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
        out[f"{tag}_tok_running_mean"] = tbn.running_mean.numpy().copy()
        out[f"{tag}_tok_running_var"] = tbn.running_var.numpy().copy()
        out[f"{tag}_tok_nbt"] = np.int64(tbn.num_batches_tracked.item())
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(view(inp["x_buf"])).numpy()
    return out


def cbam_init_fixture():
    rec = {}
    for dim, C in CBAM_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="cbam")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


def dolg_ref_params(head):
    p = head[0]
    return [p.conv1.weight, p.conv1.bias, p.bn.weight, p.bn.bias, p.conv2.weight, p.conv2.bias, head[2].weight, head[2].bias]


def run_dolg_case(case):
    """--cls_features dolg: the REAL SpatialAttention2d (poolings/dolg/dolg.py:11-62) behind BatchNorm1d + Linear."""
    inp = make_dolg_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="dolg")
    plist = dolg_ref_params(head)
    with torch.no_grad():
        for n, p in zip(DOLG_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
        head[0].bn.running_mean.copy_(torch.from_numpy(inp["tok_running_mean"]))
        head[0].bn.running_var.copy_(torch.from_numpy(inp["tok_running_var"]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(xb[:, 1:] if case.strided else xb)
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        if step == 0:
            import copy
            probe = copy.deepcopy(head[0])                                   # return_attn on a copy: no statistics side effects
            with torch.no_grad():
                _, att = probe(x, return_attn=True)
            out["attn"] = att.numpy().reshape(x.shape[0], -1)
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(DOLG_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in DOLG_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(DOLG_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in DOLG_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu_ = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu_ if n in DOLG_SMALL else keep(mu_)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
        out[f"{tag}_tok_running_mean"] = head[0].bn.running_mean.numpy().copy()
        out[f"{tag}_tok_running_var"] = head[0].bn.running_var.numpy().copy()
        out[f"{tag}_tok_nbt"] = np.int64(head[0].bn.num_batches_tracked.item())
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(view(inp["x_buf"])).numpy()
    return out


def dolg_init_fixture():
    rec = {}
    for dim, C in DOLG_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="dolg")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


def dinovit_ref_params(head):
    b = head[0].dino_block
    return [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.proj.weight, b.attn.proj.bias, b.norm2.weight, b.norm2.bias,
            b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias, head[2].weight, head[2].bias]


def run_dinovit_case(case):
    """--cls_features dinovit: the REAL DinoViTBlockPooling (poolings/other_pool.py:299-318 around dinov2_layers/block.py:43-113)
    behind BatchNorm1d + Linear."""
    inp = make_dinovit_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="dinovit")
    plist = dinovit_ref_params(head)
    with torch.no_grad():
        for n, p in zip(DINOVIT_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(xb[:, 1:] if case.strided else xb)
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        if step == 0:
            with torch.no_grad():                                            # the block's own return_attention path (block.py:91-92)
                _, att = head[0].dino_block(x, return_attention=True)
            out["attn"] = att.numpy() if case.full else att.numpy()[:, :, ::DINOVIT_ATTN_ROWS]
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(DINOVIT_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in DINOVIT_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(DINOVIT_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in DINOVIT_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu_ = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu_ if n in DINOVIT_SMALL else keep(mu_)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(view(inp["x_buf"])).numpy()
    return out


def dinovit_init_fixture():
    rec = {}
    for dim, C in DINOVIT_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="dinovit")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


def clip_ref_params(head):
    p = head[0]
    return [p.pos_embed, p.qkv.weight, p.qkv.bias, p.proj.weight, p.proj.bias, p.norm.weight, p.norm.bias, head[2].weight,
            head[2].bias]


def run_clip_case(case):
    """--cls_features clip: the REAL AttentionPool2d (poolings/clip/attention_pool2d.py:100-169) behind BatchNorm1d + Linear."""
    inp = make_clip_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="clip", model=case.model)
    assert head[0].pos_embed.shape[0] == case.N + 1
    plist = clip_ref_params(head)
    with torch.no_grad():
        for n, p in zip(CLIP_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(xb[:, 1:] if case.strided else xb)
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            with torch.no_grad():
                _, attn = head[0](x, return_attn=True)                      # attn[:, :, 0, 1:]  (B, H, N)
            out.update(pooled=pooled.detach().numpy(), attn=attn.numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(CLIP_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in CLIP_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(CLIP_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in CLIP_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu_ = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu_ if n in CLIP_SMALL else keep(mu_)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(view(inp["x_buf"])).numpy()
    return out


def clip_init_fixture():
    rec = {}
    for dim, C in CLIP_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="clip")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


def cait_ref_params(head):
    p = head[0]
    b = p.blocks_token_only[0]
    a, m = b.attn, b.mlp
    return [p.cls_token, b.gamma_1, b.gamma_2, b.norm1.weight, b.norm1.bias, a.q.weight, a.q.bias, a.k.weight, a.k.bias,
            a.v.weight, a.v.bias, a.proj.weight, a.proj.bias, b.norm2.weight, b.norm2.bias, m.fc1.weight, m.fc1.bias,
            m.fc2.weight, m.fc2.bias, p.norm.weight, p.norm.bias, head[2].weight, head[2].bias]


def run_cait_case(case):
    """--cls_features cait: the REAL CAPooling (poolings/other_pool.py:390-507) behind BatchNorm1d + Linear."""
    inp = make_cait_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="cait")
    plist = cait_ref_params(head)
    with torch.no_grad():
        for n, p in zip(CAIT_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(xb[:, 1:] if case.strided else xb)
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), z=z.detach().numpy(), logits=logits.detach().numpy(),
                       loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(CAIT_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in CAIT_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(CAIT_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in CAIT_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu_ = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu_ if n in CAIT_SMALL else keep(mu_)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(view(inp["x_buf"])).numpy()
    return out


def cait_init_fixture():
    rec = {}
    for dim, C in CAIT_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="cait")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


class _cuda_tensors_on_cpu:
    """The reference SimPool constructors create ``torch.tensor([1e-6], device='cuda')`` (simpool.py:21,109) -- a constant
    their forward only touches when gamma is set.  No GPU here: build those tensors on the CPU instead."""

    def __enter__(self):
        self._orig = torch.tensor

        def tensor(*a, **kw):
            if str(kw.get("device", "")).startswith("cuda"):
                kw["device"] = "cpu"
            return self._orig(*a, **kw)
        torch.tensor = tensor

    def __exit__(self, *exc):
        torch.tensor = self._orig


def simpool_ref_params(head, case):
    p = head[0]
    return [p.norm_patches.weight, p.norm_patches.bias] + ([p.wq.weight, p.wk.weight] if case.linears else []) + \
        [head[2].weight, head[2].bias]


def run_simpool_case(case):
    """--cls_features simpool / esimpool: the REAL SimPool / SimPool_nolinears (poolings/simpool.py) behind BN + Linear."""
    inp = make_simpool_inputs(case)
    names = simpool_param_names(case)
    out = {}
    torch.manual_seed(0)
    with _cuda_tensors_on_cpu():
        head = build_ref_head(case.D, 32, 1, case.C, cls_features=case.family)
    assert head[0].num_heads == case.heads
    plist = simpool_ref_params(head, case)
    with torch.no_grad():
        for n, p in zip(names, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    view = lambda xb: torch.from_numpy(xb[:, 1:] if case.strided else xb)
    for step in range(case.steps):
        x = view(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            with torch.no_grad():
                _, attn = head[0](x, return_attn=True)                      # (B, H, 1, N)
            out.update(pooled=pooled.detach().numpy(), attn=attn.numpy()[:, :, 0], z=z.detach().numpy(),
                       logits=logits.detach().numpy(), loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(names, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in SIMPOOL_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(names, plist):
            a = p.detach().numpy().copy()
            out[f"{tag}_{n}"] = a if n in SIMPOOL_SMALL else keep(a)
            if "mu" in opt.state[p]:
                mu_ = opt.state[p]["mu"].numpy().copy()
                out[f"{tag}_mu_{n}"] = mu_ if n in SIMPOOL_SMALL else keep(mu_)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(view(inp["x_buf"])).numpy()
    return out


def simpool_init_fixture():
    rec = {}
    for fam in ("simpool", "esimpool"):
        for dim, C in SIMPOOL_INIT_DIMS:
            torch.manual_seed(0)
            with _cuda_tensors_on_cpu():
                head = build_ref_head(dim, 32, 1, C, cls_features=fam)
            sd = head.state_dict()
            rec[f"{fam}_d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()},
                                         "sha256": {k: sha(v) for k, v in sd.items()},
                                         "n_trainable": int(sum(p.numel() for p in head.parameters())),
                                         "num_heads": int(head[0].num_heads)}
    return rec


def siglip_ref_params(head):
    p = head[0]
    return [p.latent, p.q.weight, p.q.bias, p.kv.weight, p.kv.bias, p.proj.weight, p.proj.bias, p.mlp.fc1.weight,
            p.mlp.fc1.bias, p.mlp.fc2.weight, p.mlp.fc2.bias, head[2].weight, head[2].bias]


def run_siglip_case(case):
    inp = make_siglip_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_ref_head(case.D, 32, 1, case.C, cls_features="siglip")
    plist = siglip_ref_params(head)
    with torch.no_grad():
        for n, p in zip(SIGLIP_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = (lambda a: a) if case.full else siglip_sub
    for step in range(case.steps):
        xb = inp["x_buf"] if step % 2 == 0 else inp["x_buf2"]
        x = torch.from_numpy(xb[:, 1:] if case.strided else xb)
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        if step == 0:
            pooled, attn = head[0](x, return_attn=True)
        else:
            pooled = head[0](x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), attn=attn.detach().numpy()[:, :, 0, :], z=z.detach().numpy(),
                       logits=logits.detach().numpy(), loss=np.float32(loss.item()), acc1=np.float32(a1), acc5=np.float32(a5))
            for n, p in zip(SIGLIP_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in SIGLIP_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(SIGLIP_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            mu = opt.state[p]["mu"].numpy().copy()
            out[f"{tag}_{n}"] = a if n in SIGLIP_SMALL else keep(a)
            out[f"{tag}_mu_{n}"] = mu if n in SIGLIP_SMALL else keep(mu)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        xb = inp["x_buf"]
        out["eval_logits"] = head(torch.from_numpy(xb[:, 1:] if case.strided else xb)).numpy()
    return out


def siglip_init_fixture():
    rec = {}
    for dim, C in SIGLIP_INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, 32, 1, C, cls_features="siglip")
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {"keys": {k: list(v.shape) for k, v in sd.items()}, "sha256": {k: sha(v) for k, v in sd.items()},
                               "n_trainable": int(sum(p.numel() for p in head.parameters()))}
    return rec


def abmilp_ref_params(head):
    p = head[0]
    return [p.self_attn.qkv.weight, p.self_attn.proj.weight, p.self_attn.proj.bias, p.attention_predictor[0].weight,
            p.attention_predictor[0].bias, p.attention_predictor[2].weight, p.attention_predictor[2].bias,
            head[2].weight, head[2].bias]


def build_abmilp_ref_head(dim, C, content="all"):
    enc = StubEncoder(dim, C)
    probe_heads.build_probe_head(enc, ref_args(cls_features="abmilp", nb_classes=C, abmilp_content=content))
    return enc.head


def run_abmilp_case(case):
    inp = make_abmilp_inputs(case)
    out = {}
    torch.manual_seed(0)
    head = build_abmilp_ref_head(case.D, case.C, case.content)
    plist = abmilp_ref_params(head)
    with torch.no_grad():
        for n, p in zip(ABMILP_PARAM_NAMES, plist):
            p.copy_(torch.from_numpy(inp[n]))
    head.train()
    opt = LARS(head.parameters(), lr=0.0, weight_decay=case.weight_decay)
    crit = torch.nn.CrossEntropyLoss()
    keep = keeper(case)
    for step in range(case.steps):
        x = torch.from_numpy(inp["x_buf"] if step % 2 == 0 else inp["x_buf2"])
        t = torch.from_numpy(inp["targets"] if step % 2 == 0 else inp["targets2"])
        for g in opt.param_groups:
            g["lr"] = STEP_LRS[step % len(STEP_LRS)]
        opt.zero_grad()
        pooled, amap = head[0].forward_with_attn_map(x)
        z = head[1](pooled)
        logits = head[2](z)
        loss = crit(logits, t)
        loss.backward()
        if step == 0:
            out["logits_bf16_autocast"], out["loss_bf16_autocast"] = bf16_autocast_forward(head, x, t, crit)
            a1, a5 = topk_acc(logits, t)
            out.update(pooled=pooled.detach().numpy(), attn_map=amap.detach().numpy(), z=z.detach().numpy(),
                       logits=logits.detach().numpy(), loss=np.float32(loss.item()), acc1=np.float32(a1),
                       acc5=np.float32(a5))
            for n, p in zip(ABMILP_PARAM_NAMES, plist):
                g = p.grad.detach().numpy()
                out[f"grad_{n}"] = g if n in ABMILP_SMALL else keep(g)
                out[f"gradnorm_{n}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        tag = f"lars{step + 1}"
        out[f"{tag}_loss"] = np.float32(loss.item())
        for n, p in zip(ABMILP_PARAM_NAMES, plist):
            a = p.detach().numpy().copy()
            mu = opt.state[p]["mu"].numpy().copy()
            out[f"{tag}_{n}"] = a if n in ABMILP_SMALL else keep(a)
            out[f"{tag}_mu_{n}"] = mu if n in ABMILP_SMALL else keep(mu)
        out[f"{tag}_running_mean"] = head[1].running_mean.numpy().copy()
        out[f"{tag}_running_var"] = head[1].running_var.numpy().copy()
    head.eval()
    with torch.no_grad():
        out["eval_logits"] = head(torch.from_numpy(inp["x_buf"])).numpy()
        # the reference's evaluation mode (torch.cuda.amp.autocast(), engine_finetune.py:131: fp16), placed by CPU autocast
        with torch.autocast("cpu", dtype=torch.float16):
            out["eval_logits_fp16_autocast"] = head(torch.from_numpy(inp["x_buf"])).float().numpy()
    return out


def abmilp_init_fixture():
    rec = {}
    for dim, C in ABMILP_INIT_DIMS:
        torch.manual_seed(0)
        head = build_abmilp_ref_head(dim, C)
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {
            "keys": {k: list(v.shape) for k, v in sd.items()},
            "sha256": {k: sha(v) for k, v in sd.items()},
            "n_trainable": int(sum(p.numel() for p in head.parameters())),
        }
    return rec


def coca_init_fixture():
    rec = {}
    for dim, C in COCA_INIT_DIMS:
        torch.manual_seed(0)
        head = build_coca_ref_head(dim, C)
        sd = head.state_dict()
        rec[f"d{dim}_c{C}"] = {
            "repr_pool_class": type(head[0]).__name__,
            "keys": {k: list(v.shape) for k, v in sd.items()},
            "sha256": {k: sha(v) for k, v in sd.items()},
            "head8": {k: v.flatten()[:8].double().tolist() for k, v in sd.items()},
            "n_trainable": int(sum(p.numel() for p in head.parameters())),
        }
    return rec


def init_fixture():
    """Initial weights of the head under torch.manual_seed(0), the way
    tools/inv_heads.py:102-120 fingerprints them."""
    rec = {}
    for dim, Q, d_out, C in INIT_DIMS:
        torch.manual_seed(0)
        head = build_ref_head(dim, Q, d_out, C)
        key = f"d{dim}_q{Q}_o{d_out}_c{C}"
        sd = head.state_dict()
        rec[key] = {
            "repr": repr(head),
            "keys": {k: list(v.shape) for k, v in sd.items()},
            "sha256": {k: sha(v) for k, v in sd.items()},
            "head8": {k: v.flatten()[:8].double().tolist() for k, v in sd.items()},
            "n_trainable": int(sum(p.numel() for p in head.parameters())),
        }
    return rec


def lr_fixture():
    rows = []
    for ep, lr, min_lr, warm, epochs in LR_POINTS:
        opt = Namespace(param_groups=[{"lr": -1.0}, {"lr": -1.0, "lr_scale": 0.5}])
        got = adjust_learning_rate(opt, ep, Namespace(lr=lr, min_lr=min_lr, warmup_epochs=warm, epochs=epochs))
        rows.append(dict(epoch=ep, lr=lr, min_lr=min_lr, warmup=warm, epochs=epochs, out=got,
                         group0=opt.param_groups[0]["lr"], group1=opt.param_groups[1]["lr"]))
    return rows


def lars_edge_fixture():
    """util/lars.py:26-29 edge cases: zero update norm, zero param norm, 1-D tensors."""
    out = {}
    rng = np.random.default_rng(7)
    p2 = torch.nn.Parameter(torch.from_numpy(rng.standard_normal((5, 6), dtype=np.float32)))
    pz = torch.nn.Parameter(torch.zeros(4, 3))                      # ||p|| == 0 -> ratio 1
    pg0 = torch.nn.Parameter(torch.from_numpy(rng.standard_normal((3, 4), dtype=np.float32)))  # grad 0
    p1 = torch.nn.Parameter(torch.from_numpy(rng.standard_normal((9,), dtype=np.float32)))     # 1-D
    p3 = torch.nn.Parameter(torch.from_numpy(rng.standard_normal((1, 2, 8), dtype=np.float32)))  # ndim 3
    ps = [p2, pz, pg0, p1, p3]
    gs = [rng.standard_normal(tuple(p.shape), dtype=np.float32) for p in ps]
    gs[2] = np.zeros_like(gs[2])
    opt = LARS(ps, lr=0.5, weight_decay=0.0)
    for i, (p, g) in enumerate(zip(ps, gs)):
        out[f"p{i}_before"] = p.detach().numpy().copy()
        out[f"g{i}"] = g
    for step in range(2):
        for p, g in zip(ps, gs):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
        for i, p in enumerate(ps):
            out[f"p{i}_after{step + 1}"] = p.detach().numpy().copy()
            out[f"mu{i}_after{step + 1}"] = opt.state[p]["mu"].numpy().copy()
    # weight decay variant
    ps2 = [torch.nn.Parameter(torch.from_numpy(out[f"p{i}_before"].copy())) for i in range(5)]
    opt2 = LARS(ps2, lr=0.5, weight_decay=0.01)
    for p, g in zip(ps2, gs):
        p.grad = torch.from_numpy(g.copy())
    opt2.step()
    for i, p in enumerate(ps2):
        out[f"wd_p{i}_after1"] = p.detach().numpy().copy()
    return out


def scaler_fixture():
    """torch GradScaler (util/misc.py:263-277) scale trajectory with injected overflows."""
    sc = torch.amp.GradScaler("cpu", init_scale=65536.0, growth_interval=4)
    p = torch.nn.Parameter(torch.ones(3))
    opt = torch.optim.SGD([p], lr=0.1)
    traj, stepped = [], []
    inf_at = {2, 3, 9}
    for i in range(14):
        opt.zero_grad()
        coef = torch.full((3,), float("inf") if i in inf_at else 1.0)
        before = p.detach().clone()
        sc.scale((p * coef).sum()).backward()
        sc.unscale_(opt)
        sc.step(opt)
        sc.update()
        traj.append(sc.get_scale())
        stepped.append(bool((p.detach() != before).any()))
    return dict(scale=traj, stepped=stepped, inf_at=sorted(inf_at), growth_interval=4)


def knn_fixture():
    """(top1, top5) of the REAL reference knn_classifier (engine_finetune.py:224-266, CPU path) on seeded inputs.
    engine_finetune imports the training stack (timm.data / timm.utils / backbones); none of that is used by the
    function, so the absent packages are replaced by permissive name-only stubs for this import."""
    class AnyNames(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (torch.nn.Module,), {})
    for n in ["timm.data", "timm.data.mixup", "timm.utils", "wandb", "tensorboard", "timm.models.layers", "timm.layers",
              "timm.models.registry", "timm.models.vision_transformer", "torch.utils.tensorboard", "timm.optim",
              "timm.optim.optim_factory", "timm.loss", "timm.models.layers.helpers"]:
        m = AnyNames(n); m.__path__ = []; sys.modules[n] = m
    sys.modules["timm.models.layers"].to_2tuple = lambda x: (x, x)
    sys.modules["timm.models.layers"].trunc_normal_ = torch.nn.init.trunc_normal_
    import engine_finetune as EF                            # reference
    out = {}
    for name in KNN_CASES:
        inp = make_knn_inputs(name)
        tr, te = torch.from_numpy(inp["train"]), torch.from_numpy(inp["test"])
        ltr, lte = torch.from_numpy(inp["train_labels"]), torch.from_numpy(inp["test_labels"])
        rows = []
        for k, T in KNN_GRID:
            t1, t5 = EF.knn_classifier(tr, ltr, te, lte, k, T, use_cuda=False, num_classes=inp["C"], num_chunks=4)
            rows.append(dict(k=k, T=T, top1=float(t1), top5=float(t5)))
        out[name] = rows
    return out


def main():
    """``python make_golden.py`` regenerates everything; ``python make_golden.py aim [jepa ...]`` only the named families
    (their .npz files and their entry of host_fixtures.json)."""
    only = {a.split(":")[0] for a in sys.argv[1:]}
    # ``fam:case`` regenerates one case of a family and leaves the family's other committed files alone
    only_cases = {tuple(a.split(":", 1)) for a in sys.argv[1:] if ":" in a}
    want = lambda fam: not only or fam in only
    want_case = lambda fam, case: not any(f == fam for f, _ in only_cases) or (fam, case.name) in only_cases
    meta = {"torch": torch.__version__, "reference": REF, "cases": [c.name for c in CASES]}

    amend = os.environ.get("EP_GOLDEN_AMEND") == "1"

    def dump(prefix, case, out):
        path = os.path.join(HERE, f"{prefix}_{case.name}.npz")
        if amend and os.path.exists(path):
            # EP_GOLDEN_AMEND=1: keep every committed array as it is and only ADD the keys a newer generator records; the
            # re-run must reproduce what is committed (same reference, same inputs)
            old = dict(np.load(path))
            for k, v in old.items():
                if k in out and np.asarray(v).dtype.kind == "f":
                    np.testing.assert_allclose(out[k], v, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(v).max())), err_msg=f"{prefix}_{case.name}:{k}")
            new = [k for k in out if k not in old]
            old.update({k: out[k] for k in new})
            out = old
            print(f"{prefix}_{case.name}: amended with {new}")
        np.savez_compressed(path, **out)
        print(f"{prefix}_{case.name}: {len(out)} arrays -> {os.path.getsize(path) / 1024:.0f} KiB")

    if want("ep"):
        for case in CASES:
            if not want_case("ep", case):
                continue
            out = run_case(case, "lars")
            if case.full:
                sgd = run_case(case, "sgd")
                out.update({k: v for k, v in sgd.items() if k.startswith("sgd")})
            dump("ep", case, out)
    for fam, cases, run in (("epcls", EPCLS_CASES, run_epcls_case), ("coca", COCA_CASES, run_coca_case), ("aim", AIM_CASES, run_aim_case),
                            ("jepa", JEPA_CASES, run_jepa_case), ("cae", CAE_CASES, run_cae_case),
                            ("siglip", SIGLIP_CASES, run_siglip_case), ("abmilp", ABMILP_CASES, run_abmilp_case),
                            ("simpool", SIMPOOL_CASES, run_simpool_case), ("esimpool", ESIMPOOL_CASES, run_simpool_case),
                            ("cait", CAIT_CASES, run_cait_case), ("clip", CLIP_CASES, run_clip_case),
                            ("dolg", DOLG_CASES, run_dolg_case), ("cbam", CBAM_CASES, run_cbam_case),
                            ("dinovit", DINOVIT_CASES, run_dinovit_case)):
        if want(fam):
            for case in cases:
                if want_case(fam, case):
                    dump(fam, case, run(case))
    hp = os.path.join(HERE, "host_fixtures.json")
    if only:
        with open(hp) as f:
            host = json.load(f)
    else:
        np.savez_compressed(os.path.join(HERE, "lars_edges.npz"), **lars_edge_fixture())
        host = dict(meta=meta, lr=lr_fixture(), scaler=scaler_fixture())
    for fam, key, fn in (("ep", "init", init_fixture), ("coca", "coca_init", coca_init_fixture),
                         ("abmilp", "abmilp_init", abmilp_init_fixture), ("siglip", "siglip_init", siglip_init_fixture),
                         ("cae", "cae_init", cae_init_fixture), ("jepa", "jepa_init", jepa_init_fixture),
                         ("aim", "aim_init", aim_init_fixture), ("simpool", "simpool_init", simpool_init_fixture),
                         ("cait", "cait_init", cait_init_fixture), ("clip", "clip_init", clip_init_fixture),
                         ("dolg", "dolg_init", dolg_init_fixture), ("cbam", "cbam_init", cbam_init_fixture),
                         ("dinovit", "dinovit_init", dinovit_init_fixture)):
        if want(fam):
            host[key] = fn()
    with open(hp, "w") as f:
        json.dump(host, f, indent=1, sort_keys=True)
    if want("knn"):
        with open(os.path.join(HERE, "knn_fixtures.json"), "w") as f:
            json.dump(knn_fixture(), f, indent=1, sort_keys=True)
    print("wrote host_fixtures.json" + ("" if only else ", lars_edges.npz") + (", knn_fixtures.json" if want("knn") else ""))


if __name__ == "__main__":
    main()
