"""The aux-stream schedule of the parameter-gradient contractions (csrc/ep_internal.h: AuxSide) must not change the numbers.

JEPA / CaiT / CAE / CLIP launch their weight-gradient contractions on the second stream as soon as the operands exist
(default); EP_WGRAD_EARLY=0 starts them with the second token pass (the round-3 order); EP_AUX_STREAM=0 runs everything on one
stream.  Three LARS steps in a fresh process per schedule: the same parameters to fp32 summation-order noise (the early
contractions may run on the bf16 x3 tile instead of the exact-f32 kernel: 4e-6 relative per contraction) -- a missing
event wait between the streams would show up as garbage, not as noise.  Needs an MI355X (pytest -m gpu)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import math, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from argparse import Namespace
from efficient_probing_amd import probe_heads
from efficient_probing_amd.engine import make_engine
name, out = sys.argv[1], sys.argv[2]
D, N, C, B = 768, 64, 100, 256
class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.patch_embed = Namespace(num_patches=N)
        self.head = torch.nn.Linear(D, C)
torch.manual_seed(0)
enc = Enc()
probe_heads.build_probe_head(enc, Namespace(cls_features=name, ep_queries=8, d_out=1, nb_classes=C, num_heads=12, model="vit_base_patch16"))
head = enc.head
if name == "clip":
    from efficient_probing_amd.poolings.clip import AttentionPool2d
    torch.manual_seed(0)
    head[0] = AttentionPool2d(in_features=D, feat_size=int(math.isqrt(N)))
eng = make_engine(head.to("cuda:0").train(), optimizer="lars", lr=0.2, weight_decay=1e-4)
g = torch.Generator().manual_seed(5)
for s in range(3):
    x = torch.randn(B, N, D, generator=g).to("cuda:0")
    tok = os.environ.get("EP_TEST_TOKENS", "f32")
    if tok == "bf16":
        x = x.to(torch.bfloat16)                       # bf16-STORED tokens
    elif tok == "bf16_as_f32":
        x = x.to(torch.bfloat16).float()               # the same values held as fp32
    t = torch.randint(0, C, (B,), generator=g).to("cuda:0")
    eng.train_step(x, t)
torch.cuda.synchronize()
np.savez(out, p=eng.flat_p.cpu().numpy(), mu=eng.state[0].cpu().numpy(), stats=np.array(eng.read_stats()))
'''


def run(name, env_extra):
    env = dict(os.environ, **env_extra)
    with tempfile.NamedTemporaryFile(suffix=".npz") as f:
        subprocess.run([sys.executable, "-c", CODE, name, f.name], check=True, env=env, cwd=ROOT)
        d = np.load(f.name)
        return {k: d[k] for k in d.files}


@pytest.mark.parametrize("name", ["jepa", "cait", "cae", "clip"])
def test_schedules_agree(name):
    base = run(name, {"EP_WGRAD_EARLY": "1"})
    assert np.isfinite(base["p"]).all() and base["stats"][3] == 0
    for env in ({"EP_WGRAD_EARLY": "0"}, {"EP_AUX_STREAM": "0"}, {"EP_WGRAD_EARLY": "1", "EP_AUX_B3": "0"}):
        other = run(name, env)
        for k in ("p", "mu"):
            scale = float(np.abs(base[k]).max())
            assert np.allclose(other[k], base[k], rtol=2e-4, atol=2e-5 * scale), \
                f"{name} {env} {k}: max diff {float(np.abs(other[k] - base[k]).max()):.3e} (scale {scale:.3e})"
        assert abs(other["stats"][0] - base["stats"][0]) <= 1e-4 * abs(base["stats"][0])


def test_cait_bf16_tokens_in_pass_side_work_agrees():
    """CaiT on bf16-stored tokens: its weight gradients ride in the LayerNorm-mode second pass as side workgroups (default),
    start early on the aux stream (EP_POOL_SIDE_LN=0) or run on one stream -- and equal the step on fp32 tokens holding the
    same rounded values."""
    base = run("cait", {"EP_TEST_TOKENS": "bf16"})
    assert np.isfinite(base["p"]).all() and base["stats"][3] == 0
    for env in ({"EP_TEST_TOKENS": "bf16", "EP_POOL_SIDE_LN": "0"}, {"EP_TEST_TOKENS": "bf16", "EP_AUX_STREAM": "0"},
                {"EP_TEST_TOKENS": "bf16_as_f32"}):
        other = run("cait", env)
        for k in ("p", "mu"):
            scale = float(np.abs(base[k]).max())
            assert np.allclose(other[k], base[k], rtol=2e-4, atol=2e-5 * scale), \
                f"cait {env} {k}: max diff {float(np.abs(other[k] - base[k]).max()):.3e} (scale {scale:.3e})"
