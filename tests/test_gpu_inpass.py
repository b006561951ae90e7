"""The in-pass contractions of the fused EP step (csrc/ep_inpass.h): the value projection computed inside the first token
pass (four K-quarter partials summed by the BatchNorm kernel) and its dP gradient computed inside the second one, by the
pooling workgroups themselves with per-row-block arrival counters -- against the same step with both as launches of their
own (EP_INPASS=0), in fresh processes (the switch is read once per process).  EP_INPASS is a bit mask: 1 = y inside the
first pass, 2 = dP inside the second pass (static image assignment), 4 = the TICKETED second pass (csrc/ep_pool_bwd2.hip:
images by ticket, dP at staggered task points, weight-gradient side tasks in the middle of the grid, one gradient
partial per image).

Covers the task-to-workgroup maps: one round (B <= grid), two rounds with helper workgroups (B = 1024 on 768), several
rounds without helpers (grid 256) and with a ragged helper deal (grid 320); D = 256 / 512 / 768 (KT = 1..3); bf16-stored
tokens (same kernels, other instantiation); run-to-run bit equality; the bounded flag waits never give up.

Needs an MI355X (pytest -m gpu)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys, ctypes, torch
sys.path.insert(0, os.getcwd())
from argparse import Namespace
from efficient_probing_amd import probe_heads, _native as N
from efficient_probing_amd.engine import ProbeHeadEngine
D = int(os.environ["T_D"]); B = int(os.environ["T_B"]); NT = int(os.environ["T_N"]); BF16 = os.environ.get("T_BF16", "0") == "1"
class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.head = torch.nn.Linear(D, 1000)
torch.manual_seed(0); enc = Enc()
probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=8, d_out=1, nb_classes=1000))
eng = ProbeHeadEngine(enc.head.to("cuda:0").train(), optimizer="lars", lr=0.4, weight_decay=1e-4)
g = torch.Generator().manual_seed(5)
losses, ps = [], []
for s in range(3):
    x = torch.randn(B, NT, D, generator=g).to("cuda:0"); t = torch.randint(0, 1000, (B,), generator=g).to("cuda:0")
    if BF16: x = x.to(torch.bfloat16)
    eng.train_step(x, t); losses.append(eng.read_stats()[0])
lib = N.load()
off = lib.ep_head_workspace_flag_offset(ctypes.byref(eng.dims))
giveups = int(eng._ws[off:off + 4].view(torch.int32).item())
torch.save({"loss": losses, "p": eng.flat_p.cpu(), "g": eng.flat_g.cpu(), "mu": eng.state[0].cpu(), "giveups": giveups,
            "rm": enc.head[1].running_mean.cpu()}, sys.argv[1])
'''


def run(env_extra):
    with tempfile.NamedTemporaryFile(suffix=".pt") as f:
        env = dict(os.environ)
        for k in ("EP_INPASS", "EP_POOL_GRID", "EP_BN_FOLD"):
            env.pop(k, None)
        env.update(env_extra)
        subprocess.run([sys.executable, "-c", CODE, f.name], check=True, env=env, cwd=ROOT)
        return torch.load(f.name)


CASES = [
    # (B, N, D, bf16, extra env)
    (1024, 50, 768, False, {}),                       # two rounds on 768 workgroups: helper workgroups run the early tasks
    (64, 37, 768, False, {}),                         # one round, ragged token tiles
    (1024, 33, 256, False, {}),                       # KT = 1
    (512, 40, 512, False, {}),                        # KT = 2
    (1024, 48, 768, True, {}),                        # bf16-stored tokens
    (1024, 20, 768, False, {"EP_POOL_GRID": "256"}),  # four rounds, no helpers: every workgroup runs its tasks at its end
    (1024, 20, 768, False, {"EP_POOL_GRID": "320"}),  # four rounds, 64 busy owners, 256 helpers
    (2048, 24, 768, False, {}),                       # three rounds on 768 workgroups: five later-round dP tasks per helper
                                                      # (no in-pass y: the one-launch BatchNorm takes B <= 1024)
]


@pytest.mark.parametrize("B,N,D,bf16,extra", CASES)
def test_inpass_contractions_match_the_launched_ones(B, N, D, bf16, extra):
    base = {"T_B": str(B), "T_N": str(N), "T_D": str(D), "T_BF16": "1" if bf16 else "0", **extra}
    off = run({**base, "EP_INPASS": "0"})
    assert off["giveups"] == 0
    for mask in ("3", "7"):
        on = run({**base, "EP_INPASS": mask})
        assert on["giveups"] == 0, mask
        assert np.allclose(on["loss"], off["loss"], rtol=3e-6), mask
        # same arithmetic up to the summation order of K (four quarters of y; dP is a single K = D/8 sum in both forms) and
        # of the images in the cls_token gradient (per-workgroup partials against per-image ones)
        assert torch.allclose(on["p"], off["p"], rtol=2e-4, atol=2e-6), mask
        assert torch.allclose(on["mu"], off["mu"], rtol=2e-3, atol=1e-7), mask
        assert torch.allclose(on["rm"], off["rm"], rtol=1e-5, atol=1e-7), mask


def test_each_half_alone_and_run_to_run_bit_equality():
    base = {"T_B": "1024", "T_N": "50", "T_D": "768"}
    ref = run({**base, "EP_INPASS": "0"})
    for mask in ("1", "2", "4"):
        got = run({**base, "EP_INPASS": mask})
        assert got["giveups"] == 0
        assert torch.allclose(got["p"], ref["p"], rtol=2e-4, atol=2e-6), mask
    # dP alone computes the very same sums as the launched contraction (K = 96 in one MFMA chain either way is NOT
    # guaranteed to round alike: the chains differ in length per instruction), so only closeness above; the same
    # configuration twice, however, must agree bit for bit -- no atomics feed a value, the counters only order.
    for mask in ("3", "7"):                        # (7: dynamic image assignment -- still bit-reproducible)
        a = run({**base, "EP_INPASS": mask})
        b = run({**base, "EP_INPASS": mask})
        for k in ("p", "g", "mu", "rm"):
            assert torch.equal(a[k], b[k]), (mask, k)
        assert a["loss"] == b["loss"]


@pytest.mark.parametrize("B,N,D,bf16,extra", [(1024, 50, 768, False, {}), (1024, 37, 768, True, {}), (256, 40, 768, False, {}),
                                               # a small pooling grid: the side tasks get CU slots at once and must WAIT for dy
                                               (1024, 20, 768, False, {"EP_POOL_GRID": "256"})])
def test_folded_batchnorm_backward_matches_the_separate_launch(B, N, D, bf16, extra):
    """BatchNorm1d backward folded into the in-pass dP tasks (column statistics from the epilogue of the dz contraction,
    dy formed while a task stages its A tile and published for the delta items and the dWv side tasks of the same launch)
    against ep_bn_bwd_fused_kernel in front of the pass (EP_BN_FOLD=0)."""
    base = {"T_B": str(B), "T_N": str(N), "T_D": str(D), "T_BF16": "1" if bf16 else "0", "EP_INPASS": "2", **extra}
    on = run({**base, "EP_BN_FOLD": "1"})
    off = run({**base, "EP_BN_FOLD": "0"})
    assert on["giveups"] == 0 and off["giveups"] == 0
    assert np.allclose(on["loss"], off["loss"], rtol=3e-6)
    assert torch.allclose(on["p"], off["p"], rtol=2e-4, atol=2e-6)
    assert torch.allclose(on["mu"], off["mu"], rtol=2e-3, atol=1e-7)
    again = run({**base, "EP_BN_FOLD": "1"})                    # fixed summation orders: the same bits every run
    for k in ("p", "g", "mu"):
        assert torch.equal(on[k], again[k]), k


def test_a_give_up_of_a_hand_off_wait_is_loud():
    """a bounded wait that gives up used to leave the pass reading unfinished rows and the optimizer applying the
    result silently.  Now the give-up count in the workspace makes every optimizer phase skip its update, set found_inf and
    bump the non-finite row count of the step statistics (which stops engine_finetune.train_one_epoch) -- until the
    workspace is initialised again (ep_head_workspace_init).  Simulated by poking the count."""
    import ctypes
    from argparse import Namespace
    from efficient_probing_amd import probe_heads, _native as N
    from efficient_probing_amd.engine import ProbeHeadEngine

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(256, 50)
    torch.manual_seed(0)
    enc = Enc()
    probe_heads.build_probe_head(enc, Namespace(cls_features="ep", ep_queries=8, d_out=1, nb_classes=50))
    eng = ProbeHeadEngine(enc.head.to("cuda:0").train(), optimizer="lars", lr=0.4)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 20, 256, generator=g).to("cuda:0"); t = torch.randint(0, 50, (64,), generator=g).to("cuda:0")
    eng.train_step(x, t)
    assert eng.read_stats()[3] == 0 and int(eng.found_inf.item()) == 0
    lib = N.load()
    off = lib.ep_head_workspace_flag_offset(ctypes.byref(eng.dims))
    before = eng.flat_p.clone()
    eng._ws[off:off + 4].view(torch.int32).fill_(3)                # "three waits gave up"
    eng.train_step(x, t)
    torch.cuda.synchronize()
    assert int(eng.found_inf.item()) == 1 and torch.equal(eng.flat_p, before)      # update skipped
    assert eng.read_stats()[3] >= 1                                                  # ... and the training loop would stop
    N.check(lib.ep_head_workspace_init(ctypes.byref(eng.dims), eng._ws.data_ptr(), eng._ws.numel(), N.current_stream_ptr(eng.device)), "init")
    eng.train_step(x, t)
    torch.cuda.synchronize()
    assert int(eng.found_inf.item()) == 0 and not torch.equal(eng.flat_p, before) and eng.read_stats()[3] == 0
