"""include/ep_hip.h says the entry points only enqueue work on the caller's stream and never synchronise the device, so a
caller can capture a step into a hipGraph.  VERDICT r5 (weak 9): prove it or delete it.  This captures ONE whole train step
(ep_head_train_step, phases = 3: both token passes, the contractions between them, loss, optimizer) through torch's
graph-capture API and replays it: the replayed steps must leave the bits the eager steps leave.  What a captured step can NOT
do is follow a learning-rate schedule -- lr, opt_step and planes_valid are passed by value, so a graph replays the step with
ITS hyper-parameters (documented in include/ep_hip.h); the test therefore steps with a constant lr.  Needs an MI355X."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(D, Q, C, opt):
    from efficient_probing_amd import probe_heads
    from efficient_probing_amd.engine import ProbeHeadEngine

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = torch.nn.Linear(D, C)
    torch.manual_seed(0)
    e = Enc()
    probe_heads.build_probe_head(e, Namespace(cls_features="ep", ep_queries=Q, d_out=1, nb_classes=C))
    return ProbeHeadEngine(e.head.to(DEV).train(), optimizer=opt, lr=0.2)


@pytest.mark.parametrize("shape", [(64, 50, 768, 8, 100), (32, 37, 128, 4, 10), (64, 64, 1024, 8, 50)],
                         ids=["valu-768-inpass", "generic-128", "mfma-1024"])
@pytest.mark.parametrize("opt", ["lars", "sgd"])
def test_a_captured_step_replays_bit_identically(shape, opt):
    B, Nn, D, Q, C = shape
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, Nn, D, device=DEV, generator=g)
    t = torch.randint(0, C, (B,), device=DEV, generator=g)
    eager, cap = _engine(D, Q, C, opt), _engine(D, Q, C, opt)
    n_replay = 3
    for _ in range(1 + n_replay):
        eager.train_step(x, t, lr=0.2)
    # one eager step first: it establishes the weight planes (planes_valid = 1 from then on), creates the library's events and
    # sets the kernels' LDS attributes -- none of which may happen for the first time inside a capture
    cap.train_step(x, t, lr=0.2)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            cap.train_step(x, t, lr=0.2)
    torch.cuda.current_stream().wait_stream(side)
    # (capturing enqueues nothing: the captured step has not run yet)
    for _ in range(n_replay):
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager.params_list, cap.params_list):
        assert torch.equal(a, b)
    assert torch.equal(eager.bn.running_mean, cap.bn.running_mean) and torch.equal(eager.bn.running_var, cap.bn.running_var)
    assert int(cap.bn.num_batches_tracked) == 1 + n_replay
    le, lc = eager.read_stats(), cap.read_stats()
    assert le == lc
