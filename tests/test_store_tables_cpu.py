"""Host logic of the resident store's cached tables (no GPU): a StoreBatch names its owning store, ``engine_finetune`` attaches
it to the engine once, and an engine binds the tables only for batches that really are read in place from that store."""
import torch

from efficient_probing_amd import engine_finetune as EF
from efficient_probing_amd import token_store as TS
from efficient_probing_amd.engine import ProbeHeadEngine


class FakeStore:
    def __init__(self, n=6):
        self.tokens = torch.zeros(n, 3, 4)
        self.asked = []

    def table(self, kind, eps=None):
        self.asked.append((kind, eps))
        return torch.zeros(self.tokens.shape[0], 3, 2)


class FakeEngine:
    _store_kinds = {"token_stats": ("token_stats", 1e-6)}
    attach_store = ProbeHeadEngine.attach_store
    _bind_store_tables = ProbeHeadEngine._bind_store_tables
    _table_ptr = ProbeHeadEngine._table_ptr


def test_store_batch_is_a_three_tuple_that_names_its_store():
    st = FakeStore()
    b = TS.StoreBatch(st.tokens, torch.zeros(2, dtype=torch.int32), torch.zeros(2, dtype=torch.int64), owner=st)
    x, idx, y = b
    assert len(b) == 3 and b.owner is st and b[0] is x and b[-1] is y
    assert TS.StoreBatch(x, idx, y).owner is None


def test_attach_happens_once_and_only_for_store_batches():
    st, eng = FakeStore(), FakeEngine()
    EF._attach_store(eng, (st.tokens, torch.zeros(2)))                       # an ordinary loader batch
    assert getattr(eng, "_store", None) is None
    b = TS.StoreBatch(st.tokens, torch.zeros(2, dtype=torch.int32), torch.zeros(2, dtype=torch.int64), owner=st)
    EF._attach_store(eng, b); EF._attach_store(eng, b)
    assert eng._store is st
    EF._attach_store(None, b)                                                # no fused engine: nothing to attach to


def test_tables_are_bound_only_for_in_place_batches_of_the_attached_store():
    st, eng = FakeStore(), FakeEngine()
    eng.attach_store(st)
    idx = torch.zeros(2, dtype=torch.int32)
    eng._bind_store_tables(st.tokens, None)                                   # contiguous batch: the step computes its own
    assert eng._table_ptr("token_stats") == 0 and st.asked == []
    eng._bind_store_tables(torch.zeros(6, 3, 4), idx)                         # indexed, but another tensor
    assert eng._table_ptr("token_stats") == 0 and st.asked == []
    eng._bind_store_tables(st.tokens, idx)
    assert eng._table_ptr("token_stats") != 0 and st.asked == [("token_stats", 1e-6)]
    assert eng._table_ptr("image_stats") == 0                                 # not a table this engine's kernels take
    explicit = torch.ones(6, 3, 2)
    eng._tokstat = explicit                                                   # a per-call table wins
    assert eng._table_ptr("token_stats") == explicit.data_ptr()
    eng._tokstat = None
    eng._bind_store_tables(st.tokens, None)                                   # the next contiguous call starts clean
    assert eng._table_ptr("token_stats") == 0


def test_a_view_of_the_store_with_another_token_count_does_not_bind_the_tables():
    """Round 4 advisor finding: a view with the store's base pointer and image count but fewer tokens (``store.tokens[:, :K]``)
    used to bind the (n, N_store, 2) tables, which the kernels would then read with the view's token stride."""
    st, eng = FakeStore(), FakeEngine()
    eng.attach_store(st)
    idx = torch.zeros(2, dtype=torch.int32)
    view = st.tokens[:, :2]                                                   # same data_ptr, same shape[0], 2 of 3 tokens
    assert view.data_ptr() == st.tokens.data_ptr() and view.shape[0] == st.tokens.shape[0]
    eng._bind_store_tables(view, idx)
    assert eng._table_ptr("token_stats") == 0 and st.asked == []
    eng._bind_store_tables(st.tokens[:, :, :2], idx)                          # narrower rows
    assert eng._table_ptr("token_stats") == 0 and st.asked == []
    eng._bind_store_tables(st.tokens, idx)                                    # the store itself still binds
    assert eng._table_ptr("token_stats") != 0
