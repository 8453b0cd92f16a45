"""vdb_search_dev_noalloc: the search entry point that allocates nothing (its buffers live in a fixed
array inside the library) returns what vdb_search_dev returns, bit for bit, on two calls in a row,
the second one smaller than the first, and refuses what does not fit its scratch array."""
import numpy as np
import pytest

from vectordb_amd import _lib
from vectordb_amd._lib import VdbError

pytestmark = pytest.mark.gpu


def _both(engine, torch, name, nq, d, k, seed):
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((nq, d), dtype=torch.float32, device=dev)
    engine.synth_fill_dev(q.data_ptr(), nq, d, seed=seed, first_row=0, stream=st)
    outs = []
    for fn in (engine.search_dev, engine.search_dev_noalloc):
        ids = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
        sc = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
        fn(name, nq, q.data_ptr(), k, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
        torch.cuda.synchronize()
        outs.append((ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()))
    (i0, s0, c0), (i1, s1, c1) = outs
    assert np.array_equal(c0, c1)
    assert np.array_equal(i0, i1)
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32))
    assert int(c0.min()) == k
    return i1, s1


def test_noalloc_search_matches_search_dev_twice_in_a_row(engine):
    import torch
    big, small = "na_big", "na_small"
    for n in (big, small):
        if engine.has_table(n):
            engine.drop_table(n)
    try:
        d = 768
        engine.create_table(big, d, _lib.METRIC_COSINE, _lib.F32)
        engine.insert_synthetic(big, 300_000, seed=4, first_row=0)
        engine.create_table(small, 128, _lib.METRIC_L2, _lib.BF16)
        engine.insert_synthetic(small, 9_000, seed=5, first_row=0)
        # first call: 4 queries over the large table; second call, right after it: 1 query over a
        # smaller table with a longer candidate list (the scratch array keeps the first call's contents)
        _both(engine, torch, big, 4, d, 10, seed=21)
        _both(engine, torch, small, 1, 128, 100, seed=22)
        # and the first shape once more
        _both(engine, torch, big, 2, d, 10, seed=23)
        # directly against the oracle as well (not only against the other entry point)
        from oracle import oracle as O
        got_ids, got_sc = _both(engine, torch, small, 2, 128, 10, seed=24)
        stored = O.prepare_store(O.synth_rows(5, 0, 9_000, 128), O.L2, O.BF16)
        want = O.search(stored, np.arange(9_000, dtype=np.int64), O.synth_rows(24, 0, 2, 128), 10, O.L2)
        for qi, (wi, ws) in enumerate(want):
            assert np.array_equal(got_ids[qi], wi)
            assert np.array_equal(got_sc[qi].view(np.uint32), ws.view(np.uint32))
        # what the entry point is for: warm calls take no device memory (nothing allocated, nothing freed)
        dev = torch.device("cuda", 0)
        st = torch.cuda.current_stream().cuda_stream
        q = torch.empty((2, d), dtype=torch.float32, device=dev)
        engine.synth_fill_dev(q.data_ptr(), 2, d, seed=25, first_row=0, stream=st)
        o_ids = torch.zeros((2, 10), dtype=torch.int64, device=dev)
        o_sc = torch.zeros((2, 10), dtype=torch.float32, device=dev)
        o_cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for _ in range(5):
            engine.search_dev_noalloc(big, 2, q.data_ptr(), 10, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0
        # beyond the scratch array: refused with a clear error, nothing allocated
        dev = torch.device("cuda", 0)
        nq, k = 64, 10
        q = torch.zeros((nq, d), dtype=torch.float32, device=dev)
        ids = torch.zeros((nq, k), dtype=torch.int64, device=dev)
        sc = torch.zeros((nq, k), dtype=torch.float32, device=dev)
        cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
        with pytest.raises(VdbError) as ei:
            engine.search_dev_noalloc(big, nq, q.data_ptr(), k, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), 0)
        assert ei.value.code == _lib.EINVAL and "scratch" in ei.value.message
    finally:
        for n in (big, small):
            if engine.has_table(n):
                engine.drop_table(n)
