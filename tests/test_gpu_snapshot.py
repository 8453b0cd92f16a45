"""Table snapshots (db_config.primary_path / preload_table of /root/reference/server_config.yaml:12,22):
a table that went through save -> drop -> load answers searches exactly like the oracle does on
the original rows, and damaged files are refused."""
import os

import numpy as np
import pytest

import cases
from oracle import oracle as O
from vectordb_amd import _lib
from vectordb_amd._lib import VdbError

pytestmark = pytest.mark.gpu


def _check(engine, name, stored, ids, queries, k, metric):
    got_ids, got_sc, got_cnt = engine.search(name, queries, k)
    want = O.search(stored, ids, queries, k, metric)
    for qi, (wi, ws) in enumerate(want):
        c = int(got_cnt[qi])
        assert c == len(wi)
        assert np.array_equal(got_ids[qi, :c], wi)
        assert np.array_equal(got_sc[qi, :c].view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("metric", ["ip", "cosine", "l2"])
def test_save_drop_load_round_trip(engine, table_name, tmp_path, metric, dtype):
    m, dt = cases.METRICS[metric], cases.DTYPES[dtype]
    rows = cases.make_rows("synth", 77, 5000, 100)
    ids = cases.make_ids("hash", 77, rows.shape[0])
    engine.create_table(table_name, 100, m, dt)
    engine.insert(table_name, rows, ids)
    stored = O.prepare_store(rows, m, dt)
    path = str(tmp_path / "t.vdbsnap")
    engine.save_table(table_name, path)
    before_rows, before_ids = engine.fetch_rows(table_name, 0, 5000)
    engine.drop_table(table_name)
    assert not engine.has_table(table_name)
    engine.load_table(table_name, path)
    info = engine.table_info(table_name)
    assert (info["dim"], info["metric"], info["dtype"], info["rows"]) == (100, m, dt, 5000)
    after_rows, after_ids = engine.fetch_rows(table_name, 0, 5000)
    assert np.array_equal(before_ids, after_ids)
    assert np.array_equal(before_rows.view(np.uint32), after_rows.view(np.uint32))
    queries = cases.make_rows("synth", 78, 7, 100)
    _check(engine, table_name, stored, ids, queries, 10, m)
    # a loaded table keeps growing like any other
    more = cases.make_rows("synth", 79, 300, 100)
    more_ids = cases.make_ids("index", 79, 300) + 10 ** 12
    engine.insert(table_name, more, more_ids)
    _check(engine, table_name, np.concatenate([stored, O.prepare_store(more, m, dt)]),
           np.concatenate([ids, more_ids]), queries, 10, m)


def test_empty_table_round_trip(engine, table_name, tmp_path):
    engine.create_table(table_name, 12, _lib.METRIC_L2, _lib.F32)
    path = str(tmp_path / "e.vdbsnap")
    engine.save_table(table_name, path)
    engine.drop_table(table_name)
    engine.load_table(table_name, path)
    assert engine.count(table_name) == 0
    ids, _, counts = engine.search(table_name, np.zeros((2, 12), np.float32), 3)
    assert counts.tolist() == [0, 0] and np.all(ids == -1)


def test_bad_snapshots_are_refused(engine, table_name, tmp_path):
    engine.create_table(table_name, 16, _lib.METRIC_IP, _lib.F32)
    engine.insert(table_name, cases.make_rows("synth", 3, 50, 16), np.arange(50))
    path = str(tmp_path / "g.vdbsnap")
    engine.save_table(table_name, path)
    with pytest.raises(VdbError) as e:                     # name taken
        engine.load_table(table_name, path)
    assert e.value.code == _lib.EEXISTS
    with pytest.raises(VdbError) as e:
        engine.load_table("never_there", str(tmp_path / "missing.vdbsnap"))
    assert e.value.code == _lib.ENOTFOUND
    blob = open(path, "rb").read()
    trunc = str(tmp_path / "trunc.vdbsnap")
    open(trunc, "wb").write(blob[:-5])
    junk = str(tmp_path / "junk.vdbsnap")
    open(junk, "wb").write(b"x" * 200)
    for bad in (trunc, junk):
        with pytest.raises(VdbError) as e:
            engine.load_table("from_bad_file", bad)
        assert e.value.code == _lib.EINVAL
        assert not engine.has_table("from_bad_file")
    with pytest.raises(VdbError) as e:
        engine.save_table("never_there", path)
    assert e.value.code == _lib.ENOTFOUND


def test_shim_snapshot_and_preload(engine, tmp_path):
    """The engine-client shim keeps snapshots under primary_path and preloads them by name."""
    from vectordb_amd import milvus as M
    from vectordb_amd.config import EngineConfig
    cfg = EngineConfig(primary_path=str(tmp_path / "data"), preload_table=["*"])
    old = engine.config
    engine.config = cfg
    try:
        client = M.Milvus()
        client._engine = engine                     # same process-wide engine as the other tests
        name = "snap_shim"
        assert client.create_table({"table_name": name, "dimension": 8, "metric_type": M.MetricType.IP}).OK()
        rows = cases.make_rows("synth", 4, 40, 8)
        assert client.insert(name, rows.tolist(), list(range(40)))[0].OK()
        st, want = client.search_vectors(name, 5, query_records=rows[:3].tolist())
        assert st.OK()
        assert client.snapshot(name).OK()
        assert os.path.exists(M.snapshot_path(cfg.primary_path, name))
        engine.drop_table(name)                     # the process "restarts": HBM copy is gone
        assert M._preload(engine, cfg) == [name]
        st, got = client.search_vectors(name, 5, query_records=rows[:3].tolist())
        assert st.OK()
        assert [[(r.id, r.distance) for r in q] for q in got] == [[(r.id, r.distance) for r in q] for q in want]
        assert client.preload_table(name).OK()      # already resident: nothing to do
        assert not client.snapshot("../escape").OK()
        assert client.delete_table(name).OK()
        assert not os.path.exists(M.snapshot_path(cfg.primary_path, name))
        assert not client.preload_table(name).OK()
    finally:
        engine.config = old
        if engine.has_table("snap_shim"):
            engine.drop_table("snap_shim")
