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


def _snap_header(path):
    """(rows, max_n2, uuid, nsegs) of a version-2 snapshot header."""
    import struct
    raw = open(path, "rb").read(72)
    assert raw[:8] == b"VDBSNAP2"
    dim, metric, dtype, rows, m, max_n2, uuid, nsegs = struct.unpack_from("<qiiqifQq", raw, 8)
    return rows, max_n2, uuid, nsegs


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_second_save_appends_only_the_new_rows(engine, table_name, tmp_path, dtype):
    """A later save of the same table to the same path writes one more segment with the rows that
    arrived since (the engine the reference talks to persists every insert under primary_path,
    /root/reference/server_config.yaml:12); the file loads as the whole table."""
    m, dt = cases.METRICS["cosine"], cases.DTYPES[dtype]
    d = 96
    rows = cases.make_rows("synth", 31, 6000, d)
    ids = cases.make_ids("hash", 31, 6000)
    engine.create_table(table_name, d, m, dt)
    engine.insert(table_name, rows[:4000], ids[:4000])
    path = str(tmp_path / "inc.vdbsnap")
    assert engine.save_table(table_name, path) == 4000           # whole table
    size1 = os.path.getsize(path)
    assert engine.save_table(table_name, path) == 0              # nothing new: file untouched
    assert os.path.getsize(path) == size1
    engine.insert(table_name, rows[4000:5500], ids[4000:5500])
    assert engine.save_table(table_name, path) == 1500           # only the new rows
    stride = engine.table_info(table_name)["row_stride_bytes"]
    assert os.path.getsize(path) == size1 + 24 + 1500 * (stride + 12)
    hrows, _, uuid, nsegs = _snap_header(path)
    assert (hrows, nsegs) == (5500, 2)
    engine.insert(table_name, rows[5500:], ids[5500:])
    assert engine.save_table(table_name, path) == 500
    before_rows, before_ids = engine.fetch_rows(table_name, 0, 6000)
    engine.drop_table(table_name)
    engine.load_table(table_name, path)
    after_rows, after_ids = engine.fetch_rows(table_name, 0, 6000)
    assert np.array_equal(before_ids, after_ids)
    assert np.array_equal(before_rows.view(np.uint32), after_rows.view(np.uint32))
    queries = cases.make_rows("synth", 32, 5, d)
    _check(engine, table_name, O.prepare_store(rows, m, dt), ids, queries, 10, m)
    # the loaded table is the same table: its next save extends the same file
    more = cases.make_rows("synth", 33, 100, d)
    engine.insert(table_name, more, np.arange(100) + 10 ** 15)
    assert engine.save_table(table_name, path) == 100
    assert _snap_header(path)[2] == uuid and _snap_header(path)[0] == 6100
    # another table under the same name is not: the file is written anew
    engine.drop_table(table_name)
    engine.create_table(table_name, d, m, dt)
    engine.insert(table_name, rows[:10], ids[:10])
    assert engine.save_table(table_name, path) == 10
    hrows, _, uuid2, nsegs = _snap_header(path)
    assert (hrows, nsegs) == (10, 1) and uuid2 != uuid


def test_an_append_that_did_not_reach_its_header_is_ignored(engine, table_name, tmp_path):
    """Bytes behind the last segment the header accounts for (a crash between writing a segment and
    rewriting the header) do not make the snapshot unreadable, and the next save overwrites them."""
    engine.create_table(table_name, 16, _lib.METRIC_IP, _lib.F32)
    rows = cases.make_rows("synth", 41, 300, 16)
    engine.insert(table_name, rows[:200], np.arange(200))
    path = str(tmp_path / "crash.vdbsnap")
    assert engine.save_table(table_name, path) == 200
    good = os.path.getsize(path)
    with open(path, "ab") as fh:
        fh.write(b"VDBSEG01" + b"\x07" * 1000)                   # half a segment
    engine.insert(table_name, rows[200:], np.arange(200, 300))
    assert engine.save_table(table_name, path) == 100
    assert os.path.getsize(path) == good + 24 + 100 * (64 + 12)
    engine.drop_table(table_name)
    engine.load_table(table_name, path)
    assert engine.count(table_name) == 300
    _check(engine, table_name, O.prepare_store(rows, O.IP, O.F32), np.arange(300, dtype=np.int64), rows[:4], 10, O.IP)


def test_version_1_snapshots_still_load(engine, table_name, tmp_path):
    import struct
    rows = cases.make_rows("synth", 51, 64, 8)
    ids = np.arange(64, dtype=np.int64) * 3 - 50
    engine.create_table(table_name, 8, _lib.METRIC_L2, _lib.F32)
    engine.insert(table_name, rows, ids)
    stored, _ = engine.fetch_rows(table_name, 0, 64)
    n2 = (stored.astype(np.float64) ** 2).sum(axis=1).astype(np.float32)
    path = str(tmp_path / "v1.vdbsnap")
    with open(path, "wb") as fh:
        fh.write(b"VDBSNAP1" + struct.pack("<qiiqif", 8, _lib.METRIC_L2, _lib.F32, 64, 2, float(n2.max()) * 1.001))
        fh.write(stored.tobytes() + ids.tobytes() + n2.tobytes())
    engine.drop_table(table_name)
    engine.load_table(table_name, path)
    _check(engine, table_name, O.prepare_store(rows, O.L2, O.F32), ids, rows[:3], 5, O.L2)


def test_refused_insert_leaves_the_norm_bound_alone(engine, table_name, tmp_path):
    """A batch that is refused (one row not finite) must not loosen the table-wide norm bound the
    certificates use, although its finite rows were long."""
    engine.create_table(table_name, 8, _lib.METRIC_IP, _lib.F32)
    engine.insert(table_name, cases.make_rows("synth", 61, 100, 8), np.arange(100))
    p1, p2 = str(tmp_path / "a.vdbsnap"), str(tmp_path / "b.vdbsnap")
    engine.save_table(table_name, p1)
    bound = _snap_header(p1)[1]
    bad = np.full((3, 8), 1.0e15, dtype=np.float32)
    bad[1, 4] = np.nan
    with pytest.raises(VdbError) as e:
        engine.insert(table_name, bad, np.arange(3) + 1000)
    assert e.value.code == _lib.EINVAL and engine.count(table_name) == 100
    engine.save_table(table_name, p2)
    assert _snap_header(p2)[1] == bound
    # a batch that goes through does raise it
    engine.insert(table_name, np.full((1, 8), 100.0, dtype=np.float32), np.array([5000]))
    engine.save_table(table_name, p2)
    assert _snap_header(p2)[1] >= 8 * 100.0 ** 2 > bound


def test_mapping_that_fails_part_way_keeps_the_budget_count_right(engine, tmp_path):
    """Chunks mapped before a mapping failure stay with the table and are counted; after the table is
    dropped the budget (gpu_resource_config.cache_capacity) is whole again."""
    name = "map_fail"
    if engine.has_table(name):
        engine.drop_table(name)
    try:
        engine.create_table(name, 64, _lib.METRIC_IP, _lib.BF16)     # no mirror to give up: the failure reaches the caller
        engine.set_option("debug_fail_map", 2)               # the second chunk asked for is refused
        with pytest.raises(VdbError) as e:
            engine.insert_synthetic(name, 2000, seed=1, first_row=0)
        assert e.value.code == _lib.ENOMEM
        engine.set_option("debug_fail_map", 0)
        engine.insert_synthetic(name, 2000, seed=1, first_row=0)     # the same call now goes through
        assert engine.count(name) == 2000
        engine.drop_table(name)
        engine.set_option("cache_capacity_gb", 1.0)
        engine.create_table(name, 64, _lib.METRIC_IP, _lib.F32)
        engine.insert_synthetic(name, 5000, seed=2, first_row=0)     # well inside 1 GB
        assert engine.count(name) == 5000
    finally:
        engine.set_option("debug_fail_map", 0)
        engine.set_option("cache_capacity_gb", 0.0)
        if engine.has_table(name):
            engine.drop_table(name)
