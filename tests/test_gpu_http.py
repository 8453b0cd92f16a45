"""GPU tests of the reference-facing surfaces: the pymilvus-shaped client used the way
/root/reference/app.py uses it, and the HTTP routes of server/app.py over libvdb.so (the first
BASELINE configuration: U[0,1) vectors inserted one per PUT, single-query cosine top-10)."""
import json

import numpy as np
import pytest

import cases
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_client_shim_as_the_reference_calls_it(engine):
    from milvus import IndexType, MetricType, Milvus
    m = Milvus()
    assert m.connect(host="127.0.0.1", port="19530").OK()                     # app.py:176-178
    name = "shimtable"
    m.delete_table(name)
    status, ok = m.has_table(name)                                             # app.py:120
    assert status.OK() and ok is False
    assert m.create_table({"table_name": name, "dimension": 8, "index_file_size": 32,
                           "metric_type": MetricType.IP}).OK()                 # app.py:137-143
    assert m.create_index(name, {"index_type": IndexType.IVFLAT, "nlist": 2048}).OK()   # app.py:145-149
    assert m.has_table(name)[1] is True
    rows = cases.make_rows("uniform", 9, 50, 8).astype(np.float64)             # JSON floats are float64
    ids = [O.hash_vector(r.tolist()) for r in rows]
    for r, i in zip(rows, ids):
        m.insert(name, records=[r.tolist()], ids=[i])                          # app.py:325, one row per call
    assert m.get_table_row_count(name) == (m.get_table_row_count(name)[0], 50)
    q = rows[7].tolist()
    status, results = m.search_vectors(table_name=name, query_records=[q], top_k=10, nprobe=10)   # app.py:384-391
    assert status.OK() and len(results) == 1
    res = results[0]
    stored = O.prepare_store(rows.astype(np.float32), O.IP)
    (wi, ws), = O.search(stored, np.array(ids, dtype=np.int64), np.array([q], dtype=np.float32), 10, O.IP)
    assert [r.id for r in res] == [int(x) for x in wi]
    assert [np.float32(r.distance) for r in res] == list(ws)
    assert "id=" in repr(res[0])                                               # app.py:400 prints each hit
    # errors come back as statuses, never exceptions
    assert m.create_table({"table_name": name, "dimension": 8}).code != 0
    status, results = m.search_vectors(table_name="absent", query_records=[q], top_k=10, nprobe=10)
    assert not status.OK() and results == []
    st = m.delete_table(name)                                                  # app.py:242-245
    assert st.code == 0
    assert m.delete_table(name).code != 0 and m.delete_table(name).message


def test_http_routes_over_the_engine(engine):
    from server.app import create_app
    from server.metastore import MetaStore
    http = create_app(store=MetaStore(), compat_lookup=False).test_client()      # each hit with its own asset ids
    ref_http = None
    name, d, n = "httpdb", 384, 400
    http.get("/deletedb?dbname=" + name)
    assert "created" in http.get("/createdb?dbname=%s&dimensions=%d&metric=cosine" % (name, d)).get_data(as_text=True)
    rng = np.random.default_rng(0)
    vecs = [rng.random(d).tolist() for _ in range(n)]                # createVector.py:6-8, seeded
    for i, v in enumerate(vecs):
        assert http.put("/insert?dbname=" + name, json={"name": "v%d" % i, "vector": v}).get_data(as_text=True) == "hej"
    http.put("/insert?dbname=" + name, json={"name": "alias", "vector": vecs[3]})     # second asset, same vector
    assert engine.count(name) == n
    ids = np.array([O.hash_vector(v) for v in vecs], dtype=np.int64)
    stored = O.prepare_store(np.asarray(vecs, dtype=np.float32), O.COSINE)
    queries = [vecs[3], np.random.default_rng(1).random(d).tolist()]
    for q in queries:
        out = json.loads(http.put("/lookup?dbname=" + name, json={"name": "q", "vector": q}).get_data())
        (wi, ws), = O.search(stored, ids, np.array([q], dtype=np.float32), 10, O.COSINE)
        assert out[0]["distance"] == 0
        assert out[0]["asset_ids"] == (["v3", "alias"] if q is vecs[3] else [])
        assert len(out) == 11
        for hit, want_id, want_sc in zip(out[1:], wi, ws):
            idx = int(np.nonzero(ids == want_id)[0][0])
            assert hit["asset_ids"] == (["v3", "alias"] if idx == 3 else ["v%d" % idx])
            assert np.float32(hit["distance"]) == want_sc
    # batched body and ?k=: hit by hit against the oracle
    batch = json.loads(http.put("/lookup?dbname=%s&k=5" % name, json={"vectors": queries}).get_data())
    assert len(batch) == 2
    want5 = O.search(stored, ids, np.array(queries, dtype=np.float32), 5, O.COSINE)
    for out, (wi, ws), q in zip(batch, want5, queries):
        assert out[0] == {"distance": 0, "asset_ids": ["v3", "alias"] if q is vecs[3] else []}
        assert len(out) == 6
        for hit, want_id, want_sc in zip(out[1:], wi, ws):
            idx = int(np.nonzero(ids == want_id)[0][0])
            assert hit["asset_ids"] == (["v3", "alias"] if idx == 3 else ["v%d" % idx])
            assert np.float32(hit["distance"]) == want_sc
    assert json.loads(http.put("/lookupexact?dbname=" + name, json={"vector": vecs[3]}).get_data()) == ["v3", "alias"]
    assert http.get("/checkintegrity").get_data(as_text=True) == "All ok"
    assert http.get("/deletedb?dbname=" + name).get_data(as_text=True) == "Table deleted"
    assert not engine.has_table(name)


def test_lookup_default_output_is_the_reference_s(engine):
    """/root/reference/app.py:431 attaches the exact-match asset list to every hit: the default output
    does the same (same distances, same number of entries), compat_lookup=False gives each hit its own."""
    from server.app import create_app
    from server.metastore import MetaStore
    store = MetaStore()
    ref = create_app(store=store).test_client()
    own = create_app(store=store, compat_lookup=False).test_client()
    name, d = "httpref", 16
    ref.get("/deletedb?dbname=" + name)
    assert "created" in ref.get("/createdb?dbname=%s&dimensions=%d" % (name, d)).get_data(as_text=True)
    rng = np.random.default_rng(5)
    vecs = [rng.random(d).tolist() for _ in range(40)]
    for i, v in enumerate(vecs):
        assert ref.put("/insert?dbname=" + name, json={"name": "a%d" % i, "vector": v}).get_data(as_text=True) == "hej"
    a = json.loads(ref.put("/lookup?dbname=" + name, json={"vector": vecs[7]}).get_data())
    b = json.loads(own.put("/lookup?dbname=" + name, json={"vector": vecs[7]}).get_data())
    assert [h["distance"] for h in a] == [h["distance"] for h in b] and len(a) == 11
    assert all(h["asset_ids"] == ["a7"] for h in a)                 # the exact-match list on every line
    assert b[1]["asset_ids"] == ["a7"] and any(h["asset_ids"] != ["a7"] for h in b[2:])
    assert ref.get("/deletedb?dbname=" + name).get_data(as_text=True) == "Table deleted"


@pytest.mark.parametrize("dtype,metric", [("bf16", "l2"), ("f16", "l2"), ("bf16", "ip"), ("f16", "cosine")])
def test_http_reduced_precision_tables_and_metrics(engine, dtype, metric):
    """/createdb?...&dtype=&metric= followed by /insert and /lookup (single and batched, ?k=): ids and
    distances are the oracle's for rows rounded to the table's storage type."""
    from server.app import create_app
    from server.metastore import MetaStore
    http = create_app(store=MetaStore(), compat_lookup=False).test_client()
    name, d, n, k = "http_%s_%s" % (dtype, metric), 48, 300, 7
    http.get("/deletedb?dbname=" + name)
    r = http.get("/createdb?dbname=%s&dimensions=%d&metric=%s&dtype=%s" % (name, d, metric, dtype))
    assert "created" in r.get_data(as_text=True)
    rng = np.random.default_rng(17)
    vecs = [(rng.random(d) - 0.5).tolist() for _ in range(n)]
    for i, v in enumerate(vecs):
        assert http.put("/insert?dbname=" + name, json={"name": "v%d" % i, "vector": v}).get_data(as_text=True) == "hej"
    info = engine.table_info(name)
    assert info["rows"] == n and info["dtype"] == cases.DTYPES[dtype] and info["metric"] == cases.METRICS[metric]
    ids = np.array([O.hash_vector(v) for v in vecs], dtype=np.int64)
    stored = O.prepare_store(np.asarray(vecs, dtype=np.float32), cases.METRICS[metric], cases.DTYPES[dtype])
    queries = [vecs[11], (np.random.default_rng(18).random(d) - 0.5).tolist(), vecs[200]]
    want = O.search(stored, ids, np.array(queries, dtype=np.float32), k, cases.METRICS[metric])
    batch = json.loads(http.put("/lookup?dbname=%s&k=%d" % (name, k), json={"vectors": queries}).get_data())
    assert len(batch) == len(queries)
    for qi, (out, (wi, ws)) in enumerate(zip(batch, want)):
        single = json.loads(http.put("/lookup?dbname=%s&k=%d" % (name, k), json={"vector": queries[qi]}).get_data())
        assert single == out
        assert len(out) == k + 1
        for hit, want_id, want_sc in zip(out[1:], wi, ws):
            idx = int(np.nonzero(ids == want_id)[0][0])
            assert hit["asset_ids"] == ["v%d" % idx]
            assert np.float32(hit["distance"]) == want_sc
    assert http.get("/deletedb?dbname=" + name).get_data(as_text=True) == "Table deleted"


def test_http_insert_of_a_vector_the_engine_refuses(engine):
    """A vector with a value that is not finite: the engine refuses it, /insert answers 500 and the hash
    is not recorded, so nothing is half stored."""
    from server.app import create_app
    from server.metastore import MetaStore
    store = MetaStore()
    http = create_app(store=store).test_client()
    name = "httpnan"
    http.get("/deletedb?dbname=" + name)
    http.get("/createdb?dbname=%s&dimensions=4&dtype=f16" % name)
    assert http.put("/insert?dbname=" + name, json={"name": "ok", "vector": [1.0, 2.0, 3.0, 4.0]}).get_data(as_text=True) == "hej"
    r = http.put("/insert?dbname=" + name, json={"name": "big", "vector": [1.0, 1e6, 3.0, 4.0]})      # overflows fp16
    assert r.status_code == 500 and "finite" in r.get_data(as_text=True)
    assert engine.count(name) == 1
    assert store.assets(name, O.hash_vector([1.0, 1e6, 3.0, 4.0])) == []
    http.get("/deletedb?dbname=" + name)


def test_concurrent_lookups_are_merged_and_stay_exact(engine):
    """Sixteen threads call search_vectors at once, the way one-thread-per-request HTTP serving
    does.  Calls that arrive while a search runs are merged into batched calls inside vdb_search;
    every caller still gets the oracle's answer for its own query, with merging on and off."""
    import threading
    from milvus import MetricType, Milvus
    m = Milvus()
    assert m.connect().OK()
    name = "coalesced"
    m.delete_table(name)
    from oracle import oracle_c as OC
    # a scan of this table takes about 0.2 ms, several times the spacing of the callers' arrivals
    n, d, k, nthreads, rounds = 3_000_000, 96, 10, 16, 4
    assert m.create_table({"table_name": name, "dimension": d, "metric_type": MetricType.COSINE}).OK()
    engine.insert_synthetic(name, n, seed=71, first_row=0)
    stored = O.prepare_store(OC.synth_rows(71, 0, n, d), O.COSINE)
    ids = np.arange(n, dtype=np.int64)
    q = OC.synth_rows(72, 0, nthreads * rounds, d)
    q[5] = stored[123_456]
    want = cases.oracle_search(stored, ids, q, k, O.COSINE)
    qlists = [row.tolist() for row in q]
    got = [None] * len(q)
    barrier = threading.Barrier(nthreads)

    def worker(t):
        for r in range(rounds):
            i = r * nthreads + t
            barrier.wait()
            got[i] = m.search_vectors(table_name=name, query_records=[qlists[i]], top_k=k, nprobe=10)

    for merged in (1, 0):
        engine.set_option("coalesce_host", merged)
        try:
            before = engine.stats()
            threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=120)
            after = engine.stats()
        finally:
            engine.set_option("coalesce_host", 1)
        for i, (wi, ws) in enumerate(want):
            status, results = got[i]
            assert status.OK() and len(results) == 1
            assert [r.id for r in results[0]] == [int(x) for x in wi]
            assert [np.float32(r.distance) for r in results[0]] == list(ws)
            got[i] = None
        assert after[0] - before[0] == len(q)
        # merged: some calls carried several callers and went through the batched path
        assert (after[2] - before[2] > 0) == bool(merged)
    # a failing merged call reports to every caller: the table is gone
    assert m.delete_table(name).OK()
    status, results = m.search_vectors(table_name=name, query_records=[qlists[0]], top_k=k, nprobe=10)
    assert not status.OK() and results == []


def test_integrity_after_restart_reports_the_unsaved_vectors(engine, tmp_path):
    """engine_config.snapshot_every_rows writes incremental snapshots while vectors arrive; after a restart
    (HBM copy gone, relational side kept) /checkintegrity names exactly the vectors that arrived after the
    last snapshot: /insert would never send their hashes to the engine again (/root/reference/app.py:293-295)."""
    import struct
    from server.app import create_app
    from server.metastore import MetaStore
    from vectordb_amd import milvus as M
    from vectordb_amd.config import EngineConfig, parse_config
    assert parse_config({"engine_config": {"snapshot_every_rows": 25}}).snapshot_every_rows == 25
    cfg = EngineConfig(primary_path=str(tmp_path / "data"), snapshot_every_rows=25)
    old = engine.config
    engine.config = cfg
    name, d = "durable", 16
    try:
        client = M.Milvus()
        client._engine = engine
        store = MetaStore(str(tmp_path / "meta.sqlite"))
        http = create_app(client=client, store=store).test_client()
        http.get("/deletedb?dbname=" + name)
        assert "created" in http.get("/createdb?dbname=%s&dimensions=%d" % (name, d)).get_data(as_text=True)
        rng = np.random.default_rng(11)
        vecs = [rng.random(d).tolist() for _ in range(60)]
        for i, v in enumerate(vecs):
            assert http.put("/insert?dbname=" + name, json={"name": "a%d" % i, "vector": v}).get_data(as_text=True) == "hej"
        path = M.snapshot_path(cfg.primary_path, name)
        raw = open(path, "rb").read(72)
        rows, nsegs = struct.unpack_from("<q", raw, 24)[0], struct.unpack_from("<q", raw, 48)[0]
        assert (rows, nsegs) == (50, 2) and client.last_snapshot_rows_written == 25     # the second one was incremental
        assert http.get("/checkintegrity").get_data(as_text=True) == "All ok"
        engine.drop_table(name)                                  # restart: the table comes back from its snapshot
        assert client.preload_table(name).OK() and engine.count(name) == 50
        text = http.get("/checkintegrity").get_data(as_text=True)
        assert "bad" in text and "60 vector hashes known, 50 vectors stored" in text
        missing = sorted(O.hash_vector(v) for v in vecs[50:])
        assert "known but not stored: 10 " + json.dumps(missing) in text
        # stored vectors answer as before
        out = json.loads(http.put("/lookup?dbname=" + name, json={"vector": vecs[7]}).get_data())
        assert out[0]["asset_ids"] == ["a7"] and len(out) == 11
    finally:
        engine.config = old
        if engine.has_table(name):
            engine.drop_table(name)
