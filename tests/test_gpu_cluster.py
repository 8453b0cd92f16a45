"""The multi-GPU engine client of the HTTP service (one worker process per device of
gpu_resource_config.search_resources), here with two workers that share cuda:0: its answers equal
the oracle's on the unsharded data, whatever way the rows were spread.  The workers exchange their
hits in one all-gather and merge them on the device (over gloo here: RCCL wants a GPU per rank), or,
with cluster_backend "host", send them to the parent, which merges."""
import numpy as np
import pytest

import cases
from oracle import oracle as O
from vectordb_amd.cluster import ClusterMilvus
from vectordb_amd.config import EngineConfig
from vectordb_amd.milvus import MetricType

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backend", ["", "host"])
def test_two_worker_cluster_matches_the_oracle(tmp_path, backend):
    c = ClusterMilvus(EngineConfig(search_resources=[0, 0], primary_path=str(tmp_path), cluster_backend=backend))
    assert c._backend == (backend or "gloo")
    assert c.connect(host="127.0.0.1", port="19530").OK()
    try:
        for metric, om in ((MetricType.COSINE, O.COSINE), (MetricType.L2, O.L2)):
            name = "cl_%d" % int(metric)
            n, d, k = 3001, 48, 10
            rows = cases.make_rows("ties", 11, n, d)
            ids = cases.make_ids("hash", 11, n)
            assert c.create_table({"table_name": name, "dimension": d, "metric_type": metric}).OK()
            assert c.create_index(name, {"index_type": 2, "nlist": 2048}).OK()
            assert c.has_table(name) == (c.has_table(name)[0], True)
            # a bulk insert, then single rows as /insert sends them
            st, got = c.insert(name, rows[:2500].tolist(), ids[:2500].tolist())
            assert st.OK() and got == ids[:2500].tolist()
            for i in range(2500, n):
                assert c.insert(name, [rows[i].tolist()], [int(ids[i])])[0].OK()
            assert c.get_table_row_count(name)[1] == n
            assert abs(c._rows[name][0] - c._rows[name][1]) <= 1          # shards stay level
            q = cases.make_queries("ties", 11, 6, d, rows)
            st, res = c.search_vectors(table_name=name, query_records=q.tolist(), top_k=k, nprobe=10)
            assert st.OK() and len(res) == 6
            want = O.search(O.prepare_store(rows, om), ids, q, k, om)
            for r, (wi, ws) in zip(res, want):
                assert [x.id for x in r] == [int(i) for i in wi]
                assert [np.float32(x.distance) for x in r] == list(ws)
            # a search every worker refuses (wrong query length) fails as a whole and leaves the workers in step:
            # the next one is answered
            st, res = c.search_vectors(table_name=name, query_records=[[0.5] * (d + 1)], top_k=k)
            assert not st.OK() and res == []
            st, res = c.search_vectors(table_name=name, query_records=q[:1].tolist(), top_k=k)
            assert st.OK() and [x.id for x in res[0]] == [int(i) for i in want[0][0]]
            # a single query and a batch larger than the first one (new buffers on the workers)
            q2 = cases.make_queries("ties", 12, 9, d, rows)
            st, res = c.search_vectors(table_name=name, query_records=q2.tolist(), top_k=3)
            want2 = O.search(O.prepare_store(rows, om), ids, q2, 3, om)
            assert st.OK() and [[x.id for x in r] for r in res] == [[int(i) for i in wi] for wi, _ in want2]
            assert c.delete_table(name).OK() and c.has_table(name)[1] is False
        st, res = c.search_vectors(table_name="absent", query_records=[[0.0] * 4], top_k=3)
        assert not st.OK() and res == []
    finally:
        c.disconnect()


def test_cluster_restart_preloads_every_shard(tmp_path):
    """snapshot -> all worker processes end -> a new cluster with preload_table brings every shard
    back and answers as before (metric and shard sizes are learned from the workers)."""
    n, d, k = 2000, 32, 10
    rows = cases.make_rows("synth", 12, n, d)
    ids = cases.make_ids("hash", 12, n)
    q = cases.make_queries("synth", 12, 4, d, rows)
    c = ClusterMilvus(EngineConfig(search_resources=[0, 0], primary_path=str(tmp_path)))
    assert c.connect().OK()
    try:
        assert c.create_table({"table_name": "keep", "dimension": d, "metric_type": MetricType.IP}).OK()
        assert c.insert("keep", rows.tolist(), ids.tolist())[0].OK()
        assert c.snapshot("keep").OK()
    finally:
        c.disconnect()
    c2 = ClusterMilvus(EngineConfig(search_resources=[0, 0], primary_path=str(tmp_path), preload_table=["*"]))
    assert c2.connect().OK()
    try:
        assert c2.has_table("keep")[1] is True and c2.get_table_row_count("keep")[1] == n
        st, res = c2.search_vectors(table_name="keep", query_records=q.tolist(), top_k=k)
        assert st.OK()
        want = O.search(O.prepare_store(rows, O.IP), ids, q, k, O.IP)
        for r, (wi, ws) in zip(res, want):
            assert [x.id for x in r] == [int(i) for i in wi]
            assert [np.float32(x.distance) for x in r] == list(ws)
        assert c2.insert("keep", [rows[0].tolist()], [123])[0].OK()          # placement state was recovered
        assert c2.delete_table("keep").OK()
    finally:
        c2.disconnect()
