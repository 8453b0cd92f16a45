"""CPU tests of the HTTP layer (server/app.py) with a recording stand-in for the engine client:
routes, wire format, hashing, de-duplication and result mapping.  The engine arithmetic is not
under test here (tests/test_gpu_http.py runs the same routes over libvdb.so)."""
import json
import os

import pytest

from server.app import create_app
from server.metastore import MetaStore, MetaStoreError
from vectordb_amd.hashing import hash_vector
from vectordb_amd.milvus import QueryResult, Status

HERE = os.path.dirname(os.path.abspath(__file__))


class RecordingClient:
    """Same seven methods as the engine client; remembers calls, answers canned results."""

    def __init__(self):
        self.tables, self.inserted, self.searches = {}, [], []
        self.next_results = []

    def connect(self, host=None, port=None):
        return Status()

    def has_table(self, name):
        return Status(), name in self.tables

    def create_table(self, param):
        self.tables[param["table_name"]] = param
        return Status()

    def create_index(self, name, param):
        return Status()

    def delete_table(self, name):
        return Status() if self.tables.pop(name, None) is not None else Status(-2, "no such table")

    def insert(self, name, records, ids):
        self.inserted.append((name, records, ids))
        return Status(), ids

    def search_vectors(self, table_name, query_records, top_k, nprobe):
        self.searches.append((table_name, query_records, top_k, nprobe))
        return Status(), self.next_results


@pytest.fixture
def svc():
    client = RecordingClient()
    app = create_app(client=client, store=MetaStore(), compat_lookup=False)     # each hit with its own asset ids
    return app.test_client(), client


def test_hash_vector_matches_reference_golden():
    golden = json.load(open(os.path.join(HERE, "golden", "hash_vector.json")))
    for g in golden:
        assert hash_vector(g["vector"]) == g["hash"]


def test_createdb_insert_lookup_flow(svc):
    http, client = svc
    assert "created" in http.get("/createdb?dbname=test&dimensions=8").get_data(as_text=True)
    assert client.tables["test"]["dimension"] == 8 and client.tables["test"]["index_file_size"] == 32
    assert "already exists" in http.get("/createdb?dbname=test&dimensions=8").get_data(as_text=True)
    # the request documented by the reference (curl-com.txt:1)
    body = {"name": "v1", "vector": [0.00793861557915232, 0.9997302132320715, 0.37542373014507036,
                                     0.6712801110144234, 0.35517993228837497, 0.4425925168063203,
                                     0.48268047072393094, 0.15620825516058134]}
    r = http.put("/insert?dbname=test", json=body)
    assert r.get_data(as_text=True) == "hej"
    assert client.inserted == [("test", [body["vector"]], [6495146723440581616])]
    # same vector under another asset: mapping recorded, engine not called again
    http.put("/insert?dbname=test", json={"name": "v2", "vector": body["vector"]})
    http.put("/insert?dbname=test", json={"name": "v2", "vector": body["vector"]})
    assert len(client.inserted) == 1
    assert json.loads(http.put("/lookupexact?dbname=test", json=body).get_data()) == ["v1", "v2"]

    other = {"name": "w", "vector": [0.5] * 8}
    http.put("/insert?dbname=test", json=other)
    h_other = hash_vector(other["vector"])
    client.next_results = [[QueryResult(6495146723440581616, 1.5), QueryResult(h_other, 1.25),
                            QueryResult(h_other, 1.25)]]          # a duplicate id is reported once
    out = json.loads(http.put("/lookup?dbname=test", json=body).get_data())
    assert out == [{"distance": 0, "asset_ids": ["v1", "v2"]},
                   {"distance": 1.5, "asset_ids": ["v1", "v2"]},
                   {"distance": 1.25, "asset_ids": ["w"]}]
    assert client.searches[-1] == ("test", [body["vector"]], 10, 10)      # k and nprobe of app.py:384-389,426
    out = json.loads(http.put("/lookup?dbname=test&k=3", json={"vectors": [body["vector"], other["vector"]]}).get_data())
    assert client.searches[-1][2] == 3 and out == []           # canned result count != batch size


def test_lookup_answers_as_the_reference_does_by_default():
    """/root/reference/app.py:431 appends the exact-match asset list to every hit; that output is the
    default, each hit's own asset ids the option (VDB_COMPAT_LOOKUP_BUG=0 / compat_lookup=False)."""
    client = RecordingClient()
    http = create_app(client=client, store=MetaStore()).test_client()
    http.get("/createdb?dbname=test&dimensions=4")
    a, b = {"name": "a", "vector": [0.1, 0.2, 0.3, 0.4]}, {"name": "b", "vector": [0.5] * 4}
    http.put("/insert?dbname=test", json=a)
    http.put("/insert?dbname=test", json=b)
    client.next_results = [[QueryResult(hash_vector(a["vector"]), 2.0), QueryResult(hash_vector(b["vector"]), 1.0)]]
    out = json.loads(http.put("/lookup?dbname=test", json=a).get_data())
    assert out == [{"distance": 0, "asset_ids": ["a"]}, {"distance": 2.0, "asset_ids": ["a"]},
                   {"distance": 1.0, "asset_ids": ["a"]}]


class RefusingClient(RecordingClient):
    """An engine that refuses what it is asked to store (HBM budget exhausted, a value that is not
    finite, a lost worker)."""
    refuse_create = False

    def create_table(self, param):
        if self.refuse_create:
            return Status(-1, "dimension must be in [1, 16384]")
        return super().create_table(param)

    def insert(self, name, records, ids):
        return Status(-6, "table storage would exceed gpu_resource_config.cache_capacity"), []


def test_engine_failures_are_reported_and_leave_no_bookkeeping():
    client = RefusingClient()
    store = MetaStore()
    http = create_app(client=client, store=store).test_client()
    client.refuse_create = True
    r = http.get("/createdb?dbname=big&dimensions=99999")
    assert r.status_code == 400 and "dimension" in r.get_data(as_text=True)
    assert not store.has_db("big")
    client.refuse_create = False
    assert "created" in http.get("/createdb?dbname=t&dimensions=4").get_data(as_text=True)
    body = {"name": "a", "vector": [0.1, 0.2, 0.3, 0.4]}
    r = http.put("/insert?dbname=t", json=body)
    assert r.status_code == 500 and "cache_capacity" in r.get_data(as_text=True)
    # the hash is not left behind as known: a retry reaches the engine again
    assert store.assets("t", hash_vector(body["vector"])) == []
    assert store.add_mapping("t", hash_vector(body["vector"]), "a") is True


def test_deletedb_and_integrity(svc):
    http, client = svc
    http.get("/createdb?dbname=a&dimensions=4&metric=cosine&dtype=bf16")
    assert client.tables["a"]["metric_type"] == 3 and client.tables["a"]["dtype"] == "bf16"
    assert http.get("/checkintegrity").get_data(as_text=True) == "All ok"
    client.tables.pop("a")
    assert "bad" in http.get("/checkintegrity").get_data(as_text=True)
    assert "Problem" in http.get("/deletedb?dbname=a").get_data(as_text=True)
    http.get("/createdb?dbname=b&dimensions=4")
    assert http.get("/deletedb?dbname=b").get_data(as_text=True) == "Table deleted"
    assert http.get("/createdb?dbname=x;drop&dimensions=4").status_code == 400
    assert http.get("/createdb?dbname=ok&dimensions=4&metric=hamming").status_code == 400


def test_metastore_rules():
    s = MetaStore()
    s.create_db("t", 8, "IVFLAT", "ip")
    assert s.add_mapping("t", -5, "a") is True
    assert s.add_mapping("t", -5, "b") is False
    assert s.add_mapping("t", -5, "b") is False
    assert s.assets("t", -5) == ["a", "b"] and s.assets("t", 7) == []
    assert s.describe("t") == ("t", 8, "IVFLAT", "ip")
    with pytest.raises(MetaStoreError):
        s.create_db("vectordb_meta", 8, "IVFLAT", "ip")
    with pytest.raises(MetaStoreError):
        s.assets("no such; table", 1)
    assert s.drop_db("t") is True and s.drop_db("t") is False


def test_create_vectors_tool_is_seeded():
    import importlib.util
    spec = importlib.util.spec_from_file_location("create_vectors", os.path.join(HERE, "..", "tools", "create_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = list(mod.vectors(8, 3, seed=0))
    b = list(mod.vectors(8, 3, seed=0))
    assert a == b and a[0]["name"] == "v0" and len(a[2]["vector"]) == 8
    assert all(0.0 <= x < 1.0 for x in a[1]["vector"])
    assert list(mod.vectors(8, 1, seed=1)) != a[:1]


class CountingClient(RecordingClient):
    """A stand-in that also answers the two calls /checkintegrity uses to compare both sides."""

    def get_table_row_count(self, name):
        return Status(), len([1 for n, _, _ in self.inserted if n == name])

    def table_ids(self, name):
        import numpy as np
        return Status(), np.array([ids[0] for n, _, ids in self.inserted if n == name], dtype=np.int64)


def test_integrity_compares_what_both_sides_hold():
    client, store = CountingClient(), MetaStore()
    http = create_app(client=client, store=store).test_client()
    http.get("/createdb?dbname=t&dimensions=3")
    vecs = [[0.1 * i, 0.2, 0.3] for i in range(5)]
    for i, v in enumerate(vecs):
        http.put("/insert?dbname=t", json={"name": "a%d" % i, "vector": v})
    http.put("/insert?dbname=t", json={"name": "again", "vector": vecs[0]})      # known hash: not sent again
    assert len(client.inserted) == 5
    assert http.get("/checkintegrity").get_data(as_text=True) == "All ok"
    lost = client.inserted.pop()                                                   # the engine side loses one vector
    text = http.get("/checkintegrity").get_data(as_text=True)
    assert "bad" in text and "5 vector hashes known, 4 vectors stored" in text
    assert "known but not stored: 1 [%d]" % lost[2][0] in text


def test_same_vector_from_two_requests_while_the_engine_refuses_the_first():
    """The second request must not find the hash 'known' while the first one's engine insert is still
    open: when that insert is refused, neither request may leave the hash behind as known."""
    import threading

    class SlowRefusingClient(RecordingClient):
        def __init__(self):
            super().__init__()
            self.calls, self.gate = 0, threading.Event()

        def insert(self, name, records, ids):
            self.calls += 1
            if self.calls == 1:
                self.gate.wait(5)
                return Status(-6, "refused"), []
            self.inserted.append((name, records, ids))
            return Status(), ids

    client, store = SlowRefusingClient(), MetaStore()
    app = create_app(client=client, store=store)
    app.test_client().get("/createdb?dbname=t&dimensions=3")
    vec = [0.5, 0.25, 0.125]
    out = {}

    def put(tag, asset):
        out[tag] = app.test_client().put("/insert?dbname=t", json={"name": asset, "vector": vec})

    a = threading.Thread(target=put, args=("a", "a1"))
    a.start()
    while client.calls == 0:
        pass
    b = threading.Thread(target=put, args=("b", "a2"))
    b.start()
    b.join(0.3)
    assert b.is_alive()                    # waits for the first request's verdict instead of answering 'hej'
    client.gate.set()
    a.join(5)
    b.join(5)
    assert out["a"].status_code == 500 and out["b"].get_data(as_text=True) == "hej"
    # the second request stored the vector itself; its mapping is the only one
    assert len(client.inserted) == 1 and store.assets("t", hash_vector(vec)) == ["a2"]


def test_engine_insert_that_raises_leaves_no_mapping():
    class RaisingClient(RecordingClient):
        def insert(self, name, records, ids):
            raise RuntimeError("boom")

    store = MetaStore()
    app = create_app(client=RaisingClient(), store=store)
    app.config["PROPAGATE_EXCEPTIONS"] = False
    http = app.test_client()
    http.get("/createdb?dbname=t&dimensions=3")
    vec = [1.0, 2.0, 3.0]
    assert http.put("/insert?dbname=t", json={"name": "x", "vector": vec}).status_code == 500
    assert store.assets("t", hash_vector(vec)) == []
