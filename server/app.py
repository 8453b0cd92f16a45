"""HTTP service with the reference's routes (/root/reference/app.py:194-435) over the in-process
MI355X engine.  Same wire format (curl-com.txt:1: PUT, JSON body {"name", "vector"}, query argument
`dbname`, port 5001) and the same flow per request:

  PUT /insert       hash the vector, record (hash, asset); a new hash also stores the vector
  PUT /lookup       exact matches by hash first (distance 0), then the engine's top-k, each hit
                    mapped back to its asset ids
  PUT /lookupexact  asset ids of the vector's hash
  GET /createdb     ?dbname=&dimensions=  (+ optional metric=ip|l2|cosine, dtype=f32|bf16|f16)
  GET /deletedb, /checkintegrity, /

The engine client is the `milvus` module of this repository (the reference's import line,
app.py:3, unchanged); the relational side is server/metastore.py instead of a Postgres server.

/lookup answers what the reference answers: the exact-match asset list is attached to every hit
(app.py:431 appends `assets_ids`, the exact-match list, where each hit's own `asset_ids` was
probably meant).  VDB_COMPAT_LOOKUP_BUG=0 (or create_app(compat_lookup=False)) attaches each hit's
own asset ids instead.

Differences from the reference, on purpose:
  * an engine call that fails is reported (4xx/5xx with the status message) and leaves no
    bookkeeping behind; the reference ignores the statuses of create_table and insert
    (app.py:143,149,325), so a refused vector would stay recorded as known and never be stored.
  * /lookup accepts optional ?k= (default 10, as hard-coded at app.py:426) and a JSON body with
    "vectors": [[...], ...] for a batch (answers a list of result lists).
  * database names are validated instead of being concatenated into SQL.
"""
from __future__ import annotations

import json
import os
import sys
import threading

from flask import Flask, request

from milvus import IndexType, MetricType, Milvus
from server.metastore import MetaStore, MetaStoreError, check_name
from vectordb_amd.cluster import ClusterMilvus
from vectordb_amd.config import load_config
from vectordb_amd.hashing import hash_vector

_HOST = "127.0.0.1"       # kept for the connect() call; the engine runs in this process
_PORT = "19530"
_INDEX_FILE_SIZE = 32
_METRICS = {"ip": MetricType.IP, "l2": MetricType.L2, "cosine": MetricType.COSINE}


def create_app(client: Milvus = None, store: MetaStore = None, compat_lookup: bool = None) -> Flask:
    app = Flask(__name__)
    state = {"client": client, "store": store or MetaStore(os.environ.get("VDB_META_PATH", ":memory:"))}
    lock = threading.Lock()
    # requests that carry the same vector hash take turns from the bookkeeping to the engine insert
    hash_locks = [threading.Lock() for _ in range(64)]
    compat_bug = (os.environ.get("VDB_COMPAT_LOOKUP_BUG", "1") != "0") if compat_lookup is None else bool(compat_lookup)

    def engine() -> Milvus:
        with lock:
            if state["client"] is None:
                # several devices in gpu_resource_config.search_resources: one worker process per GPU
                cfg = load_config()
                m = ClusterMilvus(cfg) if len(cfg.search_resources) > 1 else Milvus()
                status = m.connect(host=_HOST, port=_PORT)
                if not status.OK():
                    print("Server connect fail.", status.message)
                    sys.exit(1)                      # as app.py:181-182
                state["client"] = m
            return state["client"]

    def store_() -> MetaStore:
        return state["store"]

    def create_vector_db(name, dimensions, metric="ip", dtype="f32", index_type="IVFLAT"):
        if store_().has_db(name):
            return -1, "Error: Vector database with name: " + name + " already exists"
        status, exists = engine().has_table(name)
        if exists:
            return -1, "Error: table with name: " + name + " exists in milvus (but not in postgres)"
        check_name(name)
        # the engine table first: a refused table (dimension out of range, HBM budget) leaves no record
        status = engine().create_table({"table_name": name, "dimension": dimensions, "index_file_size": _INDEX_FILE_SIZE,
                                        "metric_type": _METRICS[metric], "dtype": dtype})
        if not status.OK():
            return -2, "Error: " + status.message
        try:
            store_().create_db(name, dimensions, index_type, metric)
        except Exception:
            engine().delete_table(name)
            raise
        engine().create_index(name, {"index_type": IndexType.IVFLAT, "nlist": 2048})
        return 1, "Vector database with name: " + name + " created"

    def lookup_engine(dbname, vectors, k):
        status, results = engine().search_vectors(table_name=dbname, query_records=vectors, top_k=k, nprobe=10)
        out = []
        for res in results:
            seen, hits = set(), []
            for r in res:                               # one line per distinct id, best first
                if r.id in seen:
                    continue
                seen.add(r.id)
                hits.append({"distance": r.distance, "vectorhash": r.id})
            out.append(hits)
        return out

    @app.route("/")
    def hello():
        return "Total rows: " + str(len(store_().list_dbs()) + 1)     # the reference counts from 1 (app.py:163)

    @app.route("/checkintegrity")
    def check_integrity():
        # Beyond the reference's check (does every database have an engine table, app.py:194-225): do both
        # sides hold the same vectors?  /insert never sends a known hash to the engine again
        # (app.py:293-295), so a hash that is known here and missing there (rows that arrived after the last
        # snapshot of a restarted engine) would stay missing without anybody noticing.
        all_good = True
        details = []
        print("Table name\tExists in postgres\tExists in Milvus")
        for name, _, _, _ in store_().list_dbs():
            status, in_engine = engine().has_table(name)
            print(name + "\t\tTrue\t\t\t\t" + str(in_engine))
            all_good = all_good and in_engine
            if not in_engine:
                details.append("%s: no engine table" % name)
                continue
            client = engine()
            if not hasattr(client, "get_table_row_count"):
                continue
            st, stored = client.get_table_row_count(name)
            known = store_().hashes(name)
            if st.OK() and stored == len(known):
                continue
            all_good = False
            line = "%s: %d vector hashes known, %d vectors stored" % (name, len(known), stored)
            if hasattr(client, "table_ids"):
                st, ids = client.table_ids(name)
                if st.OK():
                    have = set(int(i) for i in ids)
                    missing = sorted(h for h in known if h not in have)
                    line += "; known but not stored: %d %s" % (len(missing), json.dumps(missing[:50]))
            print(line)
            details.append(line)
        if all_good:
            return "All ok"
        return "DB integrety bad. Check server output for details" + "".join("\n" + d for d in details)

    @app.route("/createdb")
    def create_new_db():
        try:
            name = request.args.get("dbname")
            dimensions = int(request.args.get("dimensions"))
            metric = (request.args.get("metric") or "ip").lower()
            dtype = (request.args.get("dtype") or "f32").lower()
            if metric not in _METRICS:
                return "Error: unknown metric " + metric, 400
            code, text = create_vector_db(name, dimensions, metric, dtype)
            return (text, 400) if code == -2 else text
        except (MetaStoreError, TypeError, ValueError) as exc:
            return "Error: " + str(exc), 400

    @app.route("/deletedb")
    def delete_db():
        name = request.args.get("dbname")
        all_ok = True
        status = engine().delete_table(name)
        if status.code != 0:
            all_ok = False
            print(status.message)
        try:
            if not store_().drop_db(name):
                all_ok = False
        except MetaStoreError as exc:
            all_ok = False
            print("Error while deleting table", exc)
        return "Table deleted" if all_ok else "Problem with deleting table. See server log for details"

    @app.route("/snapshot")
    def snapshot_db():
        # no counterpart among the reference's routes: its external engine wrote every insert to
        # db_config.primary_path by itself; the in-process engine writes a table when asked to
        name = request.args.get("dbname")
        status = engine().snapshot(name)
        if status.code != 0:
            return "Error: " + status.message, 400
        return "Snapshot written"

    @app.route("/insert", methods=["PUT"])
    def insert_vector():
        name = request.args.get("dbname")
        body = request.json
        asset_id, vector = body["name"], body["vector"]
        vector_hash = hash_vector(vector)
        # One request at a time per hash, from the bookkeeping to the engine: a second request with the same
        # vector must not find the hash "known" while the first one's engine insert can still be refused.
        with hash_locks[vector_hash % len(hash_locks)]:
            try:
                is_new = store_().add_mapping(name, vector_hash, asset_id)
            except (MetaStoreError, Exception) as exc:      # unknown database etc.
                return "Error: " + str(exc), 400
            if is_new:
                stored, message = False, "engine insert failed"
                try:
                    result = engine().insert(name, records=[vector], ids=[vector_hash])
                    status = result[0] if isinstance(result, tuple) else result
                    stored, message = status.OK(), status.message
                finally:
                    if not stored:
                        # the vector is not stored: forget the hash, or every retry would find it known
                        store_().remove_hash(name, vector_hash)
                if not stored:
                    return "Error: " + message, 500
        return "hej"                                     # app.py:349

    @app.route("/lookupexact", methods=["PUT"])
    def lookup_exact():
        name = request.args.get("dbname")
        vector = request.json["vector"]
        try:
            return json.dumps(store_().assets(name, hash_vector(vector)))
        except Exception as exc:
            return "Error: " + str(exc), 400

    def one_lookup(name, vector, hits):
        exact = store_().assets(name, hash_vector(vector))
        results = [{"distance": 0, "asset_ids": exact}]
        for hit in hits:
            # (the reference looks every hit's assets up as well and then returns the exact match's: app.py:431)
            assets = exact if compat_bug else store_().assets(name, hit["vectorhash"])
            results.append({"distance": hit["distance"], "asset_ids": assets})
        return results

    @app.route("/lookup", methods=["PUT"])
    def lookup():
        name = request.args.get("dbname")
        body = request.json
        try:
            k = int(request.args.get("k") or 10)
            if "vectors" in body:
                vectors = body["vectors"]
                hits = lookup_engine(name, vectors, k)
                if len(hits) != len(vectors):
                    return json.dumps([])
                return json.dumps([one_lookup(name, v, h) for v, h in zip(vectors, hits)])
            vector = body["vector"]
            hits = lookup_engine(name, [vector], k)
            return json.dumps(one_lookup(name, vector, hits[0] if hits else []))
        except Exception as exc:
            return "Error: " + str(exc), 400

    return app


if __name__ == "__main__":
    create_app().run(port=5001)                          # app.py:443
