"""cProfile of PUT /lookup through the HTTP routes in this process (GPU): where a lookup of 0.43 ms spends its time.
  python tools/profile_http_lookup.py"""
import cProfile, pstats, sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, json
from server.app import create_app
from server.metastore import MetaStore
with contextlib.redirect_stdout(io.StringIO()):
    http = create_app(store=MetaStore()).test_client()
    http.get("/createdb?dbname=p&dimensions=384&metric=cosine")
    rng = np.random.default_rng(0)
    vecs = rng.random((2000, 384))
    for i in range(2000):
        http.put("/insert?dbname=p", json={"name": "v%d" % i, "vector": vecs[i].tolist()})
    qs = rng.random((300, 384))
    payloads = [json.dumps({"vector": q.tolist()}) for q in qs]
    for i in range(20):
        http.put("/lookup?dbname=p", data=payloads[i], content_type="application/json")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20, 300):
        http.put("/lookup?dbname=p", data=payloads[i], content_type="application/json")
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
