"""CPU tests of the boundary: libvdb.so loads, exports every symbol include/vdb.h declares,
refuses to run without a GPU, and the client shim reports failures the way its caller expects."""
import ctypes
import os
import re

import pytest

from vectordb_amd import _lib
from vectordb_amd.config import EngineConfig, load_config, parse_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vdb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdb_[a-z_0-9]+)\s*\(", text)))


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libvdb.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == declared


def test_constants_match_header():
    text = open(os.path.join(ROOT, "include", "vdb.h")).read()
    defs = {m.group(1): int(m.group(2).strip("()")) for m in
            re.finditer(r"#define\s+(VDB_[A-Z0-9_]+)\s+(\(?-?\d+\)?)", text)}
    assert defs["VDB_OK"] == _lib.OK
    assert (defs["VDB_METRIC_L2"], defs["VDB_METRIC_IP"], defs["VDB_METRIC_COSINE"]) == \
        (_lib.METRIC_L2, _lib.METRIC_IP, _lib.METRIC_COSINE)
    assert (defs["VDB_F32"], defs["VDB_BF16"], defs["VDB_F16"]) == (_lib.F32, _lib.BF16, _lib.F16)
    assert defs["VDB_ENOGPU"] == _lib.ENOGPU and defs["VDB_MAX_TOPK"] == _lib.MAX_TOPK
    from oracle import oracle as O
    assert (O.L2, O.IP, O.COSINE, O.F32, O.BF16, O.F16) == (1, 2, 3, 0, 1, 2)


@pytest.mark.skipif(_gpu(), reason="checks the no-GPU failure path")
def test_no_gpu_means_failure_not_fallback():
    lib = _lib.load()
    assert lib.vdb_init(0) == _lib.ENOGPU
    assert "HIP device" in _lib.last_error()
    # nothing works before a successful init
    out = ctypes.c_int(0)
    assert lib.vdb_has_table(b"x", ctypes.byref(out)) == _lib.ESTATE
    assert lib.vdb_create_table(b"x", 8, _lib.METRIC_IP, _lib.F32) == _lib.ESTATE
    from milvus import Milvus
    st = Milvus().connect(host="127.0.0.1", port="19530")
    assert not st.OK() and st.code == _lib.ENOGPU        # app.py:178-182 exits on this


def test_shim_surface_matches_reference_imports():
    import milvus
    from milvus import IndexType, MetricType, Milvus     # app.py:3
    assert MetricType.IP and IndexType.IVFLAT            # app.py:22,146
    m = Milvus()
    for meth in ("connect", "has_table", "create_table", "create_index", "delete_table", "insert",
                 "search_vectors"):
        assert callable(getattr(m, meth))
    # not connected: every call answers with a failing status instead of raising
    st, ok = m.has_table("nope")
    assert not st.OK() and ok is False
    assert not m.create_table({"table_name": "a", "dimension": 8}).OK()
    st, res = m.search_vectors(table_name="a", query_records=[[0.0] * 8], top_k=10, nprobe=10)
    assert not st.OK() and res == []
    assert m.delete_table("a").code != 0


def test_config_reads_reference_style_file(tmp_path):
    p = tmp_path / "server_config.yaml"
    p.write_text(
        "version: 0.1\n"
        "db_config:\n  primary_path: /tmp/vdb\n  insert_buffer_size: 2\n  preload_table: a, b\n"
        "engine_config:\n  use_blas_threshold: 32\n  gpu_search_threshold: 1000\n"
        "gpu_resource_config:\n  enable: true\n  cache_capacity: 64\n  search_resources:\n    - gpu0\n    - gpu3\n")
    cfg = load_config(str(p))
    assert cfg.use_blas_threshold == 32 and cfg.cache_capacity_gb == 64.0
    assert cfg.search_resources == [0, 3] and cfg.device_for_rank(1) == 3 and cfg.device_for_rank(2) == 0
    assert cfg.primary_path == "/tmp/vdb" and cfg.preload_table == ["a", "b"]
    assert cfg.insert_buffer_size_gb == 2.0
    with pytest.raises(ValueError):
        parse_config({"gpu_resource_config": {"search_resources": ["cpu0"]}})
    assert isinstance(parse_config(None), EngineConfig)
    # the engine's own additions default to: merged callers on, bf16 mirror on, mirror scan off
    d = parse_config(None)
    assert d.coalesce_searches and d.bf16_mirror and not d.scan_mirror
    e = parse_config({"engine_config": {"scan_mirror": True, "bf16_mirror": False, "coalesce_searches": False}})
    assert e.scan_mirror and not e.bf16_mirror and not e.coalesce_searches


def test_snapshot_file_names_stay_inside_primary_path():
    from vectordb_amd import milvus as M
    from vectordb_amd._lib import VdbError
    assert M.snapshot_path("/data", "demo_1") == "/data/demo_1" + M.SNAPSHOT_SUFFIX
    for bad in ("../x", "a/b", "", ".hidden", "a..b", "x" * 300):
        with pytest.raises(VdbError):
            M.snapshot_path("/data", bad)
    # not connected: the snapshot calls answer with a status like every other call
    m = M.Milvus()
    assert not m.snapshot("demo").OK() and not m.preload_table("demo").OK()


def test_bench_cpu_baseline_leg_runs_on_a_small_sample():
    """bench.py's cpu_baseline leg (the oracle's BLAS path, the only part of bench.py that may use
    oracle/) returns the fields the bench line needs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline(rows_total=20_000, dim=64, k=10, metric_name="cosine", budget_s=0.2)
    assert out["unit"] == "QPS" and out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    assert "sample" in out and "20000" in out["sample"]


@pytest.mark.skipif(_gpu(), reason="checks the no-GPU failure path")
def test_cluster_client_reports_a_missing_gpu_as_a_status():
    from vectordb_amd.cluster import ClusterMilvus
    c = ClusterMilvus(EngineConfig(search_resources=[0, 1]))
    st = c.connect(host="127.0.0.1", port="19530")
    assert not st.OK() and st.code == _lib.ENOGPU and not c.connected()
    assert not c.create_table({"table_name": "a", "dimension": 4}).OK()
    st, res = c.search_vectors(table_name="a", query_records=[[0.0] * 4], top_k=3)
    assert not st.OK() and res == []


def test_call_merger_under_thread_sanitizer(tmp_path):
    """The turn-taking that merges concurrent blocking vdb_search calls (vectordb_amd/csrc/host_merge.h)
    is plain host code: a CPU build of exactly that header runs 64 threads over three keys under
    ThreadSanitizer with a stand-in search (tests/native/merge_tsan.cpp)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    exe = str(tmp_path / "merge_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
                           os.path.join(ROOT, "tests", "native", "merge_tsan.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ThreadSanitizer" not in out.stderr and "wrong 0" in out.stdout
