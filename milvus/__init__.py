"""`from milvus import Milvus, IndexType, MetricType` (/root/reference/app.py:3) resolves here
when the repository root is on sys.path: the names are the in-process MI355X engine's."""
from vectordb_amd.milvus import IndexType, MetricType, Milvus, QueryResult, Status  # noqa: F401

__all__ = ["Milvus", "IndexType", "MetricType", "Status", "QueryResult"]
