"""MI355X-native brute-force similarity engine behind the vector-engine client interface of
mhaulrich/vectordb (see include/vdb.h and vectordb_amd/milvus.py)."""
from ._lib import (BF16, F16, F32, METRIC_COSINE, METRIC_IP, METRIC_L2, VdbError)  # noqa: F401

__all__ = ["BF16", "F16", "F32", "METRIC_COSINE", "METRIC_IP", "METRIC_L2", "VdbError"]
