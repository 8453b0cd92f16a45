"""Reads the engine's configuration file and maps its knobs onto the in-process engine.

The file format is that of /root/reference/server_config.yaml (which the reference's app.py
never opens: it configures the external engine process this library replaces).  Keys with a
meaning for an in-process, exhaustive-scan engine are honoured; the rest are accepted and kept
in `raw` so that an existing file loads unchanged.

  engine_config.use_blas_threshold (server_config.yaml:36)  batch size from which batched
      queries use the MFMA GEMM path instead of the per-query scan.  The reference's file says 20
      (the external engine's CPU BLAS crossover); here one batched pass over up to 128 queries costs
      about 1.2 scans and the library compares both estimates per call, so the built-in default is 2
      and a deployment that keeps the reference's file should lower it
  engine_config.coalesce_searches (an addition, default true)  searches that arrive while another one
      is running are merged into one batched call (inside vdb_search; in front of the workers of the
      multi-GPU client: vectordb_amd/coalesce.py)
  engine_config.scan_mirror (an addition, default false)  single queries over fp32 tables scan the bf16
      mirror of the rows: half the bytes per query and the same exact answers (the certificate of the
      fast pass widens by the rounding; tie-heavy data sends more queries to the exact pass)
  engine_config.bf16_mirror (an addition, default true)  fp32 tables keep that mirror (half of the table
      again in HBM); without it batched searches convert the fp32 rows on the fly, at half the speed
  engine_config.snapshot_every_rows (an addition, default 0 = never)  a table is written to
      db_config.primary_path whenever this many rows have arrived since its last snapshot; the write is
      incremental (only the new rows).  The engine the reference talks to persists every insert by itself,
      and /insert relies on it: a hash the relational side knows is never sent to the engine again
  engine_config.cluster_backend (an addition, default chosen by the device list)  with more than one entry in
      search_resources: how the worker processes exchange their hits of a search -- "nccl" (one RCCL
      all-gather between the GPUs, merge on the device), "gloo" (the same through host memory; for workers that
      share a GPU) or "host" (the serving process merges what the workers send it)
  gpu_resource_config.enable (:41)           must be true: there is no CPU engine
  gpu_resource_config.cache_capacity (:42)   GB of HBM the tables may occupy (0 = no cap)
  gpu_resource_config.search_resources (:43-44)  `gpuN` entries: the shard set; a process binds
      to one of them (entry LOCAL_RANK, or the first)
  db_config.insert_buffer_size (:19)         GB of staging for host inserts
  db_config.primary_path (:12)               directory of table snapshots
  db_config.preload_table (:22)              tables to load from snapshots at start
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import yaml

_GPU_RE = re.compile(r"^gpu(\d+)$")


@dataclass
class EngineConfig:
    use_blas_threshold: int = 2          # this engine's measured crossover; the file's value wins when present
    gpu_search_threshold: int = 1000
    coalesce_searches: bool = True
    scan_mirror: bool = False
    bf16_mirror: bool = True
    snapshot_every_rows: int = 0
    cluster_backend: str = ""            # "" = nccl when every worker has its own GPU, else gloo; or nccl / gloo / host
    gpu_enable: bool = True
    cache_capacity_gb: float = 0.0
    search_resources: List[int] = field(default_factory=lambda: [0])
    insert_buffer_size_gb: float = 1.0
    primary_path: str = "/var/lib/milvus"
    preload_table: List[str] = field(default_factory=list)
    raw: Dict[str, Any] = field(default_factory=dict)

    def device_for_rank(self, local_rank: int = 0) -> int:
        if not self.search_resources:
            return 0
        return self.search_resources[local_rank % len(self.search_resources)]


def _gpu_list(items) -> List[int]:
    out = []
    for it in items or []:
        m = _GPU_RE.match(str(it).strip())
        if not m:
            raise ValueError("search_resources entries must look like gpu0, gpu1, ...: %r" % (it,))
        out.append(int(m.group(1)))
    return out


def parse_config(doc: Optional[Dict[str, Any]]) -> EngineConfig:
    doc = doc or {}
    cfg = EngineConfig(raw=doc)
    eng = doc.get("engine_config") or {}
    gpu = doc.get("gpu_resource_config") or {}
    db = doc.get("db_config") or {}
    if "use_blas_threshold" in eng:
        cfg.use_blas_threshold = int(eng["use_blas_threshold"])
    if "gpu_search_threshold" in eng:
        cfg.gpu_search_threshold = int(eng["gpu_search_threshold"])
    if "coalesce_searches" in eng:
        cfg.coalesce_searches = bool(eng["coalesce_searches"])
    if "scan_mirror" in eng:
        cfg.scan_mirror = bool(eng["scan_mirror"])
    if "bf16_mirror" in eng:
        cfg.bf16_mirror = bool(eng["bf16_mirror"])
    if "snapshot_every_rows" in eng and eng["snapshot_every_rows"] is not None:
        cfg.snapshot_every_rows = max(0, int(eng["snapshot_every_rows"]))
    if eng.get("cluster_backend"):
        cfg.cluster_backend = str(eng["cluster_backend"]).strip().lower()
    if "enable" in gpu:
        cfg.gpu_enable = bool(gpu["enable"])
    if "cache_capacity" in gpu and gpu["cache_capacity"] is not None:
        cfg.cache_capacity_gb = float(gpu["cache_capacity"])
    if "search_resources" in gpu:
        cfg.search_resources = _gpu_list(gpu["search_resources"]) or [0]
    if "insert_buffer_size" in db and db["insert_buffer_size"] is not None:
        cfg.insert_buffer_size_gb = float(db["insert_buffer_size"])
    if db.get("primary_path"):
        cfg.primary_path = str(db["primary_path"])
    pre = db.get("preload_table")
    if pre:
        cfg.preload_table = [s.strip() for s in str(pre).split(",") if s.strip()]
    if cfg.use_blas_threshold < 1:
        raise ValueError("engine_config.use_blas_threshold must be >= 1")
    return cfg


def load_config(path: Optional[str] = None) -> EngineConfig:
    """Load `path`, or $VDB_SERVER_CONFIG, or ./server_config.yaml; defaults when none exists."""
    path = path or os.environ.get("VDB_SERVER_CONFIG")
    if path is None and os.path.exists("server_config.yaml"):
        path = "server_config.yaml"
    if path is None:
        return EngineConfig()
    with open(path, "r") as fh:
        return parse_config(yaml.safe_load(fh))
