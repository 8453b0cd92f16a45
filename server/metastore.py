"""Relational bookkeeping of the HTTP layer: which vector databases exist and which asset ids
belong to which vector hash.  The reference keeps this in Postgres (/root/reference/app.py:25-114,
286-312, 352-364: a meta table `vectordb_meta(name, dims, index_type)` and one table per database
`(vector_hash BIGINT, asset_id VARCHAR(20), PRIMARY KEY(vector_hash, asset_id))`).  This module
keeps the same two relations in SQLite (stdlib; ':memory:' by default, a file path for
durability) behind a small API, so the service runs without a database server.  It is not on the
data-parallel path: one or two indexed statements per request."""
from __future__ import annotations

import re
import sqlite3
import threading
from typing import List, Optional, Tuple

META_TABLE = "vectordb_meta"
_NAME_RE = re.compile(r"^[A-Za-z_][A-Za-z0-9_]{0,29}$")      # the reference's VARCHAR(30) names


class MetaStoreError(ValueError):
    pass


def check_name(name: Optional[str]) -> str:
    if not name or not _NAME_RE.match(name) or name == META_TABLE:
        raise MetaStoreError("invalid database name: %r" % (name,))
    return name


class MetaStore:
    def __init__(self, path: str = ":memory:"):
        self._lock = threading.Lock()
        self._db = sqlite3.connect(path, check_same_thread=False)
        with self._lock, self._db:
            self._db.execute("CREATE TABLE IF NOT EXISTS %s (name VARCHAR(30) NOT NULL PRIMARY KEY, "
                             "dims INT NOT NULL, index_type VARCHAR(30), metric VARCHAR(10))" % META_TABLE)

    # -- databases
    def has_db(self, name: str) -> bool:
        with self._lock:
            cur = self._db.execute("SELECT 1 FROM %s WHERE name = ?" % META_TABLE, (name,))
            return cur.fetchone() is not None

    def create_db(self, name: str, dims: int, index_type: str, metric: str) -> None:
        check_name(name)
        with self._lock, self._db:
            self._db.execute("INSERT INTO %s (name, dims, index_type, metric) VALUES (?, ?, ?, ?)" % META_TABLE,
                             (name, int(dims), index_type, metric))
            self._db.execute("CREATE TABLE %s (vector_hash BIGINT NOT NULL, asset_id VARCHAR(20) NOT NULL, "
                             "PRIMARY KEY(vector_hash, asset_id))" % name)

    def drop_db(self, name: str) -> bool:
        check_name(name)
        with self._lock, self._db:
            cur = self._db.execute("DELETE FROM %s WHERE name = ?" % META_TABLE, (name,))
            existed = cur.rowcount > 0
            self._db.execute("DROP TABLE IF EXISTS %s" % name)
            return existed

    def list_dbs(self) -> List[Tuple[str, int, str, str]]:
        with self._lock:
            return list(self._db.execute("SELECT name, dims, index_type, metric FROM %s ORDER BY name" % META_TABLE))

    def describe(self, name: str) -> Optional[Tuple[str, int, str, str]]:
        with self._lock:
            return self._db.execute("SELECT name, dims, index_type, metric FROM %s WHERE name = ?" % META_TABLE,
                                    (name,)).fetchone()

    # -- hash -> assets
    def add_mapping(self, name: str, vector_hash: int, asset_id: str) -> bool:
        """Record (hash, asset); returns True when the hash was not known before, i.e. when the
        vector itself must be stored in the engine (the three cases of app.py:275-312)."""
        check_name(name)
        with self._lock, self._db:
            known = self._db.execute("SELECT 1 FROM %s WHERE vector_hash = ? LIMIT 1" % name,
                                     (vector_hash,)).fetchone() is not None
            self._db.execute("INSERT OR IGNORE INTO %s (vector_hash, asset_id) VALUES (?, ?)" % name,
                             (vector_hash, str(asset_id)))
            return not known

    def remove_mapping(self, name: str, vector_hash: int, asset_id: str) -> None:
        """Undo add_mapping (the engine refused the vector that belongs to it)."""
        check_name(name)
        with self._lock, self._db:
            self._db.execute("DELETE FROM %s WHERE vector_hash = ? AND asset_id = ?" % name, (vector_hash, str(asset_id)))

    def remove_hash(self, name: str, vector_hash: int) -> None:
        """Forget every asset of a hash whose vector the engine refused."""
        check_name(name)
        with self._lock, self._db:
            self._db.execute("DELETE FROM %s WHERE vector_hash = ?" % name, (vector_hash,))

    def hashes(self, name: str) -> List[int]:
        """Distinct vector hashes of a database: the ids its engine table should hold."""
        check_name(name)
        with self._lock:
            return [r[0] for r in self._db.execute("SELECT DISTINCT vector_hash FROM %s" % name)]

    def assets(self, name: str, vector_hash: int) -> List[str]:
        check_name(name)
        with self._lock:
            cur = self._db.execute("SELECT asset_id FROM %s WHERE vector_hash = ? ORDER BY rowid" % name,
                                   (vector_hash,))
            return [r[0] for r in cur.fetchall()]
