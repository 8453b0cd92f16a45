"""Vector ids as the HTTP layer derives them (/root/reference/app.py:328-334): every component
printed with '%.5f', joined by ',', md5, first 8 digest bytes little-endian as a signed 64-bit
integer.  Vectors equal to 5 decimals share an id; '-0.00000' and '0.00000' differ."""
import hashlib
from typing import Iterable


def hash_vector(vector: Iterable[float]) -> int:
    text = ",".join("%.5f" % component for component in vector)
    return int.from_bytes(hashlib.md5(text.encode("utf-8")).digest()[:8], "little", signed=True)
