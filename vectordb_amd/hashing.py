"""Vector ids as the HTTP layer derives them (/root/reference/app.py:328-334): every component
printed with '%.5f', joined by ',', md5, first 8 digest bytes little-endian as a signed 64-bit
integer.  Vectors equal to 5 decimals share an id; '-0.00000' and '0.00000' differ."""
import hashlib
from typing import Iterable


_FORMATS = {}     # length -> "%.5f,%.5f,...": one format operation per vector instead of one per component


def hash_vector(vector: Iterable[float]) -> int:
    values = tuple(vector)
    fmt = _FORMATS.get(len(values))
    if fmt is None:
        if len(_FORMATS) > 64:
            _FORMATS.clear()
        fmt = _FORMATS[len(values)] = ",".join(["%.5f"] * len(values))
    text = fmt % values
    return int.from_bytes(hashlib.md5(text.encode("utf-8")).digest()[:8], "little", signed=True)
