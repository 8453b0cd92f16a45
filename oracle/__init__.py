"""CPU oracle for the brute-force similarity query. TEST INFRASTRUCTURE ONLY.

Nothing under ``vectordb_amd/`` may import this package: it exists so that
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
can check (or time a CPU baseline next to) the HIP path.
"""
