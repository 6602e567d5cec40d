"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference (UAMMD) algorithms on the hot path; see ``oracle/src/*.c`` for the
file:line citations.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker / reported baseline.  The product
(``uammd_amd``) never does.
"""
from .oracle import Oracle, build, get  # noqa: F401
