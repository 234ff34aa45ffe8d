"""CPU oracle for the GeoT hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  ``geot_amd`` (the product)
never imports it.  See ``oracle/geot_oracle.c`` for the parity status.
"""
from .capi import *  # noqa: F401,F403
