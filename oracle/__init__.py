"""CPU oracle for the CPC training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cpc2_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / the timed CPU baseline, never as
the product path.

Parity is PINNED: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors under ``tests/golden/`` that were produced by importing
the reference implementation (``/root/reference/cpc``) in the build container
with ``tools/make_golden.py``.
"""
