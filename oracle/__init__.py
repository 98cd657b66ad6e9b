"""CPU oracle for the AVT training hot path.

TEST INFRASTRUCTURE ONLY. Nothing under ``oracle/`` is part of the product: it may be imported by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py`` -- as the
checker / the timed CPU baseline -- and by nothing else.  The shipped path (``avt_amd``) never falls back
to it and raises if its HIP library is missing.
"""
