"""ORACLE — test infrastructure only.

CPU restatements of the VLM-FO1 hot path used as the parity
checker.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything from here; the product path
(``vlm_fo1_amd/``, ``vlm_fo1/``) must never do so.
"""
