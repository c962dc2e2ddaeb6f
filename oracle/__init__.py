"""CPU oracle for the talker AR decode path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (``ht_vllm_omni_amd``) never
imports it and fails loudly when its HIP library is missing.
"""
