"""CPU oracle for the plane-sweep cost-volume path — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
robustmvd_amd/ (the product fails loudly without its HIP library instead of falling back here).
"""
