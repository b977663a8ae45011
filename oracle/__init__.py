"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/bmx_oracle.c header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
