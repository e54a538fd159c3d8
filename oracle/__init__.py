"""CPU oracle of the reference algorithm — TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
