"""oracle/ — TEST INFRASTRUCTURE ONLY. CPU restatements of the reference's hot path used as the checker.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
