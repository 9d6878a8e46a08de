"""harness -- bench/test tooling shared by bench.py, __graft_entry__.smoke() and tests/ (not product code)."""
