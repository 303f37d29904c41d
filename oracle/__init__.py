"""TEST INFRASTRUCTURE ONLY: CPU restatements used as checkers by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under collab_splats_amd/ imports this package."""
