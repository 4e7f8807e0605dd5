"""CPU oracle package -- test infrastructure only (see hmr_oracle.py header). PARITY UNPINNED."""
