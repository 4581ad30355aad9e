"""Test infrastructure: CPU restatement of the reference hot path (see mvae_oracle.py header)."""
