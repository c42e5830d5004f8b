"""Test infrastructure: CPU restatement of the reference hot path (see oracle/Makefile, oracle/binding.py)."""
