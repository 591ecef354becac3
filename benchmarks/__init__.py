"""Bench legs of bench.py (the driver entry at the repo root).  Not part of the product package."""
