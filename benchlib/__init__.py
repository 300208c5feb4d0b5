"""The parts of bench.py (the benchmark driver at the repo root): see bench.py."""
