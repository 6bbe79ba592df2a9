"""The legs of bench.py, one module each (bench.py keeps the argument parsing, the rank set-up and the one JSON line)."""
