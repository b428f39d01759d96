"""Measurement scripts that need test infrastructure (the oracle-side VM for valid execution traces, the restated
verifiers as checkers): run as `python tests/perf/<name>.py` on the GPU box.  Not collected by pytest; never imported by
the product, by bench.py or by tools/."""
