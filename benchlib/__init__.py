"""The legs of bench.py (repo root): eval.py (headline cyc2 eval chain, BASELINE configs[1]), train.py (stage-4 step, configs[2] / [4]),
stress.py (eval chain at hu2048 / ld64 / cyc4), report.py (shared constants and the JSON line)."""
