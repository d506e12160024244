"""Parts of bench.py (the script at the repo root): workloads, the CPU-baseline leg, the checks, timing, the other
BASELINE configs, HBM traffic, the launcher and the report.  Test and measurement code: the product never imports it."""
