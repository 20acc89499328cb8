"""Largest measured relative error per GPU parity test (FF_TOL_REPORT=<file> python -m pytest tests -m gpu), to set tests/util.py's tolerances from data:
    python tools/tol_report.py gpurun_out/tol/report.tsv"""
import sys
from collections import defaultdict

worst = defaultdict(float)
for line in open(sys.argv[1]):
    test, v = line.rstrip("\n").split("\t")
    worst[test] = max(worst[test], float(v))
for test, v in sorted(worst.items(), key=lambda kv: -kv[1]):
    print(f"{v:.3e}  {test}")
