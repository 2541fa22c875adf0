"""oracle/ -- CPU restatements of the reference's algorithms for the hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product path (mav-detection_amd/) must never import anything from here.
"""
