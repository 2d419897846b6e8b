"""CPU oracle (test infrastructure only -- see oracle/ofxcv_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
