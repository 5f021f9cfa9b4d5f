"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- ctypes loader for oracle/libpcm_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this package.  The product package never does.
"""
from .loader import Oracle, OracleConfig, OracleResult, build, lib  # noqa: F401
