"""CPU ORACLE package -- test infrastructure only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.
The product (``flingbot_amd``) never does.
"""
from .flex import OracleSim, build_oracle, oracle_lib_path, eval_rsqrt, rsqrt_table  # noqa: F401
