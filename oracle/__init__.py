"""CPU oracle loader -- TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package (see dslam_oracle.cpp header)."""
import os
import subprocess

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


def build(verbose=False):
    res = subprocess.run(["make", "-C", ORACLE_DIR], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building liboracle.so failed")
    return LIB_PATH


def open_oracle(capi_cls, threads=1):
    """Bind liboracle.so through the product's generic ctypes wrapper class (prefix oracle_)."""
    if not os.path.exists(LIB_PATH):
        build()
    o = capi_cls(LIB_PATH, "oracle_", has_engine_device=False)
    o.set_threads(threads)
    return o
