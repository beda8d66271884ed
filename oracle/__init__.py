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


NATIVE_LIB_PATH = os.path.join(ORACLE_DIR, "_native", "liboracle_native.so")


def open_native_oracle(capi_cls, threads=1):
    """The -O3 -march=native build bench.py times as the CPU baseline; compiled on first use ON THIS MACHINE (the file
    is specific to the host's CPU and never travels).  Falls back to the portable build if the compiler is missing."""
    res = subprocess.run(["make", "-C", ORACLE_DIR, "_native/liboracle_native.so"], capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(NATIVE_LIB_PATH):
        return open_oracle(capi_cls, threads), "-O2 -march=x86-64-v2 (portable build; native build failed)"
    o = capi_cls(NATIVE_LIB_PATH, "oracle_", has_engine_device=False)
    o.set_threads(threads)
    return o, "-O3 -march=native -fopenmp -ffp-contract=off"
