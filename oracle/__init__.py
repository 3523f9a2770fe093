"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

The product package ``open_knowledge_graph_embeddings_amd`` must never import from here.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force=False):
    """Compile oracle/kge_oracle_c.c with gcc (via oracle/Makefile)."""
    so = os.path.join(_HERE, "libkge_oracle_c.so")
    src = os.path.join(_HERE, "kge_oracle_c.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libkge_oracle_c.so"])
    return so


def load_c():
    """ctypes handle to the plain-C oracle (built on demand)."""
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_c())
        i64, u8p = ctypes.c_int64, ctypes.POINTER(ctypes.c_uint8)
        lib.okge_oracle_filtered_ranks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, i64, i64, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.okge_oracle_philox_keep.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, i64, i64,
                                                ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        lib.okge_oracle_score_prefix.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, i64, i64, i64, ctypes.c_void_p]
        del u8p
        _LIB = lib
    return _LIB
