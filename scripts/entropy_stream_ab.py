"""bench.py's entropy_1080p_batch (16 Python threads calling the coders on device planes) with the three ways a caller can state
what its planes are ordered after: the legacy default stream (what a zero-initialised struct says), a stream per thread, and
MJX_STREAM_IDLE (the caller's promise that nothing is pending).  usage: MODE=null|own|idle python scripts/entropy_stream_ab.py"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--workload", "entropy_1080p_batch", "--steps", "6"]
import torch
import libmodjpeg_amd.entropy as E
from libmodjpeg_amd import _lib
mode = os.environ.get("MODE", "null")
if mode == "idle":
    E.caller_stream = lambda: _lib.MJX_STREAM_IDLE
elif mode == "own":
    local = threading.local()

    def own_stream():
        if not hasattr(local, "s"):
            local.s = torch.cuda.Stream()
        return local.s.cuda_stream
    E.caller_stream = own_stream
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
