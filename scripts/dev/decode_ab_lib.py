"""decode_only.py against an OLDER build of the library (FASTLLAMA_HIP_LIB=...): prototypes the old library does not export are dropped
from the binding first.  usage: FASTLLAMA_HIP_LIB=gpurun_variants/libold.so python scripts/dev/decode_ab_lib.py <decode_only args>"""
import ctypes as C, os, runpy, sys
import torch  # (first: its HIP runtime must be the one the process initialises)
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastllama_amd import hip
lib = C.CDLL(hip.LIB_PATH)
for n in list(hip._PROTOS):
    if not n.startswith("fl_debug_") and not hasattr(lib, n):
        hip._PROTOS.pop(n)
sys.argv = [os.path.join(ROOT, "scripts", "decode_only.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
