"""debug: where does a pending HIP error come from before the first dc_* launch of __graft_entry__.smoke()?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetLastError.restype = ctypes.c_int
hip.hipPeekAtLastError.restype = ctypes.c_int
def peek(tag):
    print(f"[dbg] {tag}: hipPeekAtLastError = {hip.hipPeekAtLastError()}", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "build"
import __graft_entry__ as g
peek("start")
if mode == "build":
    g.build(); peek("after build()")
import torch
peek("after import torch")
torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
peek("after first torch cuda op")
from dynamicrafter_amd import _hip
_hip.lib(); peek("after _hip.lib()")
try:
    g.smoke(); peek("after smoke")
except Exception as e:
    print("smoke failed:", e); peek("after failed smoke")
