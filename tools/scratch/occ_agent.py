import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "attention-models_amd"))
import torch
torch.zeros(1, device="cuda")
from amk import lib
L = lib.load()
f = ctypes.CDLL(lib.LIB_PATH).amk_debug_agent_occupancy
print("occupancy s1", f(0), "s2", f(1), "s2_bwd", f(2), "s1_bwd", f(3))
