import ctypes, sys, os
sys.path.insert(0, "attention-models_amd")
import torch
torch.zeros(1, device="cuda")
from amk import lib
L = ctypes.CDLL(lib.LIB_PATH)
print("occupancy bk16", L.amk_debug_dense_occupancy(16), "bk32", L.amk_debug_dense_occupancy(32))
p = torch.cuda.get_device_properties(0)
print(p.multi_processor_count, getattr(p, "shared_memory_per_multiprocessor", None), getattr(p, "max_threads_per_multi_processor", None), p.shared_memory_per_block)
