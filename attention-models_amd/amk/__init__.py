"""amk -- MI355X-native (gfx950) attention / MoE / VQ kernels behind the attention-models API.

``amk.lib``    ctypes binding of libamk.so (C ABI: include/amk.h)
``amk.ops``    torch.autograd wrappers (device memory + streams + graph only)
``amk.models`` the reference's nn.Module classes, same names / signatures / state_dict keys
``amk.dp``     data-parallel gradient reducer (bucketed RCCL all-reduce on a side HIP stream)
"""
__version__ = "0.1.0"
