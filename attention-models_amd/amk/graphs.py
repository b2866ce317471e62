"""HIP-graph capture of a whole training step for launch-bound regimes.

At small batch the routed-expert layers are dozens of short launches (ViTMoE, batch 2: ~600 kernels
in 9 ms, most of it launch gaps).  The libamk.so entry points are graph-safe by construction --
asynchronous on the caller's stream, no host synchronisation, no allocation, workspaces owned by
the caller -- so the whole forward + backward + optimizer step can be captured once and replayed.
torch owns the capture (torch.cuda.CUDAGraph = hipGraph on ROCm) and the static memory pool.
"""
import torch


class GraphedStep:
    """Capture ``fn(*static_inputs)`` (which may run forward, backward and optimizer.step) into a HIP
    graph after `warmup` eager runs on a side stream; ``replay(*new_inputs)`` copies the inputs into
    the static buffers and launches the graph.  `fn` must not synchronise with the host; optimizers
    need ``capturable=True``."""

    def __init__(self, fn, static_inputs, warmup=3, before_each=None):
        """before_each: called on the host before every warm-up run (a train step's schedule tick: the warm-up runs are
        real optimizer steps)."""
        self.static_inputs = [t.clone() for t in static_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if before_each is not None:
                    before_each()
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        kw = {}
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # a process group is alive: ProcessGroupNCCL's watchdog thread polls the events of collectives issued EARLIER
            # (the warm-up steps) with hipEventQuery, which a capture in "global" mode forbids to every thread of the
            # process ("operation not permitted when stream is capturing" -> the watchdog aborts the process; seen on
            # MI355X with the 962 MB ViTMoE step).  So: let those collectives finish and the watchdog drop them, and
            # capture in thread-local mode (only this thread's calls are policed; the launches of the autograd threads
            # into the capturing streams are captured all the same).
            import time

            torch.cuda.synchronize()
            time.sleep(0.5)
            kw["capture_error_mode"] = "thread_local"
        with torch.cuda.graph(self.graph, **kw):
            self.static_outputs = fn(*self.static_inputs)

    def replay(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            dst.copy_(src)
        self.graph.replay()
        return self.static_outputs
