"""hipGraph replay of a whole forward pass.

At small batch the path is launch-bound (about 130 kernel launches per utterance, each a few microseconds of GPU work at
B = 1), so the launch sequence is captured once per input shape into a hipGraph (torch.cuda.CUDAGraph is hipGraph on
ROCm; our kernels are enqueued through the C ABI on torch's current stream, which is the capture stream) and replayed
with one call.  Inputs are copied into static buffers, outputs are the graph's static tensors (clone them to keep them
across replays).  Inference only; the cache key is (input shapes, operand formats, weights generation).  Parameters updated IN PLACE by
anything but optim.FlatAdamW (which advances the weights generation) are not seen: drop the GraphedForward then."""
import torch


class GraphedForward:
    def __init__(self, fn, warmup=2):
        """fn(*tensors) -> tensor | tuple | dict of tensors; must not synchronise with the host."""
        self.fn, self.warmup = fn, warmup
        self._cache = {}

    def _key(self, args):
        # shapes + the operand formats / weights generation in force (ops.policy_key): a graph captured before an optimiser
        # step or a precision change would replay stale packed weights
        from . import ops
        return (ops.policy_key(),) + tuple((tuple(a.shape), a.dtype, a.device.index) for a in args)

    def __call__(self, *args):
        key = self._key(args)
        ent = self._cache.get(key)
        if ent is None:
            ent = self._capture(args)
            self._cache[key] = ent
        static_in, graph, static_out = ent
        for s, a in zip(static_in, args):
            s.copy_(a, non_blocking=True)
        graph.replay()
        return static_out

    def _capture(self, args):
        if not all(a.is_cuda for a in args):
            raise RuntimeError("GraphedForward: device tensors only")
        static_in = [a.clone() for a in args]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():          # constants, packed weights, allocator pools: outside the graph
            for _ in range(self.warmup):
                self.fn(*static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = self.fn(*static_in)
        return static_in, graph, static_out
