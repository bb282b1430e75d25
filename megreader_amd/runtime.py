"""hipGraph capture of a whole training step (forward + loss + backward + fused optimizer update).

The per-step host cost of the eager path (~150 kernel launches through Python autograd, ~5 ms) is comparable to the
GPU time of a CRNN step on MI355X; replaying one captured graph removes it.  Works because every kernel of the path
is launched on torch's current stream through the C ABI (captured like any other launch), the fused optimizers keep
their step counter / hyper-parameters on the device, and the library's only lazily-created state (the zero page) is
created at load time (`mr_init`).

    step = GraphedTrainStep(model_fn, optimizer, static_inputs)   # model_fn(*static_inputs) -> scalar loss tensor
    for batch in loader:
        step.copy_inputs(*batch)      # device-to-device / H2D copies into the static buffers
        loss = step()                 # graph replay; returns the static loss tensor (no host sync)
"""
import torch


class GraphedTrainStep(object):
    def __init__(self, loss_fn, optimizer, static_inputs, warmup=3):
        self.loss_fn = loss_fn
        self.optimizer = optimizer
        self.inputs = list(static_inputs)
        self.graph = None
        self.loss = None
        self._warmup = warmup
        self._capture()

    def _eager(self):
        self.optimizer.zero_grad()
        loss = self.loss_fn(*self.inputs)
        loss.backward()
        self.optimizer.step()
        return loss

    def _capture(self):
        # warm-up on a side stream (lazy initialisations, allocator pools, flat optimizer buffers) as torch requires
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def copy_inputs(self, *tensors):
        for dst, src in zip(self.inputs, tensors):
            dst.copy_(src, non_blocking=True)

    def __call__(self):
        self.graph.replay()
        return self.loss
