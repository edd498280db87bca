"""hipGraph capture of a whole forward pass.

A forward is ~50-300 short launches (ResNet-50: 60 kernels of 20-300 us); replaying them as one hipGraph
removes the per-launch host cost and the inter-kernel gaps.  Every entry point of libtlxmi.so is
capturable by construction (no allocation, no synchronisation, no host reads inside a call), and the
activations torch allocates during capture live in the graph's private pool."""
import torch


class GraphedForward:
    """g = GraphedForward(model, example_input); y = g(x) replays the captured forward on x's data.
    `x` must have the example's shape/dtype; the returned tensor is overwritten by the next replay."""

    def __init__(self, model, example, warmup=2):
        assert example.is_cuda, "hipGraph capture needs a device tensor"
        self.model = model
        self.static_in = example.clone()
        self.warmup = warmup
        self._capture()

    def _capture(self):
        model, warmup = self.model, self.warmup
        self.epoch = getattr(model, "_weights_epoch", 0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # packs filters, folds BatchNorm, raises LDS limits: all outside capture
                model(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = model(self.static_in)

    def __call__(self, x=None):
        if getattr(self.model, "_weights_epoch", 0) != self.epoch:
            # weights were (re)loaded or moved since the capture: the graph still points at the old packed filters.
            # (In-place edits of single parameters are not seen here — call recapture() after them.)
            self._capture()
        if x is not None and x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out

    def recapture(self):
        """Capture again after the model's weights changed in a way the loaders do not see (in-place parameter edits)."""
        self._capture()
