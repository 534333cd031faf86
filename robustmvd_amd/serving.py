"""Frames in flight: throughput-oriented driver for one process / one GPU.

Every C-ABI entry point of the engine only enqueues work on the caller's stream, so independent frames can be in
flight on different HIP streams of one process.  One frame's small regulariser layers (192-576 workgroups on a 256-CU
device) and the tail of every kernel leave compute units idle that another frame's kernels fill: two frames in flight
give +5-9 % depth maps per second at the headline shape, three +12 %.  Latency per frame does
not improve; the reference's evaluation loop (one frame at a time) maps to depth = 1.
"""
import torch


class _Ticket:
    def __init__(self, outputs, event, keep):
        self._outputs, self._event, self._keep = outputs, event, keep

    def done(self):
        return self._event.query()

    def result(self):
        """Blocks the host until this frame's forward has finished; returns what the model's forward returned."""
        self._event.synchronize()
        self._keep = None
        return self._outputs


class FramePipeline:
    """pipe = FramePipeline(model, depth=2); tickets = [pipe.submit(**sample) for sample in samples];
    preds = [t.result() for t in tickets]."""

    def __init__(self, model, depth=2, device=None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.model = model
        dev = device if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise ValueError("FramePipeline needs a model on a ROCm device")
        self.device = dev
        self.streams = [torch.cuda.Stream(dev) for _ in range(depth)]
        self._next = 0
        # Cold start: the engine's modules pack their weights (and upload sweep constants) lazily, on whatever stream
        # is current at their first forward.  A fresh model's first frames would otherwise race: frame 1 packs on
        # streams[0], frame 2 on streams[1] reuses the cached packed weights before streams[0] has run the pack
        # kernels.  Do all of it now, on the current stream, and finish it before any side-stream work.
        with torch.no_grad():
            for m in model.modules():
                prep = getattr(m, "_prepare", None)
                if callable(prep):
                    prep()
        torch.cuda.current_stream(dev).synchronize()

    @torch.no_grad()
    def submit(self, **sample):
        """Enqueues model(**sample) on the next stream (round-robin) and returns a ticket.  Device tensors of the sample
        must have been produced on the current stream (they are, coming from input_adapter)."""
        s = self.streams[self._next]
        self._next = (self._next + 1) % len(self.streams)
        s.wait_stream(torch.cuda.current_stream(self.device))  # inputs are ready before the frame starts
        with torch.cuda.stream(s):
            out = self.model(**sample)
            ev = torch.cuda.Event()
            ev.record(s)
        return _Ticket(out, ev, sample)  # the ticket keeps the inputs alive until the frame is done

    def drain(self):
        for s in self.streams:
            s.synchronize()
