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


class PinnedUploader:
    """Host-to-device staging of raw input images for the model adapters ("pinned-memory H2D
    overlap"): the frame's images are copied into page-locked host buffers once (or handed over already pinned) and
    uploaded on a dedicated copy stream, so the transfer of frame i+1 runs under the forward of frame i.

        up = PinnedUploader(device)
        nxt = up.stage(images_0)                      # list of numpy arrays / CPU tensors, any dtype
        for i in range(n):
            cur, nxt = nxt, up.stage(images[i + 1])   # next frame's H2D is in flight
            dev_images = cur.wait()                   # the compute stream waits for the copy (no host sync)
            sample = model.input_adapter(images=dev_images, ...)
    """

    class _Staged:
        def __init__(self, tensors, event, device):
            self._tensors, self._event, self._device = tensors, event, device

        def wait(self):
            """Makes the CURRENT stream wait for the copy and marks the tensors as used on it (they live in the copy stream's
            allocator pool: without record_stream the block could be handed to the next stage() while this stream still
            reads it).  A consumer that passes the tensors on to further streams must call record_stream for those itself."""
            cur = torch.cuda.current_stream(self._device)
            cur.wait_event(self._event)
            for t in self._tensors:
                t.record_stream(cur)
            return self._tensors

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("PinnedUploader needs a ROCm device")
        self.stream = torch.cuda.Stream(self.device)

    @staticmethod
    def pin(images):
        """numpy arrays / CPU tensors -> page-locked CPU tensors (reusable across frames of the same shape)."""
        out = []
        for im in images:
            t = im if isinstance(im, torch.Tensor) else torch.from_numpy(im)
            out.append(t if t.is_pinned() else t.contiguous().pin_memory())
        return out

    def stage(self, images):
        pinned = self.pin(images)
        with torch.cuda.stream(self.stream):
            dev = [t.to(self.device, non_blocking=True) for t in pinned]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        staged = self._Staged(dev, ev, self.device)  # wait() records the consuming stream
        staged._pinned = pinned  # keep the host buffers alive until the copy has been consumed
        return staged
