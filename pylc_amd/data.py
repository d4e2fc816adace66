"""Host -> device tile feeder (the GPU-side end of the reference's loader: db/buffer.py:53-92, db/dataset.py:75-115).

The reference converts every uint8 tile to float32 on the host (buffer.py:62), normalises on the CPU and then copies
3.1 MB per 512^2 RGB tile to the device (models/model.py:301-303).  Here tiles cross PCIe as the uint8 the database
stores (0.79 MB), through two pinned staging buffers on a dedicated copy stream so the copy of batch i+1 overlaps the
step of batch i; normalisation happens on the GPU (ops.image_pack).  Reading HDF5 files is out of scope: any iterable of
(img uint8 [B,C,H,W], mask uint8/int64 [B,H,W]) numpy arrays or tensors can be fed."""
import torch


class TileFeeder:
    def __init__(self, batches, device, depth=2):
        self.batches = iter(batches)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.depth = depth
        self._staging = []           # per slot: (pinned img, pinned mask)
        self._slot_events = {}       # per slot: event of the last H2D copy that READ the pinned buffers
        self._queue = []             # in flight: (dev img, dev mask, event)
        self._slot = 0

    def _pin(self, slot, img, mask):
        while len(self._staging) <= slot:
            self._staging.append(None)
        cur = self._staging[slot]
        if cur is None or cur[0].shape != img.shape or cur[0].dtype != img.dtype or cur[1].shape != mask.shape or cur[1].dtype != mask.dtype:
            cur = (torch.empty(img.shape, dtype=img.dtype).pin_memory(), torch.empty(mask.shape, dtype=mask.dtype).pin_memory())
            self._staging[slot] = cur
        # The slot rotation only guarantees that the slot's previous batch was handed to the consumer on the HOST; the
        # non_blocking copy out of these pinned buffers may still be queued (the host can run steps ahead of the GPU, and a
        # copy stream can share a hardware queue with the compute stream).  Wait for that copy before overwriting its source.
        ev = self._slot_events.get(slot)
        if ev is not None:
            ev.synchronize()
        cur[0].copy_(img)
        cur[1].copy_(mask)
        return cur

    def _enqueue(self):
        try:
            img, mask = next(self.batches)
        except StopIteration:
            return False
        img, mask = torch.as_tensor(img), torch.as_tensor(mask)
        if len(self._queue) >= self.depth:
            raise RuntimeError('feeder queue overflow')
        slot = self._slot
        self._slot = (self._slot + 1) % (self.depth + 1)          # a slot is reused only after its batch was consumed
        pimg, pmask = self._pin(slot, img, mask)
        with torch.cuda.stream(self.stream):
            d_img = pimg.to(self.device, non_blocking=True)
            d_mask = pmask.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot_events[slot] = ev
        self._queue.append((d_img, d_mask, ev))
        return True

    def __iter__(self):
        while len(self._queue) < self.depth and self._enqueue():
            pass
        while self._queue:
            d_img, d_mask, ev = self._queue.pop(0)
            torch.cuda.current_stream(self.device).wait_event(ev)      # compute waits for this batch's copy only
            d_img.record_stream(torch.cuda.current_stream(self.device))
            d_mask.record_stream(torch.cuda.current_stream(self.device))
            self._enqueue()                                            # start the next copy while this batch computes
            yield d_img, d_mask
