"""Double-buffered device feeder for the SR training step.

The reference converts every sample to an fp32 CHW tensor on DataLoader worker CPUs (np2tensor, dataops/common.py:
470-499) after the paired flip / rot90 (dataops/augmentations.py:790-830), collates, pins, and `feed_data` then blocks
on the fp32 H2D copy (sr_model.py:115-128).  Here the wire format stays what OpenCV produced -- uint8 HWC BGR crop
windows, 4x fewer bytes over PCIe -- and the flip / rot90 / np2tensor work runs as ONE kernel per batch
(csrc/feed.hip) on a side HIP stream:

    host batch k+1 --memcpy--> pinned staging --async H2D (copy stream)--> uint8 on device --tnr_feed_u8_to_tensor-->
    fp32 NCHW batch, ready event                      ...while the compute stream runs step k

`for data in DeviceFeeder(loader, device): model.feed_data(data); model.optimize_parameters(step)` keeps the
reference's batch-dict contract ({'LR', 'HR', 'LR_path', 'HR_path'}, data/aligned_dataset.py:166-175): the values are
device fp32 tensors, so SRModel.feed_data's `.to(device)` is a no-op.  Two slots of buffers rotate; a slot is refilled
only after the compute stream passed the point where the consumer asked for the next batch (event-ordered, the host
never blocks on the GPU).
"""

import numpy as np
import torch

from .. import hip

IMAGE_KEYS = ("LR", "HR", "A", "B", "ref")


class _Slot:
    def __init__(self):
        self.pinned, self.dev_u8, self.dev_flags, self.out = {}, {}, {}, {}
        self.h2d = {}              # key -> event recorded after the async upload out of this slot's pinned staging buffer
        self.ready = torch.cuda.Event()
        self.free = None           # recorded on the compute stream when the consumer moved on
        self.batch = None


def _as_host_tensor(v):
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(np.ascontiguousarray(v))
    return v


class DeviceFeeder:
    def __init__(self, loader, device=None, znorm=False, data_range=1.0, bgr2rgb=True, depth=2, degrade=None):
        """degrade: optional callable HR batch -> LR batch (dataops.degradations.RealESRGANDegradation): batches that
        carry no 'LR' get it synthesised from 'HR' on the copy stream (augs_strategy: resrgan)."""
        hip.require_device()
        self.loader = loader
        self.degrade = degrade
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.znorm, self.data_range, self.bgr2rgb = bool(znorm), float(data_range), bool(bgr2rgb)
        self.depth = max(2, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device, priority=-1)     # ahead of the step's kernels: its launches are few and small
        self.bytes_uploaded = 0
        # the slots (device buffers, pinned staging, `free` events) live as long as the feeder: a new epoch's first uploads wait
        # for the step that consumed the slot last, and no buffer goes back to the copy stream's allocator pool while the
        # compute stream may still read it
        self._slots = [_Slot() for _ in range(self.depth)]

    def __len__(self):
        return len(self.loader)

    # ------------------------------------------------------------------ one batch -> one slot
    def _stage(self, slot, key, host):
        """host uint8 [N,H,W,C] (or fp32 [N,C,H,W]) -> pinned staging -> device, on the copy stream."""
        host = _as_host_tensor(host)
        pin = slot.pinned.get(key)
        if pin is None or pin.shape != host.shape or pin.dtype != host.dtype:
            pin = torch.empty(host.shape, dtype=host.dtype).pin_memory()
            slot.pinned[key] = pin
            slot.dev_u8[key] = torch.empty(host.shape, dtype=host.dtype, device=self.device)
        if host.is_pinned():
            pin = host                                       # the loader pinned it already (DataLoader(pin_memory=True))
        else:
            ev = slot.h2d.get(key)
            if ev is not None:
                ev.synchronize()                             # the previous upload out of this staging buffer has executed
            pin.copy_(host)                                  # host memcpy into page-locked memory
        dev = slot.dev_u8[key]
        dev.copy_(pin, non_blocking=True)
        if pin is not host:
            ev = slot.h2d.get(key)
            if ev is None:
                ev = slot.h2d[key] = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.bytes_uploaded += host.numel() * host.element_size()
        return dev

    def _upload(self, slot, batch):
        lib = hip.load()
        out = {}
        with torch.cuda.stream(self.copy_stream):
            if slot.free is not None:
                self.copy_stream.wait_event(slot.free)       # the step that consumed this slot's tensors is past
            def device_flags(name):
                """-> (device int32 flags or None, any rotation in the batch) for `flags` or a per-image `flags_<key>` entry."""
                flags = batch.get(name)
                if flags is None:
                    return None, 0
                f = _as_host_tensor(flags).to(torch.int32).contiguous()
                ent = slot.dev_flags.get(name)
                if ent is None or ent[0].numel() != f.numel():
                    ent = slot.dev_flags[name] = (torch.empty(f.numel(), dtype=torch.int32, device=self.device),
                                                  torch.empty(f.numel(), dtype=torch.int32).pin_memory(), torch.cuda.Event())
                    ent[2].record(self.copy_stream)
                d, pin, ev = ent
                # through page-locked memory and asynchronous: a pageable (blocking) copy here made the HOST wait for the copy stream,
                # which itself waits for the step that consumed this slot (slot.free) -- the launch queue ran dry once per step
                # (bench.py --feed paired: 253.3 vs 244.1 ms, profiles/r03ag_bench_variants.txt)
                ev.synchronize()                             # the slot's previous flag upload (two steps ago) has executed
                pin.copy_(f)
                d.copy_(pin, non_blocking=True)
                ev.record(self.copy_stream)
                return d, int(bool((f & 2).any()))

            shared = device_flags("flags")
            for key, val in batch.items():
                if key not in IMAGE_KEYS:
                    if key != "flags" and not key.startswith("flags_"):
                        out[key] = val
                    continue
                # unpaired datasets (mode: unaligned) carry one flag word per image: flags_A / flags_B
                dflags, any_rot = device_flags("flags_" + key) if ("flags_" + key) in batch else shared
                host = _as_host_tensor(val)
                if host.dtype == torch.uint8:
                    if host.dim() != 4:
                        raise ValueError("feeder: uint8 image batches are [N,H,W,C], got %s for %r" % (tuple(host.shape), key))
                    dev = self._stage(slot, key, host)
                    N, H, W, C = dev.shape
                    o = slot.out.get(key)
                    if o is None or o.shape != (N, C, H, W):
                        o = torch.empty((N, C, H, W), dtype=torch.float32, device=self.device)
                        slot.out[key] = o
                    hip.check(lib.tnr_feed_u8_to_tensor(dev.data_ptr(), N, H, W, C, hip.ptr(dflags), any_rot, o.data_ptr(),
                                                        int(self.bgr2rgb), self.data_range, int(self.znorm), hip.stream()),
                              "feed_u8_to_tensor")
                    out[key] = o
                elif host.dtype == torch.float32:            # a loader that already produced fp32 CHW tensors
                    out[key] = self._stage(slot, key, host)
                else:
                    raise TypeError("feeder: %r batches must be uint8 HWC or float32 CHW, got %s" % (key, host.dtype))
            if self.degrade is not None and "LR" not in out and "HR" in out:
                if self.znorm:
                    raise NotImplementedError("on-device degradations expect images in [0, 1] (znorm: false)")
                lr = self.degrade(out["HR"])
                o = slot.out.get("LR")
                if o is None or o.shape != lr.shape:
                    o = torch.empty_like(lr)
                    slot.out["LR"] = o
                o.copy_(lr)                                   # persistent slot buffer: no allocator reuse across streams
                out["LR"] = o
                if "HR_path" in out and "LR_path" not in out:
                    out["LR_path"] = out["HR_path"]
            slot.ready.record(self.copy_stream)
        slot.batch = out

    # ------------------------------------------------------------------ iteration
    def __iter__(self):
        """Batches in loader order.  The loader iteration, the uploads and the on-device preparation (crop / flip / degradations) run in a
        worker thread on the copy stream: whatever blocks the HOST there -- pageable copies of per-sample parameters, the wait for the
        step that last read a slot -- blocks that thread (the GIL is released inside the runtime calls), never the thread that
        launches the training step.  The consumer hands a slot back (with a fresh `free` event) when it returns for the next batch."""
        import queue
        import threading
        hip.load()                          # the library handle is created on THIS thread, before the worker can race for it
        it = iter(self.loader)              # ... and so is the loader's iterator (a DataLoader starts its worker processes and installs
                                            # its signal handling here: that belongs to the consumer's thread, not to ours)
        todo, done = queue.Queue(), queue.Queue()
        STOP = object()

        def work():
            torch.cuda.set_device(self.device)
            try:
                while True:
                    slot = todo.get()
                    if slot is STOP:
                        return
                    b = next(it, None)
                    if b is None:
                        done.put(STOP)
                        return
                    self._upload(slot, b)
                    done.put(slot)
            except BaseException as e:      # surfaces in the consumer's thread
                done.put(e)

        worker = threading.Thread(target=work, name="tnr-feeder", daemon=True)
        worker.start()
        for slot in self._slots:
            todo.put(slot)
        out = set(self._slots)              # slots whose tensors may still be read (handed to the worker or the consumer)
        s = None
        try:
            while True:
                s = done.get()
                if s is STOP:
                    s = None
                    break
                if isinstance(s, BaseException):
                    e, s = s, None
                    raise e
                torch.cuda.current_stream(self.device).wait_event(s.ready)
                yield s.batch
                # the consumer came back for the next batch: everything that reads this slot is enqueued by now
                self._mark_free(s)
                todo.put(s)
                s = None
        finally:
            # also when the consumer leaves early (break / exception: GeneratorExit lands at the yield): stop the worker, then give
            # every slot a fresh `free` event so that the next epoch's uploads wait for whatever still reads them
            todo.put(STOP)
            worker.join()
            for q in out:
                self._mark_free(q)

    def iterate(self):
        """iter(self), remembered so that close() can end it (bench / trainer: an abandoned iterator would only be torn down at
        garbage collection or interpreter exit, where its `finally` would make runtime calls)."""
        self.close()
        self._active = iter(self)
        return self._active

    def close(self):
        """Stop the worker thread of the iterator handed out by iterate() and release its slots (idempotent)."""
        it, self._active = getattr(self, "_active", None), None
        if it is not None:
            it.close()                      # GeneratorExit at the yield -> the generator's `finally`

    def _mark_free(self, s):
        if s.free is None:
            s.free = torch.cuda.Event()
        s.free.record(torch.cuda.current_stream(self.device))
