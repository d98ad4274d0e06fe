"""GPU side of the data module's per-sample transform (reference dataloader.py:32-36,66-73).

The reference decodes, resizes and converts every image in DataLoader worker processes (PIL + torchvision) and
ships fp32 tensors to the GPU: 602 KB per 224 x 224 image, all of it produced by host cores.  Here host threads
only decode (PIL releases the GIL while it does) into pinned memory; the 8-bit pixels cross PCIe once (3 bytes per
source pixel), and `Resize((p, p))` + `ToTensor()` run as HIP kernels (csrc/image.hip) that reproduce Pillow's
fixed-point resampler bit for bit.  Batches are prepared `prefetch` deep on a side stream by an issuer thread
while the training step runs on the current stream; a batch is handed over with a stream event, never with a
host synchronisation.
"""
import ctypes as C
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ._lib import call, lib

try:
    import pyarrow as _pa
except ImportError:                                       # the numpy route below needs nothing else
    _pa = None

# Pillow changes the order of its two passes on extremely elongated images (seen: more than 100 times taller than
# wide, shrinking vertically); those stay on Pillow itself.
MAX_ASPECT = 64


class GpuImagePipeline:
    def __init__(self, patch_size, device, workers=8, prefetch=2):
        self.p = int(patch_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GpuImagePipeline needs a HIP device (cuda:N)")
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix="vamd-decode")
        self.prefetch = max(1, int(prefetch))
        self.stream = torch.cuda.Stream(device=self.device)
        self._tables = {}

    # ---- host side -------------------------------------------------------------------------------------
    def _decode(self, path):
        """-> (pinned uint8 tensor holding [H][W][bytes per pixel], H, W, bytes per pixel); images Pillow treats
        specially arrive already resized.  Pillow keeps an RGB image as 4 bytes per pixel; its Arrow export hands out
        that memory without a copy, and the copy into pinned memory then runs outside the interpreter lock (numpy's
        route, Image.tobytes, repacks to 3 bytes under the lock and stops the decode threads from scaling)."""
        from PIL import Image
        img = Image.open(path).convert("RGB")
        W, H = img.size
        if max(H, W) > MAX_ASPECT * min(H, W):
            img = img.resize((self.p, self.p), Image.BILINEAR)
            W, H = img.size
        if _pa is not None:
            try:
                arr = _pa.array(img)                      # fails for images Pillow holds in several memory blocks
                buf = arr.flatten().buffers()[1]
                if buf.size >= H * W * 4:
                    pinned = torch.empty(H * W * 4, dtype=torch.uint8, pin_memory=True)
                    C.memmove(pinned.data_ptr(), buf.address, H * W * 4)
                    return pinned, H, W, 4
            except Exception:
                pass
        arr = np.asarray(img, dtype=np.uint8)
        pinned = torch.empty(H * W * 3, dtype=torch.uint8, pin_memory=True)
        pinned.numpy()[:] = arr.reshape(-1)
        return pinned, H, W, 3

    def _table(self, n_in):
        """device tables of one axis resized from n_in to the patch size (None when the size already matches)"""
        if n_in == self.p:
            return None
        t = self._tables.get(n_in)
        if t is None:
            ks = lib.vamd_resize_ksize(n_in, self.p)
            bounds = np.empty((self.p, 2), dtype=np.int32)
            coef = np.empty((self.p, ks), dtype=np.int32)
            call("vamd_resize_coeffs", n_in, self.p, bounds.ctypes.data_as(C.c_void_p), coef.ctypes.data_as(C.c_void_p), ks)
            t = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(coef).to(self.device), ks)
            self._tables[n_in] = t
        return t

    # ---- device side -----------------------------------------------------------------------------------
    def _issue(self, decoded):
        """Uploads and transforms one batch on the pipeline's stream; returns (x [B][3][p][p] fp32, event)."""
        p = self.p
        with torch.cuda.stream(self.stream):
            x = torch.empty((len(decoded), 3, p, p), dtype=torch.float32, device=self.device)
            total = sum(h * w * ps for _, h, w, ps in decoded)
            tmp_total = sum(h * p * 3 for _, h, w, _ps in decoded if w != p)
            raw = torch.empty(max(total, 1), dtype=torch.uint8, device=self.device)
            tmp = torch.empty(max(tmp_total, 1), dtype=torch.uint8, device=self.device)
            st = self.stream.cuda_stream
            off = toff = 0
            for i, (pinned, H, W, ps) in enumerate(decoded):
                n = H * W * ps
                src = raw[off:off + n]
                src.copy_(pinned, non_blocking=True)
                ht, vt = self._table(W), self._table(H)
                call("vamd_image_resize_u8", src.data_ptr(), H, W, ps,
                     ht[0].data_ptr() if ht else None, ht[1].data_ptr() if ht else None, ht[2] if ht else 0,
                     vt[0].data_ptr() if vt else None, vt[1].data_ptr() if vt else None, vt[2] if vt else 0,
                     tmp[toff:].data_ptr() if ht else None, p, p, x[i].data_ptr(), None, st)
                off += n
                if ht:
                    toff += H * p * 3
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return x, ev

    def load_batch(self, paths):
        """One batch, ordered after on the current stream (no host synchronisation)."""
        decoded = list(self.pool.map(self._decode, paths))
        x, ev = self._issue(decoded)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        x.record_stream(cur)
        return x

    def batches(self, chunks):
        """chunks: iterable of lists of paths -> generator of device batches, prepared `prefetch` ahead."""
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def issuer():
            torch.cuda.set_device(self.device)
            try:
                pending = []
                it = iter(chunks)
                done = False
                while not stop.is_set():
                    while not done and len(pending) <= self.prefetch:      # decoding runs ahead of the uploads
                        try:
                            paths = next(it)
                        except StopIteration:
                            done = True
                            break
                        pending.append([self.pool.submit(self._decode, p) for p in paths])
                    if not pending:
                        break
                    decoded = [f.result() for f in pending.pop(0)]
                    item = self._issue(decoded)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                q.put(None)
            except BaseException as e:          # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=issuer, name="vamd-input-issuer", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                x, ev = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                x.record_stream(cur)
                yield x
        finally:
            stop.set()
            while th.is_alive():                 # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)

    def close(self):
        self.pool.shutdown(wait=False)
