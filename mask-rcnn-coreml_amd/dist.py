"""Multi-GPU: shard a batch of images over the ranks of one node, one all-gather of the results.

New functionality asked for by the north star (the reference is single image / single device,
SURVEY.md §8e).  Each image's pipeline is independent, so the batch is split into contiguous blocks
(image b → rank b // ceil(B/G)), weights and anchors are replicated, there is NO collective on the
data path, and the only exchange is one all-gather of fixed-size, zero-padded records per image:

    record = detections (maxDet × 6 f32) ‖ masks (maxDet × S × S f32)      316 000 B at the defaults

over RCCL/xGMI (``torch.distributed`` backend "nccl"); the same code runs on gloo/CPU tensors, which
is how the N > 1 path is tested without GPUs.  Results are identical for any G because the per-image
computation does not depend on the batch it rides in (tests/test_gpu_engine.py::test_engine_batch_independence).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of image indices [lo, hi) owned by `rank` (blocks differ by at most one image)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(det: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(b, D, 6) + (b, D, S, S) → (b, D*6 + D*S*S) contiguous."""
    b = det.shape[0]
    return torch.cat([det.reshape(b, -1), mask.reshape(b, -1)], dim=1).contiguous()


def unpack_records(rec: torch.Tensor, max_det: int, mask_size: int):
    n = rec.shape[0]
    d = rec[:, : max_det * 6].reshape(n, max_det, 6)
    m = rec[:, max_det * 6:].reshape(n, max_det, mask_size, mask_size)
    return d, m


class DetectionGather:
    """Pre-allocated all-gather of the per-image records (equal shards: the bench's weak-scaling case)."""

    def __init__(self, per_rank_batch: int, max_det: int, mask_size: int, world: int, device):
        self.b, self.D, self.S, self.world = per_rank_batch, max_det, mask_size, world
        self.rec_len = max_det * 6 + max_det * mask_size * mask_size
        self.send = torch.empty((per_rank_batch, self.rec_len), dtype=torch.float32, device=device)
        self.recv = torch.empty((world * per_rank_batch, self.rec_len), dtype=torch.float32, device=device)

    def all_gather(self, det: torch.Tensor, mask: torch.Tensor):
        self.send[:, : self.D * 6].copy_(det.reshape(self.b, -1))
        self.send[:, self.D * 6:].copy_(mask.reshape(self.b, -1))
        if self.world == 1:
            self.recv.copy_(self.send)
        elif self.send.is_cuda:
            dist.all_gather_into_tensor(self.recv, self.send)
        else:
            parts = list(self.recv.chunk(self.world, dim=0))
            dist.all_gather(parts, self.send)
        return unpack_records(self.recv, self.D, self.S)


class RemoteRankError(RuntimeError):
    """A rank of the sharded call failed its local predict: raised on EVERY rank after the exchange (csrc/dist.hip: raise_remote_status)."""

    def __init__(self, statuses: List[int], rank: int):
        self.statuses = list(statuses)
        bad = [(r, s) for r, s in enumerate(statuses) if s]
        super().__init__(f"rank(s) {[r for r, _ in bad]} of {len(statuses)} failed their local predict (status {[s for _, s in bad]}); "
                         f"the records of this batch are not valid on any rank" + (": see this rank's earlier message" if statuses[rank] else ""))


def gather_uneven(rec: torch.Tensor, global_batch: int, status=None):
    """All-gather for shards that differ by one image: pad to the largest shard, gather, drop padding.
    status (int, optional): this rank's status word rides behind its padded records (the slot's trailer of csrc/dist.hip); the call then
    returns (records, [status of every rank])."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(global_batch, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    trailer = 0 if status is None else 1
    pad = torch.zeros((mx + trailer, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    pad[: rec.shape[0]] = rec
    if trailer:
        pad[mx, 0] = float(status)
        pad[mx, 1] = float(rec.shape[0])
    parts: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
    if status is None:
        return out
    return out, [int(p[mx, 0].item()) for p in parts]


def predict_sharded(predict_fn, images, max_det: int, mask_size: int):
    """images: the GLOBAL batch (same on every rank).  `predict_fn(images_shard) -> (det, mask)` runs the
    local model.  Returns the detections and masks of the whole batch, in order, on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(images.shape[0], world, rank)
    # The contract of csrc/dist.hip (issue_exchange): a rank-LOCAL failure — a HIP error, the data-dependent fp16-range watchdog — must not
    # keep this rank out of the collective, or every other rank blocks in it.  It becomes the status word of a zeroed slot; every rank
    # raises after the gather.
    status, local = 0, None
    try:
        det, mask = predict_fn(images[lo:hi])
        det = torch.as_tensor(det)
        mask = torch.as_tensor(mask)
    except Exception as e:
        status, local = int(getattr(e, "code", 0) or 2), e
        det = torch.zeros((hi - lo, max_det, 6), dtype=torch.float32)
        mask = torch.zeros((hi - lo, max_det, mask_size, mask_size), dtype=torch.float32)
    rec = pack_records(det, mask)
    if world > 1:
        rec, statuses = gather_uneven(rec, images.shape[0], status)
        if any(statuses):
            raise RemoteRankError(statuses, rank) from local
    elif local is not None:
        raise local
    return unpack_records(rec, max_det, mask_size)


# ---------------------------------------------------------------------------------------------------------------------
# The same exchange behind the C ABI (mask-rcnn-coreml_amd/csrc/dist.hip): ncclAllGather called from librccl directly on the
# model's stream — what a Swift / C host links against.  torch is not involved; the 128-byte rendezvous id travels by
# whatever channel the host has (bench.py: a torch.distributed broadcast; examples/maskrcnn_predict_mgpu.c: a file).
# ---------------------------------------------------------------------------------------------------------------------
class NativeDist:
    """mrcnn_dist handle: one per process / GPU."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes as C
        from . import _lib
        assert len(unique_id) == 128
        self._lib, self._C = _lib, C
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().mrcnn_dist_init(rank, world, buf, C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().mrcnn_dist_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def shard(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
        import ctypes as C
        from . import _lib
        lo, hi = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().mrcnn_dist_shard(global_batch, world, rank, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def close(self):
        if self._h:
            self._lib.lib().mrcnn_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather_records(self, model, det, mask, global_batch: int, out_det, out_mask):
        """det/mask: this rank's results; out_*: (global_batch, ...) — all numpy (host) or all torch CUDA tensors (device)."""
        import numpy as np
        host = isinstance(out_det, np.ndarray)
        ptr = (lambda a: a.ctypes.data) if host else (lambda t: t.data_ptr())
        self._lib.check(self._lib.lib().mrcnn_dist_all_gather_records(
            self._h, model._h, None if det is None else ptr(det), None if mask is None else ptr(mask), global_batch,
            self._lib.HOST if host else self._lib.DEVICE, ptr(out_det), ptr(out_mask)))
        return out_det, out_mask

    def all_gather_records_async(self, model, det, mask, global_batch: int, out_det, out_mask):
        """The same exchange on the handle's own stream behind the model's stream (torch CUDA tensors only): returns at
        once; the model's next predict overlaps it.  Join with wait()."""
        self._lib.check(self._lib.lib().mrcnn_dist_all_gather_records_async(
            self._h, model._h, det.data_ptr(), mask.data_ptr(), global_batch, out_det.data_ptr(), out_mask.data_ptr()))

    def wait(self):
        self._lib.check(self._lib.lib().mrcnn_dist_wait(self._h))

    def recovered(self):
        """[range recoveries of rank r's predict] of the last completed sharded predict (mrcnn_dist_recovered): any non-zero entry means that
        rank lowered its split exponents — redistribute the element-wise minimum of the vectors for rank-independent per-image results."""
        C = self._C
        per = (C.c_int32 * self.world)()
        n = C.c_int(0)
        self._lib.check(self._lib.lib().mrcnn_dist_recovered(self._h, per, C.byref(n)))
        return list(per)

    @staticmethod
    def plan(global_batch: int, world: int, max_det: int, mask_size: int):
        """[(begin, end, slot offset in floats, record floats)] per rank and the slot size — host arithmetic of dist.hip."""
        import ctypes as C
        from . import _lib
        table = (C.c_int64 * (4 * world))()
        slot = C.c_int64(0)
        _lib.check(_lib.lib().mrcnn_dist_plan(global_batch, world, max_det, mask_size, table, C.byref(slot)))
        return [tuple(table[4 * r:4 * r + 4]) for r in range(world)], slot.value

    @staticmethod
    def simulate_host(dets, masks, global_batch: int, max_det: int, mask_size: int, status=None):
        """dist.hip's pack -> all-gather (concatenation) -> unpack on host arrays: dets[r] / masks[r] = rank r's local results
        (numpy float32).  Returns (detections, masks, statuses) of the whole batch as rank 0 would hold them."""
        import ctypes as C
        import numpy as np
        from . import _lib
        world = len(dets)
        dk = [np.ascontiguousarray(d, np.float32) for d in dets]
        mk = [np.ascontiguousarray(m, np.float32) for m in masks]
        dp = (C.c_void_p * world)(*[a.ctypes.data if a.size else None for a in dk])
        mp = (C.c_void_p * world)(*[a.ctypes.data if a.size else None for a in mk])
        out_d = np.full((global_batch, max_det, 6), np.nan, np.float32)
        out_m = np.full((global_batch, max_det, mask_size, mask_size), np.nan, np.float32)
        st_in = None if status is None else np.ascontiguousarray(status, np.int32)
        st_out = np.zeros(world, np.int32)
        _lib.check(_lib.lib().mrcnn_dist_simulate_host(world, global_batch, max_det, mask_size, dp, mp,
                                                       None if st_in is None else st_in.ctypes.data, out_d.ctypes.data, out_m.ctypes.data,
                                                       st_out.ctypes.data))
        return out_d, out_m, st_out

    def predict_sharded(self, model, images):
        """images: the GLOBAL batch (B,H,W,3) uint8, identical on every rank — numpy or torch CUDA tensor."""
        import numpy as np
        B, H, W, _ = images.shape
        D, S = model.max_detections, model.mask_size
        if isinstance(images, np.ndarray):
            imgs = np.ascontiguousarray(images, dtype=np.uint8)
            det = np.empty((B, D, 6), np.float32)
            mask = np.empty((B, D, S, S), np.float32)
            self._lib.check(self._lib.lib().mrcnn_maskrcnn_predict_sharded(self._h, model._h, imgs.ctypes.data, B, H, W, self._lib.HOST,
                                                                           det.ctypes.data, mask.ctypes.data))
            return det, mask
        det = torch.empty((B, D, 6), dtype=torch.float32, device=images.device)
        mask = torch.empty((B, D, S, S), dtype=torch.float32, device=images.device)
        self._lib.check(self._lib.lib().mrcnn_maskrcnn_predict_sharded(self._h, model._h, images.data_ptr(), B, H, W, self._lib.DEVICE,
                                                                       det.data_ptr(), mask.data_ptr()))
        return det, mask
