"""Row sharding of a frame across GPUs (one process per GPU, torch.distributed over RCCL/xGMI).

The reference has no distributed layer (OpenMP shared memory only); this is
the MI355X-native counterpart of its `#pragma omp parallel for` over rows for frames that span
the 8 GPUs of a node:

* grayscale (monolithic/src/parallel_to_grayscale.c:7-16) is embarrassingly parallel: each rank
  converts its own contiguous block of rows, no communication;
* Sobel (monolithic/src/sobel.c:10-24) needs one luma row from each neighbour: a `width`-byte
  send/recv pair per neighbour (RCCL ncclSend/ncclRecv via torch.distributed P2P, grouped in one
  batch), overlapped with the interior rows, which do not depend on the halo;
* the 3x3 convolution (old/parallel_convolution.c:8-24) needs one RGB row from each neighbour, the same exchange step;
* the multi-pass gray+Sobel pipeline (monolithic/src/main_with_sobel.c:49-76) needs that exchange once PER PASS: pass k+1
  filters what pass k wrote, so a shard's edge rows depend on the neighbours' pass-k edge rows;
* the channel sums (old/parallel_avg_pixel.c) are partial sums + one 24-byte all-reduce, the mean-grey min / max
  (old/parallel_to_grayscale.c:12,36-37) one min and one max all-reduce of an int32.

The pixel work is delegated to a `backend` object so the same code runs on the GPU (HipBackend ->
libgrayhip.so, the product path) and, in the CPU test-suite only, on a checker backend over gloo.
"""
from __future__ import annotations

from dataclasses import dataclass


def row_partition(height: int, world: int) -> list[tuple[int, int]]:
    """Contiguous row blocks [y0, y1) per rank; remainder rows go to the LAST ranks."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, rem = divmod(height, world)
    out, y = [], 0
    for r in range(world):
        rows = base + (1 if r >= world - rem else 0)
        out.append((y, y + rows))
        y += rows
    return out


class HipBackend:
    """Product backend: gfx950 kernels through the C ABI, tensors resident in HBM."""

    name = "hip"

    def __init__(self, mode: int = 0):
        from . import hostapi
        self.api = hostapi
        self.mode = mode

    def gray_to_plane(self, rgb, out):
        return self.api.gray_to_plane(rgb, self.mode, out=out)

    def sobel_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        return self.api.sobel_rows(buf, out, rows, top_is_border, bottom_is_border)

    def fused_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        return self.api.gray_sobel_fused_rows(buf, out, rows, self.mode, top_is_border, bottom_is_border)

    def sobel_of_gray_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        """Sobel of lum(e, e, e) of a band of EDGE rows (a later pass of the multi-pass pipeline), one launch."""
        return self.api.sobel_of_gray_rows(buf, out, rows, self.mode, top_is_border, bottom_is_border)

    def sum_rgb(self, rgb):
        return self.api.sum_rgb_dev(rgb)

    def conv_rows(self, buf, out, rows, kernel, norm, top_is_border, bottom_is_border):
        return self.api.conv3x3_rows(buf, out, rows, kernel, norm, top_is_border, bottom_is_border)

    def gray_of_plane(self, plane, out):
        return self.api.gray_of_plane(plane, self.mode, out=out)

    def replicate_plane(self, plane, img):
        return self.api.replicate_plane(plane, img)

    def extract_plane(self, img, out):
        return self.api.extract_plane(img, out=out)

    def mean_gray_minmax(self, rgb, out):
        """-> (min, max) of the shard as a device int32[2] tensor (256 / -1 for an empty shard)."""
        return self.api.mean_gray_minmax_dev(rgb, out)[1]

    # Pre-bound launchers: at 8 GPUs a shard's kernels take ~30 us, so per-step Python overhead matters.
    # Pointers, sizes and flags are resolved once; the returned closure is one ctypes call.
    def bind_sobel_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        import torch
        lib, chk = self.api.abi.load(), self.api.check
        w = buf.shape[1]
        assert buf.shape[0] == rows + 2 and tuple(out.shape) == (rows, w) and buf.is_contiguous() and out.is_contiguous()
        src, dst, t, b = buf.data_ptr() + w, out.data_ptr(), int(top_is_border), int(bottom_is_border)
        fn = lib.gh_sobel_u8_rows_dev
        return lambda: chk(fn(src, dst, w, rows, t, b, torch.cuda.current_stream().cuda_stream), "sobel_rows")

    def bind_fused_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        import torch
        lib, chk = self.api.abi.load(), self.api.check
        w, c = buf.shape[1], buf.shape[2]
        assert buf.shape[0] == rows + 2 and tuple(out.shape) == (rows, w) and buf.is_contiguous() and out.is_contiguous()
        src, dst, t, b, mode = buf.data_ptr() + w * c, out.data_ptr(), int(top_is_border), int(bottom_is_border), self.mode
        fn = lib.gh_gray_sobel_fused_rows_dev
        return lambda: chk(fn(src, dst, w, rows, c, mode, t, b, torch.cuda.current_stream().cuda_stream), "fused_rows")

    def bind_gray_to_plane(self, rgb, out):
        import torch
        lib, chk = self.api.abi.load(), self.api.check
        h, w, c = rgb.shape
        assert rgb.is_contiguous() and out.is_contiguous() and out.numel() == h * w
        src, dst, mode = rgb.data_ptr(), out.data_ptr(), self.mode
        fn = lib.gh_gray_rgb8_to_plane_dev
        return lambda: chk(fn(src, dst, h * w, c, mode, torch.cuda.current_stream().cuda_stream), "gray_to_plane")


class NativeHaloComm:
    """RCCL communicator owned by libgrayhip (csrc/comm.hip): the halo exchange and the whole sharded step are issued
    from C++, one ctypes call per step, instead of torch.distributed P2P ops (~100 us of host time per step, which at
    8 GPUs is several times the shard's kernel time).  torch.distributed is used once, to broadcast the 128-byte
    ncclUniqueId; `group` may be any initialised process group (nccl or gloo)."""

    ID_BYTES = 128

    def __init__(self, rank: int, world: int, group=None):
        import ctypes as C
        from . import abi
        self.abi, self.lib = abi, abi.load()
        self.rank, self.world = rank, world
        ident = (C.c_uint8 * self.ID_BYTES)()
        if rank == 0:
            abi.check(self.lib.gh_comm_unique_id(ident), "gh_comm_unique_id")
        if world > 1:
            import torch.distributed as dist
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(box[0])
        handle = C.c_void_p()
        abi.check(self.lib.gh_comm_init(ident, rank, world, C.byref(handle)), "gh_comm_init")
        self.handle = handle

    LOOPBACK_OFF, LOOPBACK_MIDDLE, LOOPBACK_TOP, LOOPBACK_BOTTOM = 0, 1, 2, 3

    def set_loopback(self, mode=1):
        """One-GPU rehearsal (gh_comm_set_loopback): the sharded steps treat this rank as a MIDDLE shard whose neighbours are itself
        (1 / True), as the TOP shard -- frame border above, itself below -- (2) or as the BOTTOM shard (3); 0 / False: off."""
        self.abi.check(self.lib.gh_comm_set_loopback(self.handle, int(mode)), "gh_comm_set_loopback")

    def set_halo_wait_ms(self, ms: int):
        """Bound of the halo-flag form's device-side waits on this communicator (-1: GH_HALO_WAIT_MS / 10 s)."""
        self.abi.check(self.lib.gh_comm_set_halo_wait_ms(self.handle, int(ms)), "gh_comm_set_halo_wait_ms")

    def step_form_in_use(self) -> int:
        """2 halo flag / 1 serial: what the next single-frame step takes where the frame allows the halo-flag form (the communicator
        falls back to 1 by itself when its edge bands report that the exchange does not run beside the launch)."""
        return int(self.lib.gh_comm_step_form_in_use(self.handle))

    def side_stream_replacements(self) -> int:
        """Side streams the halo-flag review has swapped for a fresh one (0 in a healthy process, at most 2)."""
        return int(self.lib.gh_comm_side_stream_replacements(self.handle))

    def exchange_alone_us(self) -> float:
        """One halo exchange with nothing beside it, microseconds (timed by gh_comm_init / gh_comm_set_loopback)."""
        return float(self.lib.gh_comm_exchange_alone_us(self.handle))

    def halo_wait_stats(self):
        """(waits that had to stand still, microseconds in all) reported by the edge bands of the halo-flag launches so far."""
        import ctypes as C
        n, us = C.c_uint(0), C.c_double(0.0)
        self.abi.check(self.lib.gh_comm_halo_wait_stats(self.handle, C.byref(n), C.byref(us)), "gh_comm_halo_wait_stats")
        return int(n.value), float(us.value)

    def halo_timeouts(self) -> int:
        return int(self.lib.gh_comm_halo_timeouts(self.handle))

    def check(self):
        """Call after the synchronisation that hands a sharded step's output to the host: raises GrayHipError when a halo-flag wait of
        the steps so far gave up (a neighbour's rows did not arrive within the bound) -- that output must not be used."""
        self.abi.check(self.lib.gh_comm_check(self.handle), "gh_comm_check")

    def set_step_form(self, form: int):
        """Form of the single-frame sharded steps: 2 halo flag, 1 serial (0 events: probe build only), -1 = environment, else the
        self-reviewing default (2 until the edge bands report that the exchange does not run beside the launch, then 1)."""
        self.abi.check(self.lib.gh_comm_set_step_form(self.handle, int(form)), "gh_comm_set_step_form")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gh_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def halo_exchange(self, up_peer, send_up, recv_up, down_peer, send_down, recv_down, nbytes, stream):
        p = lambda t: 0 if t is None else t.data_ptr()
        self.abi.check(self.lib.gh_halo_exchange_dev(self.handle, up_peer, p(send_up), p(recv_up), down_peer, p(send_down), p(recv_down),
                                                     nbytes, stream), "gh_halo_exchange_dev")

    def allreduce_sum_u64(self, t, stream):
        self.abi.check(self.lib.gh_comm_allreduce_sum_u64_dev(self.handle, t.data_ptr(), t.numel(), stream), "gh_comm_allreduce_sum_u64_dev")

    def allreduce_minmax_i32(self, t, stream):
        """t: device int32[2] = {min, max}; in place min / max over all ranks (gh_comm_allreduce_minmax_i32_dev)."""
        import torch
        assert t.dtype == torch.int32 and t.numel() == 2 and t.is_contiguous()
        self.abi.check(self.lib.gh_comm_allreduce_minmax_i32_dev(self.handle, t.data_ptr(), stream), "gh_comm_allreduce_minmax_i32_dev")

    def exchange_count(self) -> int:
        """Halo-row ncclGroups issued so far (one per sharded step, one per batch of prefetched frames)."""
        return int(self.lib.gh_comm_exchange_count(self.handle))

    def bind_conv_step(self, rgb_buf, out, rows, kernel, norm):
        import ctypes as C
        import torch
        w = rgb_buf.shape[1]
        assert tuple(rgb_buf.shape) == (rows + 2, w, 3) and tuple(out.shape) == (rows, w, 3) and rgb_buf.is_contiguous() and out.is_contiguous()
        k = (C.c_int32 * 9)(*[int(v) for v in kernel])
        fn, chk, h, src, dst, norm = self.lib.gh_sharded_conv3x3_step_dev, self.abi.check, self.handle, rgb_buf.data_ptr(), out.data_ptr(), int(norm)
        return lambda: chk(fn(h, src, dst, w, rows, k, norm, torch.cuda.current_stream().cuda_stream), "gh_sharded_conv3x3_step_dev")

    def bind_pipeline_gray_sobel(self, img_buf, gray_ws, edge_ws, rows, passes, mode):
        import torch
        w, c = img_buf.shape[1], img_buf.shape[2]
        assert img_buf.shape[0] == rows + 2 and tuple(gray_ws.shape) == (rows + 2, w) and tuple(edge_ws.shape) == (rows + 2, w)
        assert img_buf.is_contiguous() and gray_ws.is_contiguous() and edge_ws.is_contiguous()
        fn, chk, h = self.lib.gh_sharded_pipeline_gray_sobel_dev, self.abi.check, self.handle
        img, g, e = img_buf.data_ptr(), gray_ws.data_ptr(), edge_ws.data_ptr()
        return lambda: chk(fn(h, img, w, rows, c, passes, mode, g, e, torch.cuda.current_stream().cuda_stream), "gh_sharded_pipeline_gray_sobel_dev")

    def bind_fused_step(self, rgb_buf, edge, rows, mode):
        import torch
        w, c = rgb_buf.shape[1], rgb_buf.shape[2]
        assert rgb_buf.shape[0] == rows + 2 and tuple(edge.shape) == (rows, w) and rgb_buf.is_contiguous() and edge.is_contiguous()
        fn, chk, h, src, dst = self.lib.gh_sharded_fused_step_dev, self.abi.check, self.handle, rgb_buf.data_ptr(), edge.data_ptr()
        return lambda: chk(fn(h, src, dst, w, rows, c, mode, torch.cuda.current_stream().cuda_stream), "gh_sharded_fused_step_dev")

    def bind_fused_step_pipelined(self, bufs, edge, rows, mode, ahead=None):
        """bufs: the (rows+2, w, c) frame buffers the stream of frames cycles through.  The returned closure takes the
        step number k: it processes bufs[k % n] while the halo rows of bufs[(k+ahead) % n] are exchanged beside it
        (ahead = n - 1: as far ahead as the ring allows; 2 hides the exchange and the cross-queue event latency)."""
        import torch
        w, c = bufs[0].shape[1], bufs[0].shape[2]
        for b in bufs:
            assert tuple(b.shape) == (rows + 2, w, c) and b.is_contiguous()
        assert tuple(edge.shape) == (rows, w) and edge.is_contiguous()
        fn, chk, h, dst = self.lib.gh_sharded_fused_step_pipelined_dev, self.abi.check, self.handle, edge.data_ptr()
        chk(self.lib.gh_sharded_prefetch_reset(h), "gh_sharded_prefetch_reset")      # requests are remembered by address: start clean
        ptrs, n = [b.data_ptr() for b in bufs], len(bufs)
        a = max(n - 1, 0) if ahead is None else ahead
        return lambda k: chk(fn(h, ptrs[k % n], ptrs[(k + a) % n], dst, w, rows, c, mode, torch.cuda.current_stream().cuda_stream),
                             "gh_sharded_fused_step_pipelined_dev")

    def bind_fused_stream_step(self, bufs, edge, rows, mode):
        """Stream of frames in the halo-flag way (gh_sharded_fused_stream_step_dev): the returned closure takes the step number k,
        filters bufs[k % n] with one launch and starts the exchange of bufs[(k + 1) % n] beside it.  The rows of frame k + 1 must be in
        its buffer before step k is called."""
        import torch
        w, c = bufs[0].shape[1], bufs[0].shape[2]
        for b in bufs:
            assert tuple(b.shape) == (rows + 2, w, c) and b.is_contiguous()
        assert tuple(edge.shape) == (rows, w) and edge.is_contiguous()
        fn, chk, h, dst = self.lib.gh_sharded_fused_stream_step_dev, self.abi.check, self.handle, edge.data_ptr()
        chk(self.lib.gh_sharded_prefetch_reset(h), "gh_sharded_prefetch_reset")
        ptrs, n = [b.data_ptr() for b in bufs], len(bufs)
        return lambda k: chk(fn(h, ptrs[k % n], ptrs[(k + 1) % n] if n > 1 else None, dst, w, rows, c, mode, torch.cuda.current_stream().cuda_stream),
                             "gh_sharded_fused_stream_step_dev")

    def bind_sobel_stream_step(self, lumas, edge, rows):
        """The split pipeline's twin of bind_fused_stream_step: lumas are (rows+2, w) luma buffers; the luma of frame k + 1 must be in its
        buffer before step k is called."""
        import torch
        w = lumas[0].shape[1]
        for b in lumas:
            assert tuple(b.shape) == (rows + 2, w) and b.is_contiguous()
        assert tuple(edge.shape) == (rows, w) and edge.is_contiguous()
        fn, chk, h, dst = self.lib.gh_sharded_sobel_stream_step_dev, self.abi.check, self.handle, edge.data_ptr()
        chk(self.lib.gh_sharded_prefetch_reset(h), "gh_sharded_prefetch_reset")
        ptrs, n = [b.data_ptr() for b in lumas], len(lumas)
        return lambda k: chk(fn(h, ptrs[k % n], ptrs[(k + 1) % n] if n > 1 else None, dst, w, rows, torch.cuda.current_stream().cuda_stream),
                             "gh_sharded_sobel_stream_step_dev")

    def bind_fused_step_batched(self, bufs, edge, rows, mode, batch):
        """Stream of frames cycling through len(bufs) >= 2 * batch buffers: every `batch` steps the halo rows of the NEXT `batch`
        frames are exchanged in one group (gh_sharded_prefetch_halos_dev) while the current ones are processed; a step is one
        kernel launch.  The returned closure takes the step number k and processes bufs[k % n]."""
        import ctypes as C
        import torch
        w, c = bufs[0].shape[1], bufs[0].shape[2]
        n = len(bufs)
        assert batch >= 1 and n >= 2 * batch and n % batch == 0
        for b in bufs:
            assert tuple(b.shape) == (rows + 2, w, c) and b.is_contiguous()
        assert tuple(edge.shape) == (rows, w) and edge.is_contiguous()
        step, pre, chk, h, dst = self.lib.gh_sharded_fused_step_pipelined_dev, self.lib.gh_sharded_prefetch_halos_dev, self.abi.check, self.handle, edge.data_ptr()
        chk(self.lib.gh_sharded_prefetch_reset(h), "gh_sharded_prefetch_reset")      # requests are remembered by address: start clean
        ptrs = [b.data_ptr() for b in bufs]
        groups = [(C.c_void_p * batch)(*[ptrs[(g * batch + i) % n] for i in range(batch)]) for g in range(n // batch)]
        row_bytes = w * c

        requested = [-1]     # the group whose halos the last batch request covered

        def fn(k):
            s = torch.cuda.current_stream().cuda_stream
            if k % batch == 0:
                g = (k // batch) % len(groups)
                if requested[0] != g:                 # cold start, or the caller restarted its step counter: this batch's own halos first
                    chk(pre(h, groups[g], batch, row_bytes, rows, s), "gh_sharded_prefetch_halos_dev")
                nxt = (g + 1) % len(groups)           # the frames of the batch AFTER this one travel while this one is processed
                chk(pre(h, groups[nxt], batch, row_bytes, rows, s), "gh_sharded_prefetch_halos_dev")
                requested[0] = nxt
            chk(step(h, ptrs[k % n], None, dst, w, rows, c, mode, s), "gh_sharded_fused_step_pipelined_dev")
        return fn

    def bind_sobel_step_pipelined(self, lumas, edge, rows, ahead=None):
        import torch
        w = lumas[0].shape[1]
        for b in lumas:
            assert tuple(b.shape) == (rows + 2, w) and b.is_contiguous()
        assert tuple(edge.shape) == (rows, w) and edge.is_contiguous()
        fn, chk, h, dst = self.lib.gh_sharded_sobel_step_pipelined_dev, self.abi.check, self.handle, edge.data_ptr()
        chk(self.lib.gh_sharded_prefetch_reset(h), "gh_sharded_prefetch_reset")
        ptrs, n = [b.data_ptr() for b in lumas], len(lumas)
        a = max(n - 1, 0) if ahead is None else ahead
        return lambda k: chk(fn(h, ptrs[k % n], ptrs[(k + a) % n], dst, w, rows, torch.cuda.current_stream().cuda_stream),
                             "gh_sharded_sobel_step_pipelined_dev")

    def bind_sobel_step(self, luma_buf, edge, rows):
        import torch
        w = luma_buf.shape[1]
        assert luma_buf.shape[0] == rows + 2 and tuple(edge.shape) == (rows, w) and luma_buf.is_contiguous() and edge.is_contiguous()
        fn, chk, h, src, dst = self.lib.gh_sharded_sobel_step_dev, self.abi.check, self.handle, luma_buf.data_ptr(), edge.data_ptr()
        return lambda: chk(fn(h, src, dst, w, rows, torch.cuda.current_stream().cuda_stream), "gh_sharded_sobel_step_dev")


def _bind(backend, name, generic, *args):
    """Use the backend's pre-bound launcher when it has one (HipBackend), else wrap the generic call."""
    binder = getattr(backend, "bind_" + name, None)
    if binder is not None:
        return binder(*args)
    return lambda: generic(*args)


@dataclass
class ShardGeometry:
    rank: int
    world: int
    width: int
    y0: int
    y1: int

    @property
    def rows(self) -> int:
        return self.y1 - self.y0

    @property
    def top_is_border(self) -> bool:
        # the first rank WITH rows owns the frame's top edge: row_partition gives the remainder to the last ranks, so when the frame
        # has fewer rows than there are ranks the empty shards are the leading ones (and take part in no exchange)
        return self.y0 == 0

    @property
    def has_empty_shards(self) -> bool:
        """True on every rank alike iff some rank of the partition has no rows (frame height < world).  The C++ step functions
        address their neighbours as rank -/+ 1 and are not used then; the torch.distributed path skips the empty ranks."""
        return self.rows == 0 if self.rank == 0 else self.y0 < self.rank

    @property
    def bottom_is_border(self) -> bool:
        return self.rank == self.world - 1


class ShardedGraySobel:
    """One rank's share of `gray -> halo exchange -> Sobel` on a row-sharded frame.

    Buffers are allocated once: `luma` is (rows+2, width) with halo rows 0 and rows+1, `edge` is
    (rows, width).  step(rgb_shard) enqueues everything asynchronously and returns `edge`.
    """

    def __init__(self, geom: ShardGeometry, backend, device, group=None, overlap: bool = True, native_comm=None):
        import torch
        self.g = geom
        self.backend = backend
        self.group = group
        self.overlap = overlap
        rows = geom.rows
        self.luma = torch.zeros((rows + 2, geom.width), dtype=torch.uint8, device=device)
        self.edge = torch.empty((rows, geom.width), dtype=torch.uint8, device=device)
        self._ops = None
        self._gray_for = None          # (data_ptr of the last rgb shard, bound launcher)
        self._native = None            # exchange + Sobel launches issued by libgrayhip itself (NativeHaloComm)
        self._native_comm = None
        b, g = backend, geom
        if rows == 0:
            return
        if native_comm is not None and g.world > 1 and not g.has_empty_shards:
            assert (native_comm.rank, native_comm.world) == (g.rank, g.world)
            self._native = native_comm.bind_sobel_step(self.luma, self.edge, rows)
            self._native_comm = native_comm
            return
        if g.world == 1:
            self._whole = _bind(b, "sobel_rows", b.sobel_rows, self.luma, self.edge, rows, True, True)
        elif overlap and rows >= 3:
            # rows 1..rows-2 of the band depend on local luma only: a band whose "halo" rows are band rows 0 / rows-1
            self._inner = _bind(b, "sobel_rows", b.sobel_rows, self.luma[1:rows + 1], self.edge[1:rows - 1], rows - 2, False, False)
            self._top = _bind(b, "sobel_rows", b.sobel_rows, self.luma[0:3], self.edge[0:1], 1, g.top_is_border, False)
            self._bot = _bind(b, "sobel_rows", b.sobel_rows, self.luma[rows - 1:rows + 2], self.edge[rows - 1:rows], 1, False, g.bottom_is_border)
        else:
            self.overlap = False
            self._whole = _bind(b, "sobel_rows", b.sobel_rows, self.luma, self.edge, rows, g.top_is_border, g.bottom_is_border)

    # -- halo exchange: my first luma row goes up, my last goes down --------------------------
    def _exchange_ops(self):
        import torch.distributed as dist
        if self._ops is None:
            g, buf = self.g, self.luma
            ops = []
            if not g.top_is_border:
                ops.append(dist.P2POp(dist.isend, buf[1], g.rank - 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[0], g.rank - 1, self.group))
            if not g.bottom_is_border:
                ops.append(dist.P2POp(dist.isend, buf[g.rows], g.rank + 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[g.rows + 1], g.rank + 1, self.group))
            self._ops = ops
        return self._ops

    def result(self):
        """The edge plane of the last step() for the HOST: synchronises the current stream and -- native exchange -- checks that no
        halo-flag wait gave up (step() itself is asynchronous and cannot know)."""
        import torch
        torch.cuda.current_stream().synchronize()
        if self._native_comm is not None:
            self._native_comm.check()
        return self.edge

    def step(self, rgb_shard):
        import torch.distributed as dist
        g = self.g
        rows = g.rows
        if rows == 0:
            return self.edge
        if self._gray_for is None or self._gray_for[0] != rgb_shard.data_ptr():
            self._gray_for = (rgb_shard.data_ptr(),
                              _bind(self.backend, "gray_to_plane", self.backend.gray_to_plane, rgb_shard, self.luma[1:rows + 1]))
        self._gray_for[1]()
        if self._native is not None:
            self._native()
            return self.edge
        if g.world == 1:
            self._whole()
            return self.edge
        ops = self._exchange_ops()
        works = dist.batch_isend_irecv(ops) if ops else []
        if self.overlap:
            self._inner()
            for wk in works:
                wk.wait()
            self._top()
            self._bot()
        else:
            for wk in works:
                wk.wait()
            self._whole()
        return self.edge


class ShardedFusedGraySobel:
    """One rank's share of the FUSED `RGB -> luma -> Sobel` kernel on a row-sharded frame.

    The halo exchanged per frame is one RGB row per neighbour (width * channels bytes) instead of one luma row:
    the fused kernel recomputes the halo luma, so the luma plane never exists in HBM.  `rgb` is
    (rows+2, width, channels): rows 1..rows hold the shard (write new frames there), rows 0 / rows+1 receive the
    neighbours' edge rows.  step() enqueues exchange + kernels asynchronously and returns `edge`.
    """

    def __init__(self, geom: ShardGeometry, backend, device, channels: int = 3, group=None, overlap: bool = True, native_comm=None):
        import torch
        self.g = geom
        self.backend = backend
        self.group = group
        self.overlap = overlap
        rows = geom.rows
        self.rgb = torch.zeros((rows + 2, geom.width, channels), dtype=torch.uint8, device=device)
        self.edge = torch.empty((rows, geom.width), dtype=torch.uint8, device=device)
        self._ops = None
        self._native = None            # exchange + fused launches issued by libgrayhip itself (NativeHaloComm)
        self._native_comm = None
        b, g = backend, geom
        if rows == 0:
            return
        if native_comm is not None and g.world > 1 and not g.has_empty_shards:
            assert (native_comm.rank, native_comm.world) == (g.rank, g.world)
            self._native = native_comm.bind_fused_step(self.rgb, self.edge, rows, getattr(backend, "mode", 0))
            self._native_comm = native_comm
            return
        if g.world == 1:
            self._whole = _bind(b, "fused_rows", b.fused_rows, self.rgb, self.edge, rows, True, True)
        elif overlap and rows >= 3:
            self._inner = _bind(b, "fused_rows", b.fused_rows, self.rgb[1:rows + 1], self.edge[1:rows - 1], rows - 2, False, False)
            self._top = _bind(b, "fused_rows", b.fused_rows, self.rgb[0:3], self.edge[0:1], 1, g.top_is_border, False)
            self._bot = _bind(b, "fused_rows", b.fused_rows, self.rgb[rows - 1:rows + 2], self.edge[rows - 1:rows], 1, False, g.bottom_is_border)
        else:
            self.overlap = False
            self._whole = _bind(b, "fused_rows", b.fused_rows, self.rgb, self.edge, rows, g.top_is_border, g.bottom_is_border)

    @property
    def shard(self):
        return self.rgb[1:self.g.rows + 1]

    def result(self):
        """The edge plane of the last step() for the HOST: synchronises the current stream and -- native exchange -- checks that no
        halo-flag wait gave up (step() itself is asynchronous and cannot know)."""
        import torch
        torch.cuda.current_stream().synchronize()
        if self._native_comm is not None:
            self._native_comm.check()
        return self.edge

    def _exchange_ops(self):
        import torch.distributed as dist
        if self._ops is None:
            g, buf = self.g, self.rgb
            ops = []
            if not g.top_is_border:
                ops.append(dist.P2POp(dist.isend, buf[1], g.rank - 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[0], g.rank - 1, self.group))
            if not g.bottom_is_border:
                ops.append(dist.P2POp(dist.isend, buf[g.rows], g.rank + 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[g.rows + 1], g.rank + 1, self.group))
            self._ops = ops
        return self._ops

    def step(self):
        import torch.distributed as dist
        g = self.g
        if g.rows == 0:
            return self.edge
        if self._native is not None:
            self._native()
            return self.edge
        if g.world == 1:
            self._whole()
            return self.edge
        ops = self._exchange_ops()
        works = dist.batch_isend_irecv(ops) if ops else []
        if self.overlap:
            self._inner()
            for wk in works:
                wk.wait()
            self._top()
            self._bot()
        else:
            for wk in works:
                wk.wait()
            self._whole()
        return self.edge


def allreduce_channel_sums(local_sums, device, group=None):
    """Exact global per-channel sums from per-shard partial sums (3 x int64 all-reduce)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(local_sums, dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(v) for v in t.cpu().tolist()]


def allreduce_minmax(local_minmax, device, group=None, native_comm=None):
    """Global (min, max) of the mean-grey levels from per-shard (min, max): `reduction(min:) reduction(max:)` of
    old/parallel_to_grayscale.c:12 across ranks.  local_minmax: a device int32[2] tensor (HipBackend.mean_gray_minmax) or a pair
    of ints; a rank without rows passes the reference's initial values (256, -1), the neutral elements."""
    import torch
    import torch.distributed as dist
    t = local_minmax if isinstance(local_minmax, torch.Tensor) else torch.tensor([int(local_minmax[0]), int(local_minmax[1])], dtype=torch.int32, device=device)
    if native_comm is not None:
        native_comm.allreduce_minmax_i32(t, torch.cuda.current_stream().cuda_stream)
    elif dist.is_initialized() and dist.get_world_size(group) > 1:
        lo, hi = t[0:1].clone(), t[1:2].clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        t = torch.cat([lo, hi])
    lo, hi = t.cpu().tolist()
    return int(lo), int(hi)


class _RowHalo:
    """The exchange step shared by the sharded operators below: row 1 of `buf` goes to the rank above and its last row arrives in
    row 0; row `rows` goes down and the lower neighbour's first row arrives in row rows+1 (torch.distributed P2P ops, one batch)."""

    def __init__(self, geom: ShardGeometry, buf, group=None):
        self.g, self.buf, self.group, self._ops = geom, buf, group, None

    def __call__(self):
        import torch.distributed as dist
        g, buf = self.g, self.buf
        if g.world == 1 or g.rows == 0:
            return
        if self._ops is None:
            ops = []
            if not g.top_is_border:
                ops.append(dist.P2POp(dist.isend, buf[1], g.rank - 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[0], g.rank - 1, self.group))
            if not g.bottom_is_border:
                ops.append(dist.P2POp(dist.isend, buf[g.rows], g.rank + 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[g.rows + 1], g.rank + 1, self.group))
            self._ops = ops
        for wk in (dist.batch_isend_irecv(self._ops) if self._ops else []):
            wk.wait()


class ShardedConv3x3:
    """One rank's share of the 3x3 convolution (old/parallel_convolution.c:8-24) on a row-sharded RGB8 frame: `rgb` is
    (rows+2, width, 3), rows 1..rows = the shard; per step one RGB row is exchanged with each neighbour, then ONE launch
    covers the shard.  The frame's own top / bottom edge keeps the zero border.  step() returns `out` (rows, width, 3)."""

    def __init__(self, geom: ShardGeometry, backend, device, kernel, norm: int, group=None, native_comm=None):
        import torch
        self.g, self.backend = geom, backend
        self.kernel, self.norm = [int(v) for v in kernel], int(norm)
        rows = geom.rows
        self.rgb = torch.zeros((rows + 2, geom.width, 3), dtype=torch.uint8, device=device)
        self.out = torch.empty((rows, geom.width, 3), dtype=torch.uint8, device=device)
        self._native = None
        if rows and native_comm is not None and geom.world > 1 and not geom.has_empty_shards:
            assert (native_comm.rank, native_comm.world) == (geom.rank, geom.world)
            self._native = native_comm.bind_conv_step(self.rgb, self.out, rows, self.kernel, self.norm)
        self._halo = _RowHalo(geom, self.rgb, group)

    @property
    def shard(self):
        return self.rgb[1:self.g.rows + 1]

    def step(self):
        g = self.g
        if g.rows == 0:
            return self.out
        if self._native is not None:
            self._native()
            return self.out
        self._halo()
        self.backend.conv_rows(self.rgb, self.out, g.rows, self.kernel, self.norm, g.top_is_border, g.bottom_is_border)
        return self.out


class ShardedPipelineGraySobel:
    """One rank's share of the reference's multi-pass driver loop -- `passes` x (gray, extract, Sobel, replicate),
    monolithic/src/main_with_sobel.c:49-76 -- on a row-sharded frame, in place in `img` (rows+2, width, channels; rows
    1..rows = the shard).  Pass k+1 filters what pass k wrote, so there is ONE halo exchange per pass: an RGB row before pass 1
    (the fused kernel recomputes the halo luma), the neighbours' raw EDGE rows before every later pass -- the plane Sobel applies
    lum(e, e, e) on load (not the identity: 13 grey levels move in IEEE mode), one launch and 2 B/px per pass.  The edge planes of
    successive passes alternate between two (rows+2, width) work planes; run(passes) returns the last pass's edge plane (rows, width)."""

    def __init__(self, geom: ShardGeometry, backend, device, channels: int = 3, group=None, native_comm=None):
        import torch
        self.g, self.backend, self.channels = geom, backend, channels
        rows = geom.rows
        self.img = torch.zeros((rows + 2, geom.width, channels), dtype=torch.uint8, device=device)
        self.planes = [torch.zeros((rows + 2, geom.width), dtype=torch.uint8, device=device) for _ in range(2)]      # [0]: where the last pass lands
        self._native_comm = native_comm if (rows and native_comm is not None and geom.world > 1 and not geom.has_empty_shards) else None
        self._native = {}
        self._halo_img = _RowHalo(geom, self.img, group)
        self._halo_plane = [_RowHalo(geom, p, group) for p in self.planes]

    @property
    def shard(self):
        return self.img[1:self.g.rows + 1]

    @property
    def edge(self):
        """The last run()'s edge plane, (rows, width)."""
        return self.planes[0][1:self.g.rows + 1]

    def result(self):
        """The last run()'s edge plane for the HOST: synchronises, then checks the communicator (see ShardedFusedGraySobel.result)."""
        import torch
        torch.cuda.current_stream().synchronize()
        if self._native_comm is not None:
            self._native_comm.check()
        return self.edge

    def run(self, passes: int = 1):
        g, b, rows = self.g, self.backend, self.g.rows
        passes = max(int(passes), 1)              # main_with_sobel.c:44-45
        if rows == 0:                             # an empty (leading) shard takes part in no exchange: ShardGeometry.top_is_border
            return self.edge
        if self._native_comm is not None:
            fn = self._native.get(passes)
            if fn is None:
                fn = self._native[passes] = self._native_comm.bind_pipeline_gray_sobel(self.img, self.planes[1], self.planes[0], rows, passes, getattr(b, "mode", 0))
            fn()
            return self.edge
        for p in range(passes):
            o, i = (passes - 1 - p) & 1, (passes - p) & 1          # pass p writes planes[o]: the last pass lands in planes[0]
            out = self.planes[o][1:rows + 1]
            if p == 0 and self.channels >= 3:
                self._halo_img()
                b.fused_rows(self.img, out, rows, g.top_is_border, g.bottom_is_border)
                continue
            if p == 0:
                b.extract_plane(self.shard, self.planes[i][1:rows + 1])          # channels 1 / 2: the plane is channel 0
            self._halo_plane[i]()                                                 # a luma row (pass 1, 1 / 2 channels) or the neighbours' raw edge rows
            if p > 0 and self.channels >= 3:
                b.sobel_of_gray_rows(self.planes[i], out, rows, g.top_is_border, g.bottom_is_border)
            else:
                b.sobel_rows(self.planes[i], out, rows, g.top_is_border, g.bottom_is_border)
        b.replicate_plane(self.edge, self.shard)
        return self.edge
