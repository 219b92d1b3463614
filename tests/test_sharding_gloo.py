"""Row sharding + halo exchange (openmp-parallel-computing_amd/sharding.py) with world_size > 1 on
CPU: gloo backend, a checker backend built on the oracle.  N-shard result == unsharded result,
byte for byte.  (The GPU twin of this test, on one device, is test_gpu_parity.py::test_row_bands_*.)"""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401

sharding = importlib.import_module("openmp-parallel-computing_amd.sharding")


class OracleBackend:
    """CPU stand-in for HipBackend (tests only): same call surface, pixels from oracle/."""
    name = "oracle"

    def __init__(self, mode=0):
        from oracle import pyoracle
        self.O = pyoracle
        self.mode = mode

    def gray_to_plane(self, rgb, out):
        out.copy_(torch.from_numpy(self.O.gray_to_plane(rgb.numpy(), self.mode)))
        return out

    def sobel_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        b = buf.numpy()
        assert b.shape[0] == rows + 2
        ext = b.copy()
        if top_is_border:
            ext[0] = 0
        if bottom_is_border:
            ext[rows + 1] = 0
        e = self.O.sobel(ext)[1:rows + 1].copy()     # interior rows of the extended band
        if top_is_border:
            e[0] = 0
        if bottom_is_border:
            e[rows - 1] = 0
        out.copy_(torch.from_numpy(e))
        return out

    def fused_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        luma = torch.from_numpy(self.O.gray_to_plane(buf.numpy(), self.mode))
        return self.sobel_rows(luma, out, rows, top_is_border, bottom_is_border)

    def sobel_of_gray_rows(self, buf, out, rows, top_is_border, bottom_is_border):
        """a later pass of the multi-pass pipeline: the band holds EDGE rows, the filter sees lum(e, e, e)"""
        e = buf.numpy()
        luma = torch.from_numpy(self.O.gray_to_plane(np.repeat(e[:, :, None], 3, axis=2), self.mode))
        return self.sobel_rows(luma, out, rows, top_is_border, bottom_is_border)

    def sum_rgb(self, rgb):
        return self.O.sum_rgb(rgb.numpy())

    def conv_rows(self, buf, out, rows, kernel, norm, top_is_border, bottom_is_border):
        ext = buf.numpy().copy()
        if top_is_border:
            ext[0] = 0                                   # the zero border of the whole-frame operator
        if bottom_is_border:
            ext[rows + 1] = 0
        out.copy_(torch.from_numpy(self.O.conv3x3(ext, kernel, norm)[1:rows + 1].copy()))
        return out

    def gray_of_plane(self, plane, out):
        e = plane.numpy()
        out.copy_(torch.from_numpy(self.O.gray_to_plane(np.repeat(e[:, :, None], 3, axis=2), self.mode)))
        return out

    def extract_plane(self, img, out):
        out.copy_(img[:, :, 0])
        return out

    def replicate_plane(self, plane, img):
        img[:, :, :(3 if img.shape[2] >= 3 else 1)] = plane[:, :, None]     # channels 1 / 2: channel 0 only (deviations)
        return img

    def mean_gray_minmax(self, rgb, out):
        if rgb.numel() == 0:
            return torch.tensor([256, -1], dtype=torch.int32)
        g, lo, hi = self.O.mean_gray_minmax(rgb.numpy())
        out.copy_(torch.from_numpy(g))
        return torch.tensor([lo, hi], dtype=torch.int32)


def test_row_partition():
    assert sharding.row_partition(10, 1) == [(0, 10)]
    assert sharding.row_partition(10, 3) == [(0, 3), (3, 6), (6, 10)]          # remainder to the last ranks
    assert sharding.row_partition(16384, 8) == [(2048 * i, 2048 * (i + 1)) for i in range(8)]
    assert sharding.row_partition(2, 4) == [(0, 0), (0, 0), (0, 1), (1, 2)]
    for h, n in ((17, 5), (1, 8), (100, 7)):
        parts = sharding.row_partition(h, n)
        assert parts[0][0] == 0 and parts[-1][1] == h and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, overlap, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as O
        frame = O.frame_rand(w, h, 3, seed=seed)
        y0, y1 = sharding.row_partition(h, world)[rank]
        geom = sharding.ShardGeometry(rank, world, w, y0, y1)
        runner = sharding.ShardedGraySobel(geom, OracleBackend(0), torch.device("cpu"), overlap=overlap)
        shard = torch.from_numpy(np.ascontiguousarray(frame[y0:y1]))
        edge = runner.step(shard).numpy().copy()
        edge2 = runner.step(shard).numpy().copy()       # second step reuses the buffers
        fused = sharding.ShardedFusedGraySobel(geom, OracleBackend(0), torch.device("cpu"), overlap=overlap)
        fused.shard.copy_(shard)
        edge3 = fused.step().numpy().copy()
        sums = sharding.allreduce_channel_sums(OracleBackend(0).sum_rgb(shard) if y1 > y0 else [0, 0, 0], torch.device("cpu"))
        want = O.sobel(O.gray_to_plane(frame, 0))
        ok = (np.array_equal(edge, want[y0:y1]) and np.array_equal(edge2, edge) and np.array_equal(edge3, edge)
              and sums == O.sum_rgb(frame))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,overlap", [(2, 64, 20, True), (2, 48, 7, False), (3, 32, 11, True), (4, 16, 9, True), (2, 80, 5, True)])
def test_sharded_equals_unsharded_gloo(world, w, h, overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, overlap, 100 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert results == {r: True for r in range(world)}


# ---- the remaining SURVEY 8(e) paths: convolution on a shard, the multi-pass pipeline, min / max ------------------------------

GAUSS, BOX, SIGNED = [1, 2, 1, 2, 4, 2, 1, 2, 1], [1] * 9, [-1, -2, -1, 0, 0, 0, 1, 2, 1]


def _worker_more(rank, world, port, w, h, channels, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as O
        cpu = torch.device("cpu")
        frame = O.frame_rand(w, h, channels, seed=seed)
        y0, y1 = sharding.row_partition(h, world)[rank]
        geom = sharding.ShardGeometry(rank, world, w, y0, y1)
        be = OracleBackend(0)
        shard = torch.from_numpy(np.ascontiguousarray(frame[y0:y1]))
        ok = True
        # (1) 3x3 convolution: sharded == unsharded, zero border only at the frame's own edges (old/parallel_convolution.c:8-24)
        if channels == 3:
            for kern, norm in ((GAUSS, 16), (BOX, 9), (SIGNED, 1)):
                conv = sharding.ShardedConv3x3(geom, be, cpu, kern, norm)
                conv.shard.copy_(shard)
                got = conv.step().numpy()
                ok = ok and np.array_equal(got, O.conv3x3(frame, kern, norm)[y0:y1])
        # (2) passes x (gray, extract, Sobel, replicate): one exchange per pass (main_with_sobel.c:49-76)
        for passes in (1, 2, 3):
            pipe = sharding.ShardedPipelineGraySobel(geom, be, cpu, channels=channels)
            pipe.shard.copy_(shard)
            pipe.run(passes)
            want_img, _, want_edge = O.pipeline(frame.copy(), passes, 0)
            ok = ok and np.array_equal(pipe.shard.numpy(), want_img[y0:y1]) and np.array_equal(pipe.edge.numpy(), want_edge[y0:y1])
        # (3) mean-grey min / max over the ranks (old/parallel_to_grayscale.c:12,36-37); empty shards contribute 256 / -1
        if channels == 3:
            mm = be.mean_gray_minmax(shard, torch.empty_like(shard))
            lo, hi = sharding.allreduce_minmax(mm, cpu)
            _, wlo, whi = O.mean_gray_minmax(frame)
            ok = ok and (lo, hi) == (wlo, whi)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,channels", [(2, 40, 12, 3), (3, 24, 10, 3), (2, 33, 7, 4), (3, 16, 3, 3), (4, 20, 3, 3), (2, 18, 9, 1)])
def test_conv_multipass_minmax_sharded_equals_unsharded_gloo(world, w, h, channels):
    """(4, 20, 3): a frame with fewer rows than ranks -- the leading rank has no rows and takes part in no exchange."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_more, args=(r, world, port, w, h, channels, 200 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_empty_leading_shards_are_agreed_on_by_every_rank():
    for h, world in ((3, 4), (1, 8), (8, 8), (9, 8), (100, 7), (2, 4)):
        parts = sharding.row_partition(h, world)
        flags = [sharding.ShardGeometry(r, world, 16, *parts[r]).has_empty_shards for r in range(world)]
        assert flags == [h < world] * world
        tops = [r for r in range(world) if parts[r][1] > parts[r][0] and sharding.ShardGeometry(r, world, 16, *parts[r]).top_is_border]
        assert len(tops) == 1 and parts[tops[0]][0] == 0
