"""Row tiling of one frame across the GPUs of a node (scheme A "per-pass halo").

One process per GPU. Rank r owns a contiguous band of rows of the global frame and keeps `halo` extra rows of every plane on
each side. After every recorded dispatch the rows a neighbour may read are exchanged with the <= 2 row neighbours by
point-to-point send/recv through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests,
where the oracle stands in for the kernels). Neighbour pairs use their own xGMI link, there is no collective and no ring.
The result is bit-identical to a single-GPU run as long as no pass reads farther than `halo` rows beyond its band
(filter radii + motion); `required_halo()` derives that bound from the dispatch list.

The reference has no counterpart (single adapter, single queue: Source/NRDSample.cpp:755-778); the per-pass stepping uses the
GetComputeDispatches-style part of the C-ABI (include/nrdhip.h nrdhip_dispatch_info_get / nrdhip_denoise_range).
"""
import numpy as np

from . import api
from .harness import INPUT_SLOTS, OUTPUT_SLOTS, Harness

DEFAULT_HALO = 80  # rows; multiple of 16 so band tile grids coincide with the single-GPU tile grid


def band_layout(frame_h, world, rank, halo):
    """rows owned by `rank` and the local window [row0, row0 + local_h) it stores"""
    base = (frame_h // world) // 16 * 16 if world > 1 else frame_h
    own0 = rank * base
    own1 = frame_h if rank == world - 1 else own0 + base
    row0 = max(own0 - halo, 0)
    row1 = min(own1 + halo, frame_h)
    return dict(own0=own0, own1=own1, row0=row0, local_h=row1 - row0, own_first=own0 - row0, own_rows=own1 - own0)


def required_halo(dispatches, motion_rows=8):
    h = max([d["halo_rows"] for d in dispatches] + [0]) + motion_rows
    return (h + 15) // 16 * 16


class BandHarness(Harness):
    """Harness for one row band: planes are local_h rows tall, CommonSettings describe the whole frame."""

    def __init__(self, backend, denoisers, width, frame_h, rank, world, halo=DEFAULT_HALO):
        self.layout = band_layout(frame_h, world, rank, halo)
        self.frame_h, self.rank, self.world, self.halo = frame_h, rank, world, halo
        L = self.layout
        super().__init__(backend, denoisers, width, L["local_h"], frame_height=frame_h, band_row0=L["row0"],
                         band_own_first=L["own_first"], band_own_rows=L["own_rows"])

    def local_rows(self, global_plane):
        L = self.layout
        return global_plane[L["row0"]:L["row0"] + L["local_h"]]

    def own_rows(self, local_plane):
        L = self.layout
        return local_plane[L["own_first"]:L["own_first"] + L["own_rows"]]


class Tiler:
    """Steps a BandHarness dispatch by dispatch, exchanging halo rows after each one."""

    def __init__(self, band, dist, group=None):
        self.band, self.dist, self.group = band, dist, group
        self.bytes_exchanged = 0
        self._plan_cache = {}

    def _as_tensor(self, buf):
        import torch

        return buf if hasattr(buf, "data_ptr") else torch.from_numpy(buf)

    def _plane_of(self, code):
        pool, index = code >> 16, code & 0xFFFF
        if pool > 1:
            return None
        return self.band.nrd.pools[pool][index]

    def _plan(self, ids, dispatches):
        """for every dispatch: the pool planes whose halo rows must be refreshed before a later reader runs"""
        key = tuple((d["name"], tuple(d["written"])) for d in dispatches)
        if key in self._plan_cache:
            return self._plan_cache[key]
        plan = []
        for i, d in enumerate(dispatches):
            todo = []
            for code in d["written"]:
                if (code >> 16) > 1:
                    continue  # output slots are final
                permanent = (code >> 16) == 0
                later_stencil = any(code in r["read"] and r["halo_rows"] > 0 for r in dispatches[i + 1:])
                if permanent or later_stencil:
                    todo.append(code)
            plan.append(todo)
        self._plan_cache[key] = plan
        return plan

    def exchange(self, bufs_rows):
        """bufs_rows: list of (2-D byte tensor [local rows, pitch], rows-per-texel-row divisor). Exchanges the owned boundary
        rows with rank-1 / rank+1 into their halos."""
        dist, b = self.dist, self.band
        L, halo = b.layout, b.halo
        ops, keep = [], []
        for t, div in bufs_rows:
            hrows = max(halo // div, 1)
            first, n = L["own_first"] // div, max(L["own_rows"] // div, 1)
            total = t.shape[0]
            if b.rank > 0:  # upper neighbour: send my first owned rows, receive into my top halo
                send = t[first:first + min(hrows, n)]
                top = first - min(hrows, first)
                recv = t[top:first]
                ops.append((dist.isend, send, b.rank - 1))
                if recv.shape[0] > 0:
                    ops.append((dist.irecv, recv, b.rank - 1))
            if b.rank < b.world - 1:  # lower neighbour
                send = t[first + n - min(hrows, n):first + n]
                bot = min(first + n + hrows, total)
                recv = t[first + n:bot]
                ops.append((dist.isend, send, b.rank + 1))
                if recv.shape[0] > 0:
                    ops.append((dist.irecv, recv, b.rank + 1))
        if not ops:
            return
        p2p = [dist.P2POp(fn, ten, peer, self.group) for fn, ten, peer in ops]
        for fn, ten, _ in ops:
            if fn is dist.isend:
                self.bytes_exchanged += ten.numel() * ten.element_size()
        for w in dist.batch_isend_irecv(p2p):
            w.wait()

    def exchange_inputs(self, planes):
        """external inputs arrive per band; refresh their halos once (a renderer would have produced them per band)"""
        items = []
        for key in sorted(planes):  # same order on every rank (send/recv pairs match by order)
            if key == "confidence":
                continue
            items.append((self._as_tensor(planes[key]), 1))
        self.exchange(items)

    def denoise(self, identifiers):
        ids = [int(i) for i in identifiers]
        nrd = self.band.nrd
        dispatches = nrd.dispatches(ids)
        plan = self._plan(ids, dispatches)
        local_h = self.band.layout["local_h"]
        for i, todo in enumerate(plan):
            nrd.denoise_range(ids, i, 1)
            items = []
            for code in todo:
                p = self._plane_of(code)
                div = max(int(round(local_h / p["height"])), 1)
                items.append((self._as_tensor(p["buf"]), div))
            if items:
                self.exchange(items)


class TiledRunner:
    """bench.py's N > 1 path: every rank renders its band of the synthetic scene on its GPU, then steps the tiler."""

    def __init__(self, pkg, backend, device, dens, width, frame_h, rank, world, unique, dolly, settings_of):
        import torch
        import torch.distributed as dist

        from .harness import pingpong  # same camera path as the single-GPU runner

        self.pingpong = pingpong
        self.api, self.dens, self.unique = pkg.api, dens, unique
        self.band = BandHarness(backend, dens, width, frame_h, rank, world)
        self.tiler = Tiler(self.band, dist)
        L = self.band.layout
        self.scene = pkg.synth.Scene(width, L["local_h"], dolly=dolly, device=device, frame_height=frame_h, row0=L["row0"])
        self.settings = settings_of(pkg.api, self.scene, dens)
        self.frames = []
        for i in range(unique):
            fwd = self.scene.frame(i, prev_index=max(i - 1, 0))
            bwd = self.scene.frame(i, prev_index=min(i + 1, unique - 1))
            planes = self.band.upload(fwd)
            mv_b = self.band.upload({"mv": bwd["mv"]})["mv"]
            self.tiler.exchange_inputs(planes)
            self.tiler.exchange_inputs({"mv": mv_b})
            self.frames.append(dict(planes=planes, mv_f=planes["mv"], mv_b=mv_b, fwd=fwd, bwd=bwd))
        torch.cuda.synchronize()
        self.ids = [int(d) for d in dens]
        self.events_on = False
        self.events, self.names = [], None

    def enable_events(self, on):
        self.events_on = on

    def step(self, f, reset):
        import torch

        band, api = self.band, self.api
        cur = self.pingpong(self.unique, f)
        prev = self.pingpong(self.unique, f - 1) if f > 0 else min(1, self.unique - 1)
        fr = self.frames[cur]
        backward = prev > cur
        planes = dict(fr["planes"])
        planes["mv"] = fr["mv_b"] if backward else fr["mv_f"]
        cs = self.scene.common_settings(api, fr["bwd"] if backward else fr["fwd"], f, reset=reset)
        band.nrd.new_frame()
        band.nrd.set_common_settings(cs)
        band.bind(planes)
        for d in self.dens:
            band.nrd.set_denoiser_settings(int(d), self.settings[d])
        if not self.events_on:
            self.tiler.denoise(self.ids)
            return
        # timed variant: same stepping, HIP events around each kernel (exchange excluded from the per-kernel time)
        ids = self.ids
        dispatches = band.nrd.dispatches(ids)
        if self.names is None:
            self.names = [(x["name"], x["bytes_per_pixel"]) for x in dispatches]
        plan = self.tiler._plan(ids, dispatches)
        local_h = band.layout["local_h"]
        evs = []
        for i, todo in enumerate(plan):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            band.nrd.denoise_range(ids, i, 1)
            b.record()
            evs.append((a, b))
            items = [(self.tiler._as_tensor(self.tiler._plane_of(c)["buf"]), max(int(round(local_h / self.tiler._plane_of(c)["height"])), 1)) for c in todo]
            if items:
                self.tiler.exchange(items)
        self.events.append(evs)

    def pass_times_ms(self):
        import torch

        torch.cuda.synchronize()
        acc = {}
        for evs in self.events:
            for (name, bpp), (a, b) in zip(self.names, evs):
                t, n, _ = acc.get(name, (0.0, 0, bpp))
                acc[name] = (t + a.elapsed_time(b), n + 1, bpp)
        return {k: (t / n, bpp) for k, (t, n, bpp) in acc.items()}
