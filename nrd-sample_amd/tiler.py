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
        self.split_dispatches = 0  # dispatches run as boundary strips + interior with the exchange in flight
        self._plan_cache = {}
        self._p2p_cache = {}  # dispatch todo -> (P2POp list, bytes sent): pool planes never move, so the row slices are built once
        self._deferred = []   # exchanges of planes only the NEXT frame reads: in flight until the next dispatch list starts

    def _as_tensor(self, buf):
        import torch

        return buf if hasattr(buf, "data_ptr") else torch.from_numpy(buf)

    def _plane_of(self, code):
        pool, index = code >> 16, code & 0xFFFF
        if pool > 1:
            return None
        return self.band.nrd.pools[pool][index]

    def _plan(self, ids, dispatches):
        """for every dispatch: (now, later), two lists of (plane code, rows) over the pool planes it writes.
        now: boundary rows a LATER DISPATCH OF THIS FRAME reads - the largest halo of the dispatches that read the plane before
        it is written again; these are exchanged strips-first and waited for before the next dispatch.
        later: permanent planes that survive the frame get the full halo (next frame's reprojection reads them at
        motion-displaced rows), but nobody needs those rows before the next frame: their exchange is enqueued after the
        dispatch and stays in flight behind the rest of the frame (no strips, no wait) - most dispatches end with such a plane
        (history, fast history, stabilized luma, accumulation speeds), and an 80-row strip is two mostly empty workgroup rounds"""
        key = tuple((d["name"], tuple(d["written"]), tuple(d["read"]), d["halo_rows"]) for d in dispatches)
        if key in self._plan_cache:
            return self._plan_cache[key]
        plan = []
        for i, d in enumerate(dispatches):
            now, later = [], []
            for code in d["written"]:
                if (code >> 16) > 1:
                    continue  # output slots are final
                rows, rewritten = 0, False
                for r in dispatches[i + 1:]:
                    if code in r["read"]:
                        rows = max(rows, r["halo_rows"])
                    if code in r["written"]:
                        rewritten = True
                        break
                rows = min(rows, self.band.halo)
                if (code >> 16) == 0 and not rewritten and rows < self.band.halo:
                    later.append((code, self.band.halo, rows))  # the `rows` nearest the band edge travel at once (below)
                if rows > 0:
                    now.append((code, rows))
            plan.append((now, later))
        self._plan_cache[key] = plan
        return plan

    def _ops(self, bufs_rows):
        """send / recv descriptors for [(2-D byte tensor [local rows, pitch], rows-per-texel-row divisor, rows[, skip])]:
        the `rows` owned rows next to each band edge go to that neighbour's halo; `skip` = how many of them (nearest the edge)
        an earlier exchange already delivered"""
        dist, b = self.dist, self.band
        L = b.layout
        ops = []
        for item in bufs_rows:
            t, div, rows = item[:3]
            hskip = (item[3] // div) if len(item) > 3 else 0
            hrows = max((rows + div - 1) // div, 1)
            first, n = L["own_first"] // div, max(L["own_rows"] // div, 1)
            total = t.shape[0]
            if b.rank > 0:  # upper neighbour: send my first owned rows, receive into my top halo
                send = t[first + hskip:first + min(hrows, n)]
                top = first - min(hrows, first)
                recv = t[top:first - hskip]
                if send.shape[0] > 0:
                    ops.append((dist.isend, send, b.rank - 1))
                if recv.shape[0] > 0:
                    ops.append((dist.irecv, recv, b.rank - 1))
            if b.rank < b.world - 1:  # lower neighbour
                send = t[first + n - min(hrows, n):first + n - hskip]
                bot = min(first + n + hrows, total)
                recv = t[first + n + hskip:bot]
                if send.shape[0] > 0:
                    ops.append((dist.isend, send, b.rank + 1))
                if recv.shape[0] > 0:
                    ops.append((dist.irecv, recv, b.rank + 1))
        return ops

    def exchange_start(self, bufs_rows):
        """enqueue the exchange (RCCL: on the communicator's stream, ordered after the work already on the current stream);
        returns the handles to wait on"""
        ops = self._ops(bufs_rows)
        if not ops:
            return []
        dist = self.dist
        for fn, ten, _ in ops:
            if fn is dist.isend:
                self.bytes_exchanged += ten.numel() * ten.element_size()
        if ops[0][1].is_cuda and dist.get_backend(self.group) != "nccl":
            # only RCCL orders itself after the work already enqueued on the current stream; any other backend (gloo in the
            # tests) must see finished rows
            import torch
            torch.cuda.current_stream().synchronize()
        return dist.batch_isend_irecv([dist.P2POp(fn, ten, peer, self.group) for fn, ten, peer in ops])

    def exchange(self, bufs_rows):
        """blocking exchange of the full halo of [(tensor, divisor)] planes (external inputs)"""
        for w in self.exchange_start([(t, div, self.band.halo) for t, div in bufs_rows]):
            w.wait()

    def exchange_inputs(self, planes):
        """external inputs arrive per band; refresh their halos once (a renderer would have produced them per band)"""
        items = []
        for key in sorted(planes):  # same order on every rank (send/recv pairs match by order)
            if key == "confidence":
                continue
            items.append((self._as_tensor(planes[key]), 1))
        self.exchange(items)

    def exchange_start_cached(self, todo):
        """exchange_start(items_of(todo)) with the P2POp list of a dispatch built once: the pool planes live as long as the
        instance, so per frame only batch_isend_irecv itself runs on the host (seven exchanges per frame sit on the critical
        path of the launch thread)"""
        key = tuple(todo)
        hit = self._p2p_cache.get(key)
        if hit is None:
            dist = self.dist
            ops = self._ops(self.items_of(todo))
            sent = sum(ten.numel() * ten.element_size() for fn, ten, _ in ops if fn is dist.isend)
            hit = ([dist.P2POp(fn, ten, peer, self.group) for fn, ten, peer in ops], sent,
                   bool(ops) and ops[0][1].is_cuda and dist.get_backend(self.group) != "nccl")
            self._p2p_cache[key] = hit
        p2p, sent, host_sync = hit
        if not p2p:
            return []
        self.bytes_exchanged += sent
        if host_sync:  # see exchange_start
            import torch
            torch.cuda.current_stream().synchronize()
        return self.dist.batch_isend_irecv(p2p)

    def items_of(self, todo):
        local_h = self.band.layout["local_h"]
        items = []
        for code, rows, *skip in todo:
            p = self._plane_of(code)
            div = max(int(round(local_h / p["height"])), 1)
            items.append((self._as_tensor(p["buf"]), div, rows, *skip))
        return items

    def finish(self):
        """wait for the exchanges still in flight (rows only the next frame reads)"""
        for w in self._deferred:
            w.wait()
        self._deferred = []

    def run_dispatch(self, ids, i, plan_entry):
        """one dispatch + the halo exchange of what it wrote. When the band is tall enough the rows a neighbour needs are
        computed FIRST (boundary strips), their exchange is started, and the interior is computed while the rows travel -
        the copies overlap the compute instead of serialising with it (nrdhip_denoise_rows). Rows that only the next frame
        reads follow without strips and without a wait (see _plan)."""
        if i == 0:
            self.finish()  # a new dispatch list: last frame's permanent planes must have arrived
        todo, later = plan_entry
        self._run_dispatch_now(ids, i, todo)
        if later:
            self._deferred += self.exchange_start_cached(later)

    def _run_dispatch_now(self, ids, i, todo):
        nrd, L, b = self.band.nrd, self.band.layout, self.band
        own0, own_n = L["own_first"], L["own_rows"]
        strip = (max([t[1] for t in todo] + [0]) + 15) // 16 * 16
        up, down = b.rank > 0, b.rank < b.world - 1
        if not todo or not (up or down) or own_n < 4 * strip or own0 % 16:
            nrd.denoise_range(ids, i, 1)
            for w in self.exchange_start_cached(todo):
                w.wait()
            return
        self.split_dispatches += 1
        lo = own0 + (strip if up else 0)
        hi = own0 + own_n - (strip if down else 0)
        first = True
        if up:
            nrd.denoise_rows(ids, i, own0, strip, part=1)
            first = False
        if down:
            nrd.denoise_rows(ids, i, hi, own0 + own_n - hi, part=1 if first else 0)
        works = self.exchange_start_cached(todo)
        nrd.denoise_rows(ids, i, lo, hi - lo, part=2)
        for w in works:
            w.wait()

    def denoise(self, identifiers):
        ids = [int(i) for i in identifiers]
        nrd = self.band.nrd
        dispatches = nrd.dispatches(ids)
        plan = self._plan(ids, dispatches)
        for i, entry in enumerate(plan):
            self.run_dispatch(ids, i, entry)


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
        # timed variant: same stepping, HIP events around each dispatch (its strips + interior; the wait for the rows in
        # flight falls between the events of consecutive dispatches)
        ids = self.ids
        dispatches = band.nrd.dispatches(ids)
        if self.names is None:
            self.names = [(x["name"], x["bytes_per_pixel"]) for x in dispatches]
        plan = self.tiler._plan(ids, dispatches)
        evs = []
        for i, entry in enumerate(plan):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.tiler.run_dispatch(ids, i, entry)
            b.record()
            evs.append((a, b))
        self.events.append(evs)

    def finish(self):
        self.tiler.finish()

    def pass_times_ms(self):
        import torch

        torch.cuda.synchronize()
        acc = {}
        for evs in self.events:
            for (name, bpp), (a, b) in zip(self.names, evs):
                t, n, _ = acc.get(name, (0.0, 0, bpp))
                acc[name] = (t + a.elapsed_time(b), n + 1, bpp)
        return {k: (t / n, bpp) for k, (t, n, bpp) in acc.items()}
