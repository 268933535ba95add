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

DEFAULT_HALO = 80  # rows; multiple of 16 so band tile grids coincide with the single-GPU tile grid (default settings need 80)
DEFAULT_MOTION_ROWS = 8  # vertical motion (rows) reprojection may follow beyond a pass's own reach


class HaloError(ValueError):
    """a pass reads farther than the rows the band stores: the tiled result would differ from the single-GPU frame"""


SKY_TILE_COST = 0.15  # cost of a 16x16 tile without geometry relative to a filtered one (4K bench scene: 28 % sky costs the frame ~19 %)


def band_bounds(frame_h, world, tile_row_cost=None, min_rows=16):
    """First owned row of every rank, plus frame_h (world + 1 entries). Bands are whole TILE ROWS (16 pixel rows; the band tile grids
    then coincide with the single-GPU tile grid) and every rank gets its share of them, not a remainder: 270 tile rows over 8 ranks
    are 34, 34, 34, 34, 34, 34, 33, 33 (== nrd::TiledIntegration::BandBounds).
    ``tile_row_cost``: one non-negative number per tile row (e.g. tiles with geometry + SKY_TILE_COST x tiles without, from last
    frame's Tiles mask or the depth buffer): the boundaries then balance the COST of the bands instead of their height - a frame
    whose upper third is sky would otherwise hand the upper ranks almost nothing to do. Every band keeps at least ``min_rows`` rows
    (a neighbour's halo must come from ONE band: pass the halo). Boundaries are fixed for the life of the band instances (plane
    sizes depend on them): re-balancing means re-creating the bands, i.e. an accumulation restart - nothing moves them mid-stream."""
    n = (frame_h + 15) // 16
    if world <= 1:
        return [0, frame_h]
    min_t = max((min_rows + 15) // 16, 1)
    if n < world * min_t:
        raise ValueError("%d rows cannot be split into %d bands of at least %d rows" % (frame_h, world, min_rows))
    if tile_row_cost is None:
        q, rem = divmod(n, world)
        counts = [q + (1 if r < rem else 0) for r in range(world)]
        cuts = [0]
        for k in counts:
            cuts.append(cuts[-1] + k)
    else:
        cost = [max(float(c), 0.0) for c in tile_row_cost]
        if len(cost) != n:
            raise ValueError("tile_row_cost needs one entry per tile row (%d), got %d" % (n, len(cost)))
        total = sum(cost)
        if total <= 0.0:
            return band_bounds(frame_h, world, None, min_rows)
        cum = [0.0]
        for c in cost:
            cum.append(cum[-1] + c)
        cuts = [0]
        for k in range(1, world):
            lo, hi = cuts[-1] + min_t, n - (world - k) * min_t  # leave room for the bands above and below
            target = total * k / world
            i = lo
            while i < hi and cum[i] < target:
                i += 1
            if i > lo and (target - cum[i - 1]) < (cum[i] - target):  # the nearer of the two candidate cuts
                i -= 1
            cuts.append(i)
        cuts.append(n)
    return [min(c * 16, frame_h) for c in cuts]


def band_layout(frame_h, world, rank, halo, bounds=None):
    """rows owned by `rank` and the local window [row0, row0 + local_h) it stores; ``bounds`` from band_bounds() (default: even)"""
    bounds = bounds if bounds is not None else band_bounds(frame_h, world)
    own0, own1 = bounds[rank], bounds[rank + 1]
    row0 = max(own0 - halo, 0)
    row1 = min(own1 + halo, frame_h)
    return dict(own0=own0, own1=own1, row0=row0, local_h=row1 - row0, own_first=own0 - row0, own_rows=own1 - own0)


def tile_row_cost_from_depth(viewz, denoising_range, sky_cost=SKY_TILE_COST):
    """band_bounds() cost profile from a depth buffer (any resolution that is a whole fraction of the frame's: one texel row of a
    1/16-resolution buffer = one tile row): geometry tiles count 1, sky tiles ``sky_cost``"""
    import numpy as np

    z = np.abs(np.asarray(viewz, dtype=np.float64))
    geo = (z <= denoising_range)
    return [float(row.sum() + sky_cost * (row.size - row.sum())) for row in geo]


def required_halo(dispatches, motion_rows=DEFAULT_MOTION_ROWS):
    """rows a band must store beyond its owned rows: the largest read reach of the dispatch list + the vertical motion the
    temporal passes may follow, rounded up to whole tiles (== nrdhip_required_halo of the C-ABI)"""
    h = max([d["halo_rows"] for d in dispatches] + [0]) + motion_rows
    return (h + 15) // 16 * 16


def probe_halo(backend, denoisers, settings=None, motion_rows=DEFAULT_MOTION_ROWS):
    """required_halo() for a denoiser list + settings BEFORE the band instances exist (their size depends on it): the reach of
    a pass depends on settings only (blur radii, pre-pass radii, A-trous iteration count), so a 64x64 throw-away instance
    answers. ``settings``: {Denoiser: settings struct} (defaults for the rest)."""
    nrd = api.Integration(backend)
    r = nrd.recreate([(int(d), d) for d in denoisers], 64, 64)
    if r != api.Result.SUCCESS:
        raise api.NrdError("Recreate(probe)", int(r))
    try:
        cs = api.CommonSettings()
        cs.rectSize[0] = cs.rectSize[1] = cs.resourceSize[0] = cs.resourceSize[1] = 64
        cs.rectSizePrev[0] = cs.rectSizePrev[1] = cs.resourceSizePrev[0] = cs.resourceSizePrev[1] = 64
        cs.viewToClipMatrix[0] = cs.viewToClipMatrix[5] = cs.viewToClipMatrix[11] = 1.0
        cs.viewToClipMatrixPrev[0] = cs.viewToClipMatrixPrev[5] = cs.viewToClipMatrixPrev[11] = 1.0
        for i in (0, 5, 10, 15):
            cs.worldToViewMatrix[i] = cs.worldToViewMatrixPrev[i] = 1.0
        nrd.set_common_settings(cs)
        for d, st in (settings or {}).items():
            if d in denoisers:
                nrd.set_denoiser_settings(int(d), st)
        return required_halo(nrd.dispatches([int(d) for d in denoisers]), motion_rows)
    finally:
        nrd.destroy()


def plan_bytes(plan, planes, width, neighbours=1):
    """bytes ONE band sends to ONE neighbour per dispatch of a plan (Tiler._plan): [(strips-first bytes, strip rows, deferred bytes)].
    `planes`: code -> bytes per texel of the pool plane (tap-texel planes with a guide prefix send their 8-byte signal half only)."""
    out = []
    for now, later in plan:
        nb = sum((8 if len(e) > 3 else planes[e[0]]) * (e[1] - (e[2] if len(e) > 2 else 0)) * width for e in now)
        lb = sum((8 if len(e) > 3 else planes[e[0]]) * (e[1] - e[2]) * width for e in later)
        strip = (max([e[1] for e in now] + [0]) + 15) // 16 * 16
        out.append((nb * neighbours, strip, lb * neighbours))
    return out


def exchange_overlap_model(names, pass_ms, bytes_per_dispatch, own_rows, neighbours, link_gbs=50.0, latency_us=20.0):
    """What a row-tiled frame should cost on real links, dispatch by dispatch, with the strips-first schedule of both tilers
    (csrc/nrdhip_tiler.cpp nrdhip_tiler_denoise, Tiler._run_dispatch_now) - the number a first SCALE_rNN.json run is to be compared with.
      pass_ms[i]            GPU time of dispatch i on the band (strips + interior; measured on one GPU)
      bytes_per_dispatch[i] (bytes sent strips-first to ONE neighbour, strip rows, bytes sent deferred to ONE neighbour)  [plan_bytes]
      neighbours            1 for the first / last band, 2 for an interior band (each neighbour has its own xGMI link: transfers to the two
                            run side by side, so the exchange TIME does not grow with it - only the share of the band that is strips does)
    Per dispatch with a strips-first exchange: the strips (`neighbours` x strip rows of `own_rows`) run first, then the rows travel
    (latency + bytes / link rate per direction) WHILE the interior runs; the next dispatch waits for both, so the dispatch costs
    strips + max(interior, exchange) and `unhidden` = max(0, exchange - interior). Deferred rows (read by the next frame only) queue on
    the same side stream behind the frame's exchanges; they cost nothing unless their total exceeds what is left of the frame."""
    rows, total, unhidden_total, deferred_ms = [], 0.0, 0.0, 0.0
    for name, ms, (now_b, strip, later_b) in zip(names, pass_ms, bytes_per_dispatch):
        x = (latency_us * 1e-3 + now_b / (link_gbs * 1e9) * 1e3) if now_b else 0.0
        split = now_b > 0 and own_rows >= 4 * strip
        strips_ms = ms * min(neighbours * strip / max(own_rows, 1), 1.0) if split else (ms if now_b else 0.0)
        interior_ms = ms - strips_ms if split else (0.0 if now_b else ms)
        unhidden = max(0.0, x - interior_ms) if now_b else 0.0
        deferred_ms += (latency_us * 1e-3 + later_b / (link_gbs * 1e9) * 1e3) if later_b else 0.0
        rows.append({"dispatch": name, "compute_ms": round(ms, 4), "strips_first_bytes": int(now_b), "exchange_ms": round(x, 4),
                     "interior_ms": round(interior_ms, 4), "unhidden_ms": round(unhidden, 4)})
        total += ms + unhidden
        unhidden_total += unhidden
    compute = sum(pass_ms)
    return {"link_GBs_per_direction": link_gbs, "latency_us_per_exchange": latency_us, "compute_ms": round(compute, 4), "unhidden_exchange_ms": round(unhidden_total, 4),
            "deferred_exchange_ms": round(deferred_ms, 4), "deferred_fits_behind_frame": deferred_ms <= total,
            "predicted_frame_ms": round(total + max(0.0, deferred_ms - total), 4), "serial_frame_ms": round(compute + sum(r["exchange_ms"] for r in rows) + deferred_ms, 4),
            "per_dispatch": rows}


class BandHarness(Harness):
    """Harness for one row band: planes are local_h rows tall, CommonSettings describe the whole frame."""

    def __init__(self, backend, denoisers, width, frame_h, rank, world, halo=DEFAULT_HALO, bounds=None):
        self.layout = band_layout(frame_h, world, rank, halo, bounds)
        self.bounds = bounds if bounds is not None else band_bounds(frame_h, world)
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


class _Exchange:
    """the handles of one batch of sends / receives + what has to happen to the received rows once they are there"""

    def __init__(self, works, post):
        self.works, self.post = works, post

    def wait(self):
        for w in self.works:
            w.wait()
        for f in self.post:
            f()


class Tiler:
    """Steps a BandHarness dispatch by dispatch, exchanging halo rows after each one."""

    def __init__(self, band, dist, group=None):
        self.band, self.dist, self.group = band, dist, group
        self.bytes_exchanged = 0
        self.split_dispatches = 0  # dispatches run as boundary strips + interior with the exchange in flight
        self._plan_cache = {}
        self._p2p_cache = {}  # dispatch todo -> (P2POp list, bytes sent): pool planes never move, so the row slices are built once
        self._deferred = {}   # dispatch list -> exchanges of planes only the NEXT frame reads: in flight until that list's next call
        self._later_todo = []  # ... and their (plane, rows, skip) items while a dispatch list is still running
        self._staging = {}     # tap-texel planes (nrdhip_dispatch_info.written_prefix): the signal halves of the rows that travel

    def _as_tensor(self, buf):
        import torch

        return buf if hasattr(buf, "data_ptr") else torch.from_numpy(buf)

    def _plane_of(self, code):
        pool, index = code >> 16, code & 0xFFFF
        if pool > 1:
            return None
        return self.band.nrd.pools[pool][index]

    def reprojection_rows(self, dispatches):
        """rows of previous-frame state a band needs beyond its own: the motion allowance the stored halo leaves on top of the widest
        spatial reach of the list (halo = reach + motion_rows rounded up to 16: required_halo) + 2 rows for the bilinear footprint.
        Only valid when EVERY read of previous-frame state in the list - a permanent plane read before the list writes it - says it is
        reprojected (or at the pixel's own position): one spatial reader of last frame's planes and the full halo travels, as before."""
        from . import api

        written = set()
        for d in dispatches:
            for code, rows in zip(d["read"], d.get("read_rows", [d["halo_rows"]] * len(d["read"]))):
                if (code >> 16) == 0 and code not in written and rows not in (0, api.READ_REPROJECTED):
                    return self.band.halo
            written.update(d["written"])
        motion = self.band.halo - max([d["halo_rows"] for d in dispatches] + [0])
        return max(min(self.band.halo, motion + 2), 0)

    def _plan(self, ids, dispatches):
        """for every dispatch: (now, later), two lists of (plane code, rows[, skip[, guide plane]]) over the pool planes it writes.
        now: boundary rows a LATER DISPATCH OF THIS FRAME reads - the largest reach INTO THAT PLANE (nrdhip_dispatch_info.read_rows: 0 for
        a read at the pixel's own position, the window for a 5x5 stencil, the tap reach for a gather) of the dispatches that read it
        before it is written again; these are exchanged strips-first and waited for before the next dispatch.
        later: permanent planes that survive the frame are read by the NEXT frame's reprojection at motion-displaced rows - the band's
        motion allowance + 2 rows (reprojection_rows), not the whole halo (round 3 sent all 80 rows of history, fast history, speeds,
        stabilized luma and guide: a third of the volume) - and nobody needs those rows before the next frame: their exchange is
        enqueued after the dispatch and stays in flight behind the rest of the frame (no strips, no wait).
        Nothing a pass with `all_rows` writes travels at all: the ClassifyTiles passes run on every stored row of the band (their inputs
        carry valid halo rows), so guide and tile planes are complete on every rank."""
        from . import api

        key = tuple((d["name"], tuple(d["written"]), tuple(d["read"]), tuple(d.get("read_rows", ())), d.get("all_rows", False), d["halo_rows"]) for d in dispatches)
        L = self.band.layout
        if key in self._plan_cache:
            plan, reproj = self._plan_cache[key]
            self.band.nrd.set_history_rows(L["own_first"] - reproj, L["own_rows"] + 2 * reproj)
            return plan
        reproj = self.reprojection_rows(dispatches)
        # the kernels must not trust previous-frame rows this plan never refreshes: a footprint beyond owned rows +- reproj is rejected
        # (a disocclusion) instead of being read from stale halo rows (include/nrdhip.h nrdhip_set_history_rows)
        self.band.nrd.set_history_rows(L["own_first"] - reproj, L["own_rows"] + 2 * reproj)

        def reach_into(r, code):
            rows = r.get("read_rows")
            v = rows[r["read"].index(code)] if rows else r["halo_rows"]
            return reproj if v == api.READ_REPROJECTED else v

        plan = []
        for i, d in enumerate(dispatches):
            now, later = [], []
            for code in d["written"]:
                if (code >> 16) > 1 or d.get("all_rows"):
                    continue  # output slots are final; a pass over all stored rows leaves nothing to exchange
                rows, rewritten = 0, False
                for r in dispatches[i + 1:]:
                    if code in r["read"]:
                        rows = max(rows, reach_into(r, code))
                    if code in r["written"]:
                        rewritten = True
                        break
                if rows > self.band.halo:  # never clamp: a clamped reach is a silently different image
                    raise HaloError("a pass after %s reads %d rows beyond its band, the band stores %d: create the bands with "
                                    "halo=required_halo(dispatches, motion_rows) (probe_halo())" % (d["name"], rows, self.band.halo))
                # a plane of tap texels {guide texel | signal} names the guide plane its texels start with (written_prefix): entry + (0, guide)
                guide = d.get("written_prefix", {}).get(code)
                if (code >> 16) == 0 and not rewritten and rows < reproj:
                    later.append((code, reproj, rows) + (() if guide is None else (guide,)))  # the `rows` nearest the band edge travel at once (below)
                if rows > 0:
                    now.append((code, rows) + (() if guide is None else (0, guide)))
            plan.append((now, later))
        self._plan_cache[key] = (plan, reproj)
        return plan

    def _ops(self, bufs_rows):
        """send / recv descriptors for [(2-D byte tensor [local rows, pitch], rows-per-texel-row divisor, rows, skip[, guide tensor])]:
        the `rows` owned rows next to each band edge go to that neighbour's halo; `skip` = how many of them (nearest the edge)
        an earlier exchange already delivered. Returns (ops, pre, post).
        An item with a guide tensor is a plane of 16-byte tap texels {guide texel | signal} (written_prefix): only the signal half travels,
        through a staging tensor - `pre` gathers the halves to send (run before the batch is enqueued), `post` scatters the received
        halves into the halo rows and puts the guide half back from this rank's own guide plane (run after the wait)."""
        dist, b = self.dist, self.band
        L = b.layout
        ops, pre, post = [], [], []
        W = b.w

        def texels(t, bpt):  # [rows, pitch] bytes -> [rows, W, bpt]
            return t.unflatten(1, (t.shape[1] // bpt, bpt))[:, :W]

        def send(t, a, z, peer, guide, key):
            if z <= a:
                return
            if guide is None:
                ops.append((dist.isend, t[a:z], peer))
                return
            st = self._stage(key, (z - a, W, 8), t)
            src = texels(t[a:z], 16)[:, :, 8:16]
            pre.append(lambda: st.copy_(src))
            ops.append((dist.isend, st, peer))

        def recv(t, a, z, peer, guide, key):
            if z <= a:
                return
            if guide is None:
                ops.append((dist.irecv, t[a:z], peer))
                return
            st = self._stage(key, (z - a, W, 8), t)
            dst, g = texels(t[a:z], 16), texels(guide[a:z], 8)

            def scatter():
                dst[:, :, 8:16] = st
                dst[:, :, 0:8] = g
            post.append(scatter)
            ops.append((dist.irecv, st, peer))

        for n_item, item in enumerate(bufs_rows):
            t, div, rows = item[:3]
            hskip = (item[3] // div) if len(item) > 3 else 0
            guide = item[4] if len(item) > 4 else None
            hrows = max((rows + div - 1) // div, 1)
            first, n = L["own_first"] // div, max(L["own_rows"] // div, 1)
            total = t.shape[0]
            key = (t.data_ptr(), rows, hskip)
            if b.rank > 0:  # upper neighbour: send my first owned rows, receive into my top halo
                send(t, first + hskip, first + min(hrows, n), b.rank - 1, guide, key + ("su",))
                recv(t, first - min(hrows, first), first - hskip, b.rank - 1, guide, key + ("ru",))
            if b.rank < b.world - 1:  # lower neighbour
                send(t, first + n - min(hrows, n), first + n - hskip, b.rank + 1, guide, key + ("sd",))
                recv(t, first + n + hskip, min(first + n + hrows, total), b.rank + 1, guide, key + ("rd",))
        return ops, pre, post

    def _stage(self, key, shape, like):
        import torch

        st = self._staging.get(key)
        if st is None or tuple(st.shape) != tuple(shape):
            st = self._staging[key] = torch.empty(shape, dtype=torch.uint8, device=like.device)
        return st

    def exchange_start(self, bufs_rows):
        """enqueue the exchange (RCCL: on the communicator's stream, ordered after the work already on the current stream);
        returns the handles to wait on"""
        ops, pre, post = self._ops(bufs_rows)
        if not ops:
            return []
        dist = self.dist
        for fn, ten, _ in ops:
            if fn is dist.isend:
                self.bytes_exchanged += ten.numel() * ten.element_size()
        for f in pre:
            f()
        if ops[0][1].is_cuda and dist.get_backend(self.group) != "nccl":
            # only RCCL orders itself after the work already enqueued on the current stream; any other backend (gloo in the
            # tests) must see finished rows
            import torch
            torch.cuda.current_stream().synchronize()
        return [_Exchange(dist.batch_isend_irecv([dist.P2POp(fn, ten, peer, self.group) for fn, ten, peer in ops]), post)]

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
            ops, pre, post = self._ops(self.items_of(todo))
            sent = sum(ten.numel() * ten.element_size() for fn, ten, _ in ops if fn is dist.isend)
            hit = ([dist.P2POp(fn, ten, peer, self.group) for fn, ten, peer in ops], sent,
                   bool(ops) and ops[0][1].is_cuda and dist.get_backend(self.group) != "nccl", pre, post)
            self._p2p_cache[key] = hit
        p2p, sent, host_sync, pre, post = hit
        if not p2p:
            return []
        self.bytes_exchanged += sent
        for f in pre:
            f()
        if host_sync:  # see exchange_start
            import torch
            torch.cuda.current_stream().synchronize()
        return [_Exchange(self.dist.batch_isend_irecv(p2p), post)]

    def items_of(self, todo):
        local_h = self.band.layout["local_h"]
        items = []
        for code, rows, *rest in todo:  # rest: [skip[, guide plane]]
            p = self._plane_of(code)
            div = 1 if p["height"] == local_h else 16  # pool planes: full resolution, or one texel per 16x16 tile
            item = (self._as_tensor(p["buf"]), div, rows, rest[0] if rest else 0)
            g = rest[1] if len(rest) > 1 else None
            if g is not None:  # tap texels: the guide half stays at home
                item += (self._as_tensor(self._plane_of(g)["buf"]),)
            items.append(item)
        return items

    def finish(self, ids=None):
        """wait for the exchanges still in flight (rows only the next frame reads): all of them, or (`ids`) those the dispatch list `ids` sent
        behind its previous call - another list's call inside the same frame (SIGMA, then REBLUR, then REFERENCE) does not wait for them"""
        # (`ids`: every list that SHARES a denoiser with it - the same denoiser may appear in two lists, [REBLUR] one frame, [REBLUR, SIGMA] the
        # next, and its permanent rows must have arrived whichever list sent them: csrc/nrdhip_tiler.cpp wait_deferred_sharing)
        keys = list(self._deferred) if ids is None else [k for k in list(self._deferred) if set(k) & set(ids)]
        for k in keys:
            for w in self._deferred.pop(k, []):
                w.wait()

    def run_dispatch(self, ids, i, plan_entry, last=None):
        """one dispatch + the halo exchange of what it wrote. When the band is tall enough the rows a neighbour needs are
        computed FIRST (boundary strips), their exchange is started, and the interior is computed while the rows travel -
        the copies overlap the compute instead of serialising with it (nrdhip_denoise_rows). Rows that only the next frame
        reads follow without strips and without a wait (see _plan)."""
        if i == 0:
            self.finish(ids)  # this list's next call: the permanent planes it sent behind its last call must have arrived
            self._later_todo = []
        todo, later = plan_entry
        self._run_dispatch_now(ids, i, todo)
        # rows only the NEXT frame reads: collected over the dispatch list and sent as ONE batch behind its last dispatch (`last`
        # known: five batch calls per frame become one - the launch thread is the bottleneck when a band is a fraction of a
        # millisecond of GPU work); `last` unknown: sent right away
        self._later_todo += later
        if (last is None or last) and self._later_todo:
            self._deferred.setdefault(tuple(ids), []).extend(self.exchange_start_cached(self._later_todo))
            self._later_todo = []

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

    def check_halo(self, dispatches, motion_rows=0):
        need = max([d["halo_rows"] for d in dispatches] + [0]) + motion_rows
        if self.band.world > 1 and need > self.band.halo:
            worst = max(dispatches, key=lambda d: d["halo_rows"])
            raise HaloError("%s reads %d rows (+ %d rows of motion) beyond its band, the band stores %d: create the bands with "
                            "halo=probe_halo(...)" % (worst["name"], worst["halo_rows"], motion_rows, self.band.halo))

    def denoise(self, identifiers):
        ids = [int(i) for i in identifiers]
        nrd = self.band.nrd
        dispatches = nrd.dispatches(ids)
        self.check_halo(dispatches)
        plan = self._plan(ids, dispatches)
        for i, entry in enumerate(plan):
            self.run_dispatch(ids, i, entry, last=(i == len(plan) - 1))


class NativeTiler:
    """The row tiler BELOW the C-ABI (csrc/nrdhip_tiler.cpp, include/nrdhip.h nrdhip_tiler_*): exchange plan, strips-first
    overlap and the exchanges themselves live in C++. ``transport``: "rccl" - ncclSend / ncclRecv groups on a side stream (the
    ncclUniqueId of rank 0 travels through torch.distributed once); "dist" - callbacks that move the rows with
    torch.distributed isend / irecv (gloo in the CPU tests; any backend), used to check the C++ plan without RCCL."""

    def __init__(self, band, dist=None, transport="rccl", group=None):
        import ctypes as C

        self.band, self.dist, self.group = band, dist, group
        b = band.backend
        if not getattr(b, "has_tiler", False):
            raise RuntimeError("this backend library has no nrdhip_tiler_* entry points")
        self._works, self._keep = [], []
        self._tr = None
        if transport == "dist":
            self._tr = api.Transport(None, api.TRANSPORT_BEGIN(self._begin), api.TRANSPORT_XFER(self._send), api.TRANSPORT_XFER(self._recv),
                                     api.TRANSPORT_END(self._end))
        h = C.c_void_p()
        r = b.tiler_create(band.nrd.handle, band.rank, band.world, C.byref(self._tr) if self._tr is not None else None, C.byref(h))
        if r != 0:
            raise api.NrdError("nrdhip_tiler_create (band shorter than its halo?)", r)
        self.handle = h
        if transport == "rccl" and band.world > 1:
            import torch

            uid = torch.zeros(128, dtype=torch.uint8)
            if band.rank == 0:
                buf = (C.c_uint8 * 128)()
                self._check(b.tiler_rccl_unique_id(buf), "ncclGetUniqueId")
                uid = torch.tensor(list(buf), dtype=torch.uint8)
            dev = uid.to(b.device) if dist.get_backend(group) == "nccl" else uid
            dist.broadcast(dev, 0, group=group)
            raw = (C.c_uint8 * 128)(*dev.cpu().tolist())
            self._check(b.tiler_rccl_init(self.handle, raw), "ncclCommInitRank")

    def _check(self, r, what):
        if r != 0:
            raise api.NrdError(what, r, (self.band.backend.tiler_last_error(self.handle) or b"").decode())

    # ---- "dist" transport: the C++ tiler calls back with raw (pointer, bytes) ranges of planes this process allocated
    def _view(self, ptr, nbytes):
        nrd = self.band.nrd
        cands = [p["buf"] for pool in (0, 1) for p in nrd.pools[pool]] + list(nrd._bound.values())
        for buf in cands:
            base = buf.data_ptr() if hasattr(buf, "data_ptr") else buf.ctypes.data
            size = buf.numel() * buf.element_size() if hasattr(buf, "numel") else buf.nbytes
            if base <= ptr and ptr + nbytes <= base + size:
                flat = buf.reshape(-1) if hasattr(buf, "data_ptr") else buf.reshape(-1)
                off = ptr - base
                view = flat[off:off + nbytes]
                if not hasattr(view, "data_ptr"):
                    import torch
                    view = torch.from_numpy(view)
                return view
        # not a plane: a staging buffer the C++ tiler allocated itself (the packed signal halves of tap-texel rows)
        import torch

        if self.band.backend.is_device:
            class _Raw:  # a raw device range as the CUDA array interface describes it
                __cuda_array_interface__ = dict(shape=(nbytes,), typestr="|u1", data=(ptr, False), version=2)
            return torch.as_tensor(_Raw(), device=self.band.backend.device)
        import ctypes as C

        return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))

    def _begin(self, user):
        self._works, self._keep = [], []
        return 0

    def _xfer(self, fn, ptr, nbytes, peer):
        try:
            t = self._view(ptr, nbytes)
            if t.is_cuda and self.dist.get_backend(self.group) != "nccl":
                import torch
                torch.cuda.synchronize()
            self._keep.append(t)
            self._works.append(fn(t, peer, group=self.group))
            return 0
        except Exception as e:  # never raise through the C frame
            self._error = e
            return 1

    def _send(self, user, ptr, nbytes, peer, stream):
        return self._xfer(self.dist.isend, ptr, nbytes, peer)

    def _recv(self, user, ptr, nbytes, peer, stream):
        return self._xfer(self.dist.irecv, ptr, nbytes, peer)

    def _end(self, user, stream):
        try:
            for w in self._works:
                w.wait()
            self._works, self._keep = [], []
            return 0
        except Exception as e:
            self._error = e
            return 1

    # ---- the Tiler interface
    def _stream(self):
        return self.band.nrd._stream()

    def exchange_inputs(self, planes):
        import ctypes as C

        keys = set(planes)
        slots = sorted(int(slot) for slot, (key, _) in INPUT_SLOTS.items() if key in keys and key != "confidence")
        self.band.bind(planes)
        arr = (C.c_uint32 * len(slots))(*slots)
        self._check(self.band.backend.tiler_exchange_inputs(self.handle, arr, len(slots), self._stream()), "tiler_exchange_inputs")

    def denoise(self, identifiers):
        import ctypes as C

        ids = [int(i) for i in identifiers]
        arr = (C.c_uint32 * len(ids))(*ids)
        self._check(self.band.backend.tiler_denoise(self.handle, arr, len(ids), self._stream()), "tiler_denoise")

    def finish(self):
        self._check(self.band.backend.tiler_finish(self.handle, self._stream()), "tiler_finish")

    def stats(self):
        import ctypes as C

        out = (C.c_uint64 * 4)()
        self.band.backend.tiler_stats(self.handle, out)
        return dict(bytes_sent=int(out[0]), split_dispatches=int(out[1]), exchanges=int(out[2]), deferred=int(out[3]))

    @property
    def bytes_exchanged(self):
        return self.stats()["bytes_sent"]

    @property
    def split_dispatches(self):
        return self.stats()["split_dispatches"]

    def destroy(self):
        if self.handle is not None:
            self.band.backend.tiler_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class TiledRunner:
    """bench.py's N > 1 path: every rank renders its band of the synthetic scene on its GPU, then steps the tiler."""

    def __init__(self, pkg, backend, device, dens, width, frame_h, rank, world, unique, dolly, settings_of, tiler="python",
                 motion_rows=DEFAULT_MOTION_ROWS, balance=True):
        import torch
        import torch.distributed as dist

        from .harness import pingpong  # same camera path as the single-GPU runner

        self.pingpong = pingpong
        self.api, self.dens, self.unique = pkg.api, dens, unique
        # the rows a band stores beyond its own come from the settings actually used (reach of every pass + motion), not a constant
        probe_scene = pkg.synth.Scene(64, 64, dolly=dolly, device=device)
        self.halo = probe_halo(backend, dens, settings_of(pkg.api, probe_scene, dens), motion_rows)
        # Band boundaries balance the COST of the bands, not their height: a tile without geometry costs a fraction of a filtered one,
        # and the sky of this scene sits in the upper rows. The profile comes from the depth of the first frame at one texel per tile
        # (every rank renders the same 1/16-resolution proxy: no communication); a renderer would use last frame's Tiles mask and
        # re-create its bands on an accumulation restart.
        self.bounds = None
        if balance and world > 1:
            proxy = pkg.synth.Scene((width + 15) // 16, (frame_h + 15) // 16, dolly=dolly, device="cpu")
            pz = proxy.frame(0, noise=False)
            rng = float(proxy.common_settings(pkg.api, pz, 0).denoisingRange)
            vz = pz["viewz"]
            vz = vz.cpu().numpy() if hasattr(vz, "cpu") else vz
            self.bounds = band_bounds(frame_h, world, tile_row_cost_from_depth(vz, rng), min_rows=self.halo)
        self.band = BandHarness(backend, dens, width, frame_h, rank, world, halo=self.halo, bounds=self.bounds)
        if tiler == "native":
            self.tiler = NativeTiler(self.band, dist, transport="rccl" if dist.get_backend() == "nccl" else "dist")
        else:
            self.tiler = Tiler(self.band, dist)
        self.native = tiler == "native"
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
        if not self.events_on or self.native:  # the C++ tiler steps the dispatches itself (no per-dispatch events)
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
            self.tiler.run_dispatch(ids, i, entry, last=(i == len(plan) - 1))
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

    def dispatch_table(self):
        """[(pass name, algorithmic bytes per pixel)] of one frame (for the roofline object when no per-dispatch events exist)"""
        return [(x["name"], x["bytes_per_pixel"]) for x in self.band.nrd.dispatches(self.ids)]


def verify_tiled_against_single(pkg, backend, device, dens, width, frame_h, rank, world, settings_of, dolly, halo, bounds, tiler="python",
                                frames=3):
    """Bit-identity of the row-tiled frame against ONE instance over the whole frame, on the workload's own frame size: every rank
    renders the same full frames (same seeds, same code: identical planes on every GPU), cuts out its band, and steps `frames` frames
    from a restart through the tiler; rank 0 also runs the single instance and compares every output row the bands own.
    Returns (identical, detail) on rank 0, (None, "") elsewhere."""
    import torch
    import torch.distributed as dist

    api = pkg.api
    scene = pkg.synth.Scene(width, frame_h, dolly=dolly, device=device)
    settings = settings_of(api, scene, dens)
    band = BandHarness(backend, dens, width, frame_h, rank, world, halo=halo, bounds=bounds)
    t = NativeTiler(band, dist, transport="rccl" if dist.get_backend() == "nccl" else "dist") if tiler == "native" else Tiler(band, dist)
    single = Harness(backend, dens, width, frame_h) if rank == 0 else None
    L = band.layout
    ids = [int(d) for d in dens]
    for f in range(frames):
        fr = scene.frame(f)
        cs = scene.common_settings(api, fr, f, reset=(f == 0))
        local = {k: (v[L["row0"]:L["row0"] + L["local_h"]] if hasattr(v, "shape") and v.shape[0] == frame_h else v) for k, v in fr.items()}
        planes = band.upload(local)
        band.nrd.new_frame()
        band.nrd.set_common_settings(cs)
        band.bind(planes)
        for d in dens:
            band.nrd.set_denoiser_settings(int(d), settings[d])
        t.denoise(ids)
        if single is not None:
            single.frame(cs, single.upload(fr), settings)
        del fr, local, planes
    t.finish()
    torch.cuda.synchronize()
    keys = [k for k in ("out_diff", "out_spec") if k in band.outputs]
    bad = []
    for key in keys:
        own = band.own_rows(band.outputs[key])
        own = own.contiguous() if hasattr(own, "contiguous") else torch.from_numpy(np.ascontiguousarray(own))  # (numpy planes: the CPU backends of the tests)
        if rank == 0:
            ref = single.outputs[key]
            ref = ref if hasattr(ref, "contiguous") else torch.from_numpy(ref)
            rows = [own] + [None] * (world - 1)
            for r in range(1, world):
                b0, b1 = band.bounds[r], band.bounds[r + 1]
                buf = torch.empty((b1 - b0, own.shape[1]), dtype=own.dtype, device=own.device)
                dist.recv(buf, src=r)
                rows[r] = buf
            for r in range(world):
                b0, b1 = band.bounds[r], band.bounds[r + 1]
                if not torch.equal(rows[r], ref[b0:b1]):
                    bad.append("%s rows %d-%d (rank %d): %d bytes differ" % (key, b0, b1, r, int((rows[r] != ref[b0:b1]).sum().item())))
        else:
            dist.send(own, dst=0)
    if hasattr(t, "destroy"):
        t.destroy()
    band.nrd.destroy()
    if single is not None:
        single.nrd.destroy()
    torch.cuda.empty_cache()
    if rank != 0:
        return None, ""
    return (not bad), ("; ".join(bad) if bad else "%d frames, %s, every owned row of %d bands" % (frames, " + ".join(keys), world))
