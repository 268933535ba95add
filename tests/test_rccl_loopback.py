""""execute the RCCL symbols once before an 8-GPU node does it for you".

The row tiler's RCCL transport (csrc/nrdhip_tiler.cpp: librccl through dlopen, ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a side
stream, evCompute -> side stream -> evComm -> compute stream, a separate event for deferred groups) had run nowhere: every multi-rank test
goes through the caller-supplied transports (gloo, stream-ordered loopback). Here ONE rank on ONE GPU creates the communicator
(ncclCommInitRank, nranks = 1) and pushes real exchange groups addressed to itself through that very code
(nrdhip_tiler_rccl_loopback -> comm_after_compute -> run_ops -> event ordering), rows compared byte for byte with their source.
If RCCL refuses (no bootstrap interface on the box, self-send unsupported), the error text is recorded under gpurun_out/ and the test fails
with it - INTEGRATION.md quotes the outcome."""
import ctypes as C
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(text):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_loopback.txt"), "a") as f:
        f.write(text + "\n")


@pytest.mark.timeout(300)
def test_rccl_send_recv_to_self_through_the_tilers_exchange_path(pkg, api, hip):
    import torch

    assert hip.has_tiler and "tiler_rccl_loopback" in hip._fn
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # the box has no network: the bootstrap of a one-rank communicator only needs loopback
    D = api.Denoiser
    hz = pkg.harness.Harness(hip, [D.REBLUR_DIFFUSE_SPECULAR], 256, 144)
    h = C.c_void_p()
    assert hip.tiler_create(hz.nrd.handle, 0, 1, None, C.byref(h)) == 0
    try:
        uid = (C.c_uint8 * 128)()
        r = hip.tiler_rccl_unique_id(uid)
        if r != 0:
            record("ncclGetUniqueId failed (librccl not loadable?) rc=%d" % r)
        assert r == 0, "ncclGetUniqueId"
        r = hip.tiler_rccl_init(h, uid)
        if r != 0:
            record("ncclCommInitRank(nranks=1) failed: %s" % hip.tiler_last_error(h).decode())
        assert r == 0, hip.tiler_last_error(h).decode()
        stream = torch.cuda.current_stream().cuda_stream
        # the shapes a band edge sends: 80 halo rows of a 16-byte-per-pixel plane at 3840 and at 7680 width, and a short odd-sized one
        for case, (n, deferred) in enumerate([(80 * 3840 * 16, 0), (80 * 7680 * 16, 0), (12345, 0), (80 * 7680 * 16, 1), (4099, 1)]):
            src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
            dst = torch.zeros_like(src)
            # work in front of the exchange on the compute stream: the side stream must pick up BEHIND it (evCompute) ...
            src.add_(1)
            r = hip.tiler_rccl_loopback(h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, deferred, C.c_void_p(stream))
            if r != 0:
                record("ncclSend / ncclRecv to self refused: %s" % hip.tiler_last_error(h).decode())
            assert r == 0, hip.tiler_last_error(h).decode()
            # ... and work behind it must see the rows (evComm / the deferred event): a copy enqueued on the compute stream, no host sync in between
            seen = dst.clone()
            torch.cuda.synchronize()
            assert torch.equal(seen, src), "case %d: %d bytes, deferred=%d" % (case, n, deferred)
        stats = (C.c_uint64 * 4)()
        assert hip.tiler_stats(h, stats) == 0
        assert stats[0] >= 2 * 80 * 7680 * 16 and stats[2] == 3 and stats[3] == 2  # bytes sent, in-frame exchanges, deferred exchanges
        record("OK: ncclCommInitRank(nranks=1) + 5 groups {ncclSend, ncclRecv} to self through run_ops on the side stream, rows byte-identical; "
               "bytes sent %d, in-frame exchanges %d, deferred %d" % (stats[0], stats[2], stats[3]))
    finally:
        hip.tiler_destroy(h)
