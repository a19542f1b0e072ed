"""The C-ABI RCCL gather (rvseg_comm_* / rvseg_gather_frames): world size 1 on the one-GPU box, and a
two-thread world on one device where RCCL allows it (skipped otherwise).  The N-GPU run is the driver's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_gather_world_size_one(gpu_ctx_factory):
    torch = pytest.importorskip("torch")
    import rovinasemanticsegmentation_amd as rv
    dev = torch.device("cuda", 0)
    ctx = gpu_ctx_factory(width=160, height=120)
    uid = rv.Context.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(0, 1, uid)
    local = torch.arange(0, 3 * 160 * 120, device=dev).remainder(120).to(torch.int8)
    recv = torch.full_like(local, -1)
    s = torch.cuda.current_stream(dev).cuda_stream
    ctx.gather_frames(local.data_ptr(), local.numel(), recv.data_ptr(), 0, s)
    torch.cuda.synchronize(dev)
    assert torch.equal(recv, local)
    # a second communicator on the same context is refused; arguments are checked
    with pytest.raises(rv.capi.RvsegError):
        ctx.comm_init(0, 1, uid)
    with pytest.raises(rv.capi.RvsegError):
        ctx.gather_frames(local.data_ptr(), local.numel(), recv.data_ptr(), 3, s)


def test_gather_without_communicator_is_an_error(gpu_ctx_factory):
    import rovinasemanticsegmentation_amd as rv
    ctx = gpu_ctx_factory(width=160, height=120)
    with pytest.raises(rv.capi.RvsegError) as e:
        ctx.gather_frames(1, 16, 1, 0, 0)
    assert e.value.status == rv.capi.ERR_INVALID_ARG
