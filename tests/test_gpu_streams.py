"""The start-up probe that picks the step's side streams (include/cpc_hip.h: cpc_streams_overlap; ops.pick_concurrent_stream): the
HIP runtime maps streams onto four hardware queues per priority in creation order, two streams on one queue serialise."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda:0")


def test_a_stream_does_not_overlap_with_itself_and_arguments_are_checked():
    dev = _dev()
    from cpc_audio_amd import _lib, ops
    s = torch.cuda.Stream(dev)
    assert ops.streams_overlap(s, s) is False
    lib = _lib.get()
    assert lib.cpc_streams_overlap(ctypes.c_void_p(s.cuda_stream), ctypes.c_void_p(s.cuda_stream), None) == 2      # CPC_ERR_ARG


def test_pool_streams_share_hardware_queues_and_the_pick_avoids_that():
    """Seven normal-priority streams cannot all have a queue of their own (four queues): the probe must see at least one
    serialised pair among them, and pick_concurrent_stream must return streams that overlap pairwise with what they were
    picked against."""
    dev = _dev()
    from cpc_audio_amd import ops
    pool = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(6)]
    serial = [(i, j) for i in range(len(pool)) for j in range(i + 1, len(pool)) if not ops.streams_overlap(pool[i], pool[j])]
    print("serialised pairs among the default stream + 6 pool streams:", serial)
    assert serial, "seven streams on four queues and no shared queue seen: the probe is blind"
    main = torch.cuda.current_stream(dev)
    a = ops.pick_concurrent_stream(dev, 0, [main])
    b = ops.pick_concurrent_stream(dev, 0, [main, a])
    c = ops.pick_concurrent_stream(dev, -1, [main, a, b])
    for x, y in ((main, a), (main, b), (a, b), (main, c), (a, c), (b, c)):
        assert ops.streams_overlap(x, y) and ops.streams_overlap(y, x)


def test_step_context_streams_are_pairwise_concurrent_after_rccl_created_its_own():
    """What a data-parallel rank sees: the process group (and its six streams) first, the train loop's streams after."""
    dev = _dev()
    import os
    import torch.distributed as dist
    from cpc_audio_amd import ops
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        t = torch.ones(8, device=dev)
        dist.all_reduce(t)
        ops._side_streams.clear()
        ctx = ops.StepContext(overlap=True)
        streams = [torch.cuda.current_stream(dev)] + ctx.reserve(dev)
        assert ops.StepContext(overlap=True).reserve(dev) == streams[1:]          # one set of side streams per process
        # no high-priority stream: its queue would be the process's fifth
        assert all(st.priority == 0 for st in streams[1:]), [st.priority for st in streams]
        for i in range(4):
            for j in range(4):
                if i != j:
                    assert ops.streams_overlap(streams[i], streams[j]), (i, j)
    finally:
        ops._side_streams.clear()
        if own:
            dist.destroy_process_group()
