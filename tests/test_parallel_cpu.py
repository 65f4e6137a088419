"""world_size-2 gloo tests (CPU) of the data-parallel host logic: weight broadcast from rank 0, SUM all-reduce of the flat gradient and
the 1/world scaling folded into the optimizer, checked against the single-process reference semantics (keras multi_gpu_model,
trainer.py:40-43: per-tower BatchNorm, one loss = mean over the concatenated batch => gradient = average of the tower gradients)."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeBN(object):
    def __init__(self, c):
        self.moving_mean, self.moving_var = torch.zeros(c), torch.ones(c)


class _FakePS(object):
    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.flat = torch.randn(n, generator=g)
        self.grad = torch.zeros(n)
        self.bf16 = torch.zeros(n, dtype=torch.bfloat16)
        self.n = n


class _FakeGraph(object):
    def __init__(self, n, rank):
        self.ps = _FakePS(n, rank)
        self.bns = [_FakeBN(4)]

    def refresh_dgrad_weights(self):
        pass


class _FakeModel(object):
    def __init__(self, n, rank):
        self.g = _FakeGraph(n, rank)
        self.device = torch.device('cpu')
        self.world_size, self.rank, self.process_group = 1, 0, None

    def set_distributed(self, w, r, pg=None):
        self.world_size, self.rank, self.process_group = w, r, pg


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank))
    import contextlib
    from yolov3_tensorflow_amd import parallel, ops
    ops.cast_f32_to_bf16 = lambda x, y, n: y.copy_(x.to(torch.bfloat16))          # the only kernel touched by broadcast_weights
    torch.cuda.device = lambda d: contextlib.nullcontext()
    m = _FakeModel(1024, rank)
    m.g.bns[0].moving_mean += rank
    assert parallel.setup_data_parallel(m, backend='gloo')
    assert (m.world_size, m.rank) == (world, rank)
    w_after = m.g.ps.flat.clone()
    # each rank's "tower gradient" on its half of the batch
    gen = torch.Generator().manual_seed(7 + rank)
    tower = torch.randn(1024, generator=gen)
    m.g.ps.grad.copy_(tower)
    dist.all_reduce(m.g.ps.grad, op=dist.ReduceOp.SUM)
    scaled = m.g.ps.grad * (1.0 / m.world_size)                                    # grad_scale of yolo_radam_l2_step
    out[rank] = (w_after.numpy(), m.g.bns[0].moving_mean.numpy().copy(), tower.numpy(), scaled.numpy(), m.g.ps.bf16.float().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_broadcast_and_average():
    world, port = 2, 29731
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    w0, bn0, t0, s0, b0 = out[0]
    w1, bn1, t1, s1, b1 = out[1]
    np.testing.assert_array_equal(w0, w1)                       # all ranks start from rank 0's weights
    np.testing.assert_array_equal(bn0, bn1)
    np.testing.assert_array_equal(bn0, np.zeros(4, np.float32))
    np.testing.assert_array_equal(b0, torch.from_numpy(w0).to(torch.bfloat16).float().numpy())
    np.testing.assert_allclose(s0, (t0 + t1) / 2, rtol=1e-6)    # average of the tower gradients, identical on every rank
    np.testing.assert_array_equal(s0, s1)


def test_single_process_is_a_noop(monkeypatch):
    from yolov3_tensorflow_amd import parallel
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    m = _FakeModel(256, 0)
    assert parallel.setup_data_parallel(m) is False and m.world_size == 1


def _agree_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank))
    from yolov3_tensorflow_amd import parallel
    dist.init_process_group('gloo')
    # the ranks see different shards, hence different epoch losses: rank 1's own loss stops improving at epoch 2, rank 0's never does
    own = [[10.0, 9.0, 8.0, 7.0, 6.0, 5.0], [10.0, 9.0, 9.5, 9.5, 9.5, 9.5]][rank]
    local, agreed = parallel.EarlyStopping(1e-4, 2), parallel.EarlyStopping(1e-4, 2)
    local_stop = agreed_stop = None
    means = []
    for epoch, loss in enumerate(own):
        if local_stop is None and local.should_stop(loss):
            local_stop = epoch
        mean = parallel.agree_mean(loss)
        means.append(mean)
        if agreed_stop is None and agreed.should_stop(mean):
            agreed_stop = epoch
    failure = parallel.agree_any(rank == 1)                       # only rank 1 hit a device-protocol error
    nofail = parallel.agree_any(False)
    out[rank] = (local_stop, agreed_stop, means, failure, nofail)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_early_stopping_and_failures():
    """every rank must take the same early-stopping decision and leave together on a failure, or the others hang in the next all-reduce"""
    world, port = 2, 29741
    out = mp.Manager().dict()
    mp.spawn(_agree_worker, args=(world, port, out), nprocs=world, join=True)
    (l0, a0, m0, f0, n0), (l1, a1, m1, f1, n1) = out[0], out[1]
    assert l0 != l1                                               # decided locally the ranks would diverge ...
    assert a0 == a1 and m0 == m1                                  # ... on the agreed mean they do not
    np.testing.assert_allclose(m0, [10.0, 9.0, 8.75, 8.25, 7.75, 7.25])
    assert f0 is True and f1 is True and n0 is False and n1 is False


def test_per_rank_batch_is_the_global_batch_divided():
    from yolov3_tensorflow_amd import parallel
    assert parallel.per_rank_batch(256, 8) == 32 and parallel.per_rank_batch(3, 1) == 3
    with pytest.raises(ValueError):
        parallel.per_rank_batch(3, 2)


# ---- gradient buckets (engine.plan_buckets): stage marks, and the finer cut a data-parallel run uses ----
def _resnet18_like_layout():
    """(convs, n) of a ResNet18-YOLOv3-shaped parameter buffer: one entry per block of layers, creation order (backbone, then heads)"""
    blocks = [(1728, 208), (4 * 36864 + 4096, 104), (73728 + 147456 * 3 + 8192, 52), (294912 + 589824 * 3 + 32768, 26),
              (1179648 + 2359296 + 131072, 13), (2 * 2359296, 13), (2359296 + 130560, 13), (1179648 + 131072 + 1179648 + 130560, 26),
              (32768 + 32768 + 294912 + 65280, 52)]
    convs, off = [], 0
    for i, (params, h) in enumerate(blocks):
        convs.append((len(blocks) - 1 - i, off, h))
        off += params
    return convs, off


def test_bucket_plan_stage_marks_and_size_cuts():
    from yolov3_tensorflow_amd.engine import plan_buckets
    convs, n = _resnet18_like_layout()
    stage = plan_buckets(convs, 416, n, None)
    assert [b[1:] for b in stage] == [(2774720, None), (153280, 2774720)]          # [first stride-32 conv, n), [first stride-8 conv, that)
    fine = plan_buckets(convs, 416, n, 4 << 20)                                    # 16 MB of float32 gradient
    los = [lo for _, lo, _ in fine]
    assert los == sorted(los, reverse=True) and los[-1] == 153280 and 2774720 in los and len(fine) >= 4
    assert fine[0][2] is None and all(fine[i + 1][2] == fine[i][1] for i in range(len(fine) - 1))     # a partition of [tail, n)
    cuts = [c for c, _, _ in fine]
    assert cuts == sorted(cuts)                                                    # completion order = backward launch order
    sizes = [(n if hi is None else hi) - lo for _, lo, hi in fine]
    assert max(sizes) < 0.5 * n                                                    # no single collective carries most of the gradient any more
    assert plan_buckets(convs, 416, n, 1 << 40) == stage                           # a huge target changes nothing


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank))
    from yolov3_tensorflow_amd.engine import plan_buckets
    dist.init_process_group('gloo')
    convs, n = _resnet18_like_layout()
    buckets = plan_buckets(convs, 416, n, 4 << 20)
    ranges = [(lo, n if hi is None else hi) for _, lo, hi in buckets] + [(0, buckets[-1][1])]       # + the tail range after the backward pass
    gen = torch.Generator().manual_seed(11 + rank)
    grad = torch.randn(n // 64, generator=gen).repeat_interleave(64)              # (cheap to generate, still rank-specific everywhere)
    whole = grad.clone()
    for lo, hi in ranges:                                                          # bucket by bucket, in completion order, as the step does
        dist.all_reduce(grad[lo:hi], op=dist.ReduceOp.SUM)
    dist.all_reduce(whole, op=dist.ReduceOp.SUM)
    out[rank] = (bool(torch.equal(grad, whole)), sum(hi - lo for lo, hi in ranges) == n, len(ranges))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_one_allreduce_world2():
    """the finer buckets of a data-parallel run partition the flat gradient: all-reducing them one by one (gloo, two ranks) gives the sum
    one collective over the whole buffer gives, bit for bit"""
    world, port = 2, 29751
    out = mp.Manager().dict()
    mp.spawn(_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1] and out[0][0] and out[0][1] and out[0][2] >= 5
