"""Data parallelism: one process per GPU, gradient all-reduce over RCCL/xGMI (torch.distributed backend 'nccl'), replacing
keras.utils.multi_gpu_model (reference yolov3/trainer.py:40-43).  BatchNorm statistics stay per replica (as in the reference's
towers); gradients are summed and scaled by 1/world inside the fused RAdam kernel; the rectified-image counter advances by
the GLOBAL batch (yolov3_loss.py:151-152 semantics of the single-process reference)."""
import os
import torch


def setup_data_parallel(model, backend=None):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return False
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend or os.environ.get('YOLO_DIST_BACKEND', 'nccl'))
    model.set_distributed(dist.get_world_size(), dist.get_rank())
    broadcast_weights(model)
    return True


def broadcast_weights(model):
    """all ranks start from rank 0's weights (the reference shares one set of variables between towers)"""
    import torch.distributed as dist
    ps = model.g.ps
    dist.broadcast(ps.flat, src=0, group=model.process_group)
    for bn in model.g.bns:
        dist.broadcast(bn.moving_mean, src=0, group=model.process_group)
        dist.broadcast(bn.moving_var, src=0, group=model.process_group)
    from . import ops
    with torch.cuda.device(model.device):
        ops.cast_f32_to_bf16(ps.flat, ps.bf16, ps.n)
        model.g.refresh_dgrad_weights()


def world_size():
    return int(os.environ.get('WORLD_SIZE', '1'))


def per_rank_batch(global_batch, world=None):
    """FLAGS.batch_size is the GLOBAL batch, as in the reference (keras multi_gpu_model splits the one batch of ``fit`` across the towers,
    trainer.py:40-43); each rank trains on ``global_batch / world`` images and steps_per_epoch stays ``ceil(train_set_size / batch_size)``"""
    world = world_size() if world is None else int(world)
    if world < 1 or global_batch % world:
        raise ValueError('FLAGS.batch_size (%d, the global batch) must be a multiple of the number of ranks (%d)' % (global_batch, world))
    return global_batch // world


def _reduce_scalar(value, op, device=None, group=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    on_gpu = dist.get_backend(group) == 'nccl'
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if on_gpu else 'cpu')
    dist.all_reduce(t, op=op, group=group)
    return float(t.item())


def agree_mean(value, device=None, group=None):
    """the same number on every rank: the mean of the ranks' values (epoch loss: every rank sees another shard, and a per-rank early-stopping
    decision would leave the other ranks waiting in the next step's all-reduce)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    return _reduce_scalar(value, dist.ReduceOp.SUM, device, group) / dist.get_world_size(group)


def agree_any(flag, device=None, group=None):
    """True on every rank if ``flag`` is true on any (a failure on one rank must stop all of them together)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    return _reduce_scalar(1.0 if flag else 0.0, dist.ReduceOp.MAX, device, group) > 0.0


class EarlyStopping(object):
    """keras EarlyStopping(monitor='loss', min_delta, patience) of the reference (trainer.py:92-93) on a loss all ranks agree on"""

    def __init__(self, min_delta, patience):
        self.min_delta, self.patience = float(min_delta), int(patience)
        self.best, self.wait = float('inf'), 0

    def should_stop(self, loss):
        if self.best - loss > self.min_delta:
            self.best, self.wait = loss, 0
            return False
        self.wait += 1
        return self.wait >= self.patience
