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
