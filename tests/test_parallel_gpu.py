"""Two data-parallel ranks on ONE MI355X (both processes use cuda:0, gloo carries the all-reduce through the host): the real step path
of world_size > 1 -- bucketed gradient all-reduce on the communication stream followed by that bucket's RAdam + L2 launch, both overlapped
with the rest of the backward pass, the 1/world scaling in the optimizer, the three-kernel BatchNorm backward -- rehearsed on hardware.  RCCL itself needs one GPU per rank and is
exercised by the driver's multi-GPU bench; everything around the collective is the same code.

Checked: both ranks hold bit-identical weights after 3 steps (same summed gradient, same update), those weights differ from the start,
the loss is finite, and the summed gradient of step 1 equals the sum of the two single-rank gradients computed without any overlap."""
import os
import subprocess
import sys
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import bench
from yolov3_tensorflow_amd import parallel
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
dist.init_process_group('gloo')
model, loss, opt, grids = bench.build_model('resnet-18', 224, 224, 4, 13, dev)
assert parallel.setup_data_parallel(model, backend='gloo')
assert model.world_size == 2 and model.g.fused_bn_bwd is False
images, labels = bench.synthetic_batch(4, 224, 224, 13, rank)          # every rank its own shard
model.stage_batch(images, labels)
w0 = model.g.ps.flat.clone()
# reference gradient of this rank's shard: plain forward / backward, no collective, no overlap
model.g.training = True
model.g.on_bucket = None
model._fwd_bwd()
torch.cuda.synchronize()
g_local = model.g.ps.grad.clone()
model.g.ps.grad.zero_()
if model.loss_obj.current_num is not None:
    model.loss_obj.current_num.zero_()
g_sum = g_local.clone()
dist.all_reduce(g_sum, op=dist.ReduceOp.SUM)
# step 1 through the real path, gradient captured just before the optimizer consumes it
# (per gradient bucket: each range is all-reduced and then updated on the communication stream while the backward pass continues)
captured = torch.zeros_like(g_sum)
ranges = []
orig_range = opt.launch_range
def spy(m, lo, hi, first):
    torch.cuda.synchronize()
    captured[lo:hi] = model.g.ps.grad[lo:hi]
    ranges.append((lo, hi, first))
    orig_range(m, lo, hi, first)
opt.launch_range = spy
model.run_step()
opt.launch_range = orig_range
torch.cuda.synchronize()
# (world size 2: the stage buckets are cut further at 16 MB -- module512 + heads no longer travel as one ~47 MB collective)
assert len(ranges) >= 4 and [r[2] for r in ranges] == [True] + [False] * (len(ranges) - 1), ranges
assert sum(hi - lo for lo, hi, _ in ranges) == model.g.ps.n and max(hi - lo for lo, hi, _ in ranges) < 0.5 * model.g.ps.n, ranges
assert all(ranges[i + 1][1] == ranges[i][0] for i in range(len(ranges) - 1)) and ranges[-1][0] == 0, ranges
err = float((captured - g_sum).abs().max()) / max(float(g_sum.abs().max()), 1e-12)
for _ in range(2):
    model.run_step()
torch.cuda.synchronize()
w = model.g.ps.flat.detach().cpu()
other = w.clone()
dist.broadcast(other, src=0)
same = bool(torch.equal(w, other))
moved = float((w - w0.cpu()).abs().max())
lossv = float(model.loss_value.item())
print('RESULT rank %%d grad_rel_err %%.3e same %%s moved %%.3e loss %%.5f' %% (rank, err, same, moved, lossv), flush=True)
assert err < 1e-5, err
assert same and moved > 0 and np.isfinite(lossv)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_on_one_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    script = tmp_path / 'ddp_worker.py'
    script.write_text(WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK='0'), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail('data-parallel worker timed out')
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, out[-3000:])
        assert 'RESULT rank %d' % r in out


def test_bench_command_at_two_ranks():
    """the driver's multi-GPU command line, rehearsed with both ranks on this one GPU (gloo): bench.py must reach its JSON line --
    in particular nothing after the timed loop may issue a collective on a subset of the ranks"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import json
    env = dict(os.environ, YOLO_BENCH_SHARE_GPU='1', YOLO_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29623', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--size', '224',
           '--batch', '4', '--classes', '13']
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    assert p.returncode == 0, p.stdout[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-3000:]                       # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 8 and out['config']['parallelism'] == 'dp2' and out['value'] > 0
    assert 'roofline' in out and 'cpu_baseline' not in out          # the CPU baseline is an N = 1 item
