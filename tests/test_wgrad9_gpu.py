"""wgrad9_kernel (conv_wgrad9.hip: stationary-output weight gradient of 3x3 / stride-1 convolutions, padded coordinates, X through an LDS ring)
through the C-ABI, forced with yolo_set_tuning("wgrad9", 1), against the strip / generic weight-gradient kernels ("wgrad9" = 0) on the same inputs
(same bf16 products, float32 sums in another order) and against float32 torch autograd.  Shapes: several (co, ci) units, pixel ranges that end
inside an image / inside a row, maps narrower than a ring piece's reach, many small images (the ring wraps every few stages), one-stage splits."""
import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from yolov3_tensorflow_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


@pytest.fixture(params=['bf16', 'fp16'])
def dtype(request):
    from yolov3_tensorflow_amd import backend
    if request.param == 'fp16':
        backend.set_compute_dtype('float16')
    yield request.param
    backend.set_compute_dtype('bfloat16')


SHAPES = [
    # N, H, W, Cin, Cout
    (2, 52, 52, 64, 64),
    (3, 21, 19, 128, 128),      # four units, ragged last stage, image boundaries inside stages
    (40, 13, 13, 64, 128),      # many small images: rows and images wrap inside every piece
    (1, 104, 104, 64, 64),      # the widest benchmark map: ring of 416 rows
    (2, 7, 9, 64, 64),          # the smallest map the kernel takes
    (5, 26, 30, 128, 64),
    (32, 26, 26, 256, 256),     # benchmark layer: 16 units x 16 splits
]


@pytest.mark.parametrize('shape', SHAPES, ids=str)
def test_wgrad9_matches_strip_kernel_and_reference(dev, dtype, shape):
    from yolov3_tensorflow_amd import ops, backend
    N, H, W, Cin, Cout = shape
    ACT = backend.torch_dtype()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, H, W, Cin, generator=g).to(ACT).to(dev)
    dy = torch.randn(N, H, W, Cout, generator=g).to(ACT).to(dev)
    p = ops.conv_problem(N, H, W, Cin, Cout, 3, 1, 'same')

    def run():
        splits = ops.conv2d_wgrad_splits(p)
        dw = torch.zeros(Cout, 3, 3, Cin, device=dev)
        ops.conv2d_wgrad(p, x, dy, dw)                                   # atomics into a zeroed buffer
        ws = torch.empty(max(ops.conv2d_wgrad_workspace_bytes(p), 16) // 4, device=dev)
        dw2 = torch.full_like(dw, 7.0)
        ops.conv2d_wgrad_reduce(p, x, dy, dw2, ws)                        # slabs + summing pass: overwrites
        dw3 = torch.empty_like(dw)
        ops.conv2d_wgrad_reduce(p, x, dy, dw3, ws)
        torch.cuda.synchronize()
        assert torch.equal(dw2, dw3), 'two-phase weight gradient must be run-to-run deterministic'
        return dw.cpu(), dw2.cpu(), splits

    try:
        ops.set_tuning('wgrad9', 0)
        ref = run()
        ops.set_tuning('wgrad9', 1)
        got = run()
    finally:
        ops.set_tuning('wgrad9', -1)
    units = (Cin // 64) * (Cout // 64)
    stages = (N * (H + 1) * (W + 1) + 63) // 64
    want_splits = min(max(128 // units, 1), stages)              # ("wgrad9_wgs" = 128, the library default)
    sps = (stages + want_splits - 1) // want_splits
    assert got[2] == (stages + sps - 1) // sps
    scale = ref[1].abs().max().item()
    for t in got[:2]:
        torch.testing.assert_close(t, ref[1], rtol=1e-4, atol=2e-5 * scale)
    if N * H * W * Cin * Cout <= 2 ** 31:                 # float32 autograd reference on the CPU (skipped for the benchmark-sized case)
        xr = x.float().cpu().permute(0, 3, 1, 2)
        w0 = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
        F.conv2d(xr, w0, padding=1).backward(dy.float().cpu().permute(0, 3, 1, 2))
        torch.testing.assert_close(got[1], w0.grad.permute(0, 2, 3, 1), rtol=1e-3, atol=1e-4 * max(scale, 1.0))
