"""The pieces of keras.backend state the reference touches (run.py:21-26): the fuzz factor and the learning phase -- plus the
16-bit compute type of this implementation ('bfloat16' default, 'float16' selects libyolov3_amd_fp16.so and static loss scaling)."""
_EPSILON = 1e-8          # the reference calls keras.backend.set_epsilon(1e-8) before building anything (run.py:26)
_LEARNING_PHASE = True


def epsilon():
    return _EPSILON


def set_epsilon(value):
    global _EPSILON
    _EPSILON = float(value)


def set_learning_phase(value):
    global _LEARNING_PHASE
    _LEARNING_PHASE = bool(value)


def learning_phase():
    return _LEARNING_PHASE


_COMPUTE_DTYPE = 'bfloat16'
_LOSS_SCALE = {'bfloat16': 1.0, 'float16': 1024.0}


def compute_dtype():
    return _COMPUTE_DTYPE


def set_compute_dtype(name):
    """'bfloat16' or 'float16' ('bf16' / 'fp16' accepted).  Set it before building a model: buffers are allocated in this type."""
    global _COMPUTE_DTYPE
    name = {'bf16': 'bfloat16', 'fp16': 'float16', 'half': 'float16'}.get(name, name)
    if name not in _LOSS_SCALE:
        raise ValueError("compute dtype must be 'bfloat16' or 'float16', got %r" % (name,))
    _COMPUTE_DTYPE = name


def torch_dtype():
    import torch
    return torch.float16 if _COMPUTE_DTYPE == 'float16' else torch.bfloat16


def loss_scale():
    """factor on the 16-bit gradients that enter the backward pass (undone inside the optimizer kernel): float16 has 5 exponent bits, the
    d(logits) / N of a large batch would fall into its subnormals"""
    return _LOSS_SCALE[_COMPUTE_DTYPE]


def set_loss_scale(value):
    _LOSS_SCALE[_COMPUTE_DTYPE] = float(value)
