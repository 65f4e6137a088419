"""The two pieces of keras.backend state the reference touches (run.py:21-26): the fuzz factor and the learning phase."""
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
