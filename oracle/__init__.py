"""CPU oracle for the YOLOv3 training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (PyTorch-CPU / NumPy, float32 by
default, float64 selectable) of the reference algorithm in /root/reference (zheng-yuwei/YOLOv3-tensorflow).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only
as the checker.  The product package (``yolov3_tensorflow_amd``) never imports it and fails loudly when its
HIP library is missing.

Pinning status:
  * ``oracle.postprocess`` is pinned against golden vectors generated from the reference's own
    ``yolov3/yolov3_post_process.py`` (tests/golden/make_postprocess_golden.py -> tests/golden/postprocess_*.npz).
  * everything that the reference delegates to TensorFlow (conv/BN/pool arithmetic, autodiff, the tf.* ops of the
    decoder / loss, the Keras optimizer plumbing) is **parity unpinned**: TensorFlow is not installed here, the
    reference has no tests or golden vectors for it, so these restatements follow the reference source line by
    line plus documented TF semantics and are cross-checked only by self-consistency tests
    (hand-derived gradients vs autograd, known-answer cases computed by hand).
"""
