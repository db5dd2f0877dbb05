"""CPU oracle for the micrograph-denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
directory: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

The oracle restates, on the CPU, the TensorFlow-1.x op semantics that the
reference's graphs are built from (the reference has no native compute code;
all of its arithmetic lives in TensorFlow, which is not vendored under
/root/reference and is not installable here):

* ``tf_ops.py``            TF op semantics (SAME padding, depthwise / dense /
                           transposed convolution, inference batch-norm, relu6,
                           legacy bilinear resize, REFLECT pad) in two
                           independent forms: PyTorch-CPU and plain numpy.
* ``kernel_denoiser.py``   graph K  - misc_py/noise-removal-kernels.py:96-431
* ``k_oracle.c``           graph K again in plain C (also the timed CPU port)
* ``denoiser_graph.py``    graph D  - machine_learning/denoiser.py:58-398

PARITY UNPINNED: the reference ships no tests, golden vectors, checkpoints
or sample outputs for this path and TensorFlow
cannot be run here, so the oracle is pinned only by (a) the analytic
known-answer tests listed in SURVEY.md 8(c) and (b) agreement between its two
independent implementations of every op.
"""
