"""CPU oracle for the mkb triplet-scoring / self-adversarial / negative-sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mkb_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.

It is a restatement (own code) of the reference algorithms, each function citing the
``/root/reference`` file:line it follows:

* ``oracle.scoring``   torch-CPU fp32, same op order as ``mkb/models/*.py`` + autograd
* ``oracle.closed``    numpy float64 closed-form scores / loss / gradients (independent of autograd)
* ``oracle.sampler``   own MT19937 + masked rejection + the three ``np.in1d`` paths
* ``oracle/csrc``      the sampler in plain C (built by ``oracle/Makefile`` -> ``liboracle.so``)

Parity pinning: the restatement is checked (a) against golden vectors captured from the live
reference import in the build container (``tools/make_golden.py`` -> ``tests/golden/*.npz``)
and (b) against the reference's own doctest known-answers (``sampling/negative_sampling.py:101-126``,
``models/*.py`` init doctests, ``evaluation/evaluation.py:101-119``), see ``tests/test_oracle_*.py``.
"""
