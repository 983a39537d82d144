"""CPU oracle for the SCaSML_GP hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm (multilevel-Picard
Monte-Carlo + PDE-constrained Gaussian process on ``Grad_Dependent_Nonlinear``).
It exists to check the HIP product path (``scasml_gp_amd``) and to be timed as the
``cpu_baseline`` leg of ``bench.py``.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``scasml_gp_amd/`` does.

Parity status: the reference cannot be imported in this image (``jax``/``deepxde``/``optax`` are not installed:
an ordinary ModuleNotFoundError, nothing was denied) and ships no golden vectors -- but its runs printed what they computed, and the
oracle is pinned by those numbers, deterministically:

* ``replay.py`` + ``jax_random.py``: ``MLP.u_solve`` with the reference's own random stream (JAX threefry, float16 normals, its key
  schedule) and the dtype of every operation: the MLP numbers of results/**/SimpleUniform.log to all sixteen printed digits and of
  RepeatedExperiment.log to the seven printed, at d = 20, 40, 60, 80 (tests/test_reference_replay.py);
* ``gp_compat.py`` on the reference's training / test sets (``equation.deepxde_points``) and Hutchinson indices: its printed "Real Solution"
  to sixteen digits, its logged GP errors to 3e-4 (tests/test_reference_logs.py);
* the closed-form exact solution (equations/equations.py:306-323), the integer tables Q / Mf / Mg (solvers/MLP.py:57-69, 111-139), the
  reference-compat ``lgwt`` tables (:71-109, including the scalar assignment at :99), the recursion call counts 19 and 5 of the committed
  cProfile dumps, Random123 known-answer vectors for Philox4x32-10 and Threefry-2x32.

``mlp.py`` -- the estimator the HIP path is checked against -- runs the same recursion in float64 on Philox normals keyed by tree position;
tests/test_reference_replay.py ties the two together by feeding ``mlp.py`` the replay's normals.
"""
