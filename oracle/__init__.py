"""CPU oracle for the SCaSML_GP hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm (multilevel-Picard
Monte-Carlo + PDE-constrained Gaussian process on ``Grad_Dependent_Nonlinear``).
It exists to check the HIP product path (``scasml_gp_amd``) and to be timed as the
``cpu_baseline`` leg of ``bench.py``.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``scasml_gp_amd/`` does.

Parity status (see DESIGN.md "Oracle"): the reference cannot be imported in this
image (``jax``/``deepxde``/``optax`` are not installed: an ordinary
ModuleNotFoundError, nothing was denied), it ships no golden vectors, and its random
numbers come from JAX threefry, an un-vendored dependency.  The oracle is therefore
pinned by what the reference's own files do pin:

* the closed-form exact solution      (equations/equations.py:306-323),
* the integer tables Q / Mf / Mg      (solvers/MLP.py:57-69, 111-139),
* the reference-compat ``lgwt`` node / weight tables for q in {1, 3, 4}
  (solvers/MLP.py:71-109, including the scalar assignment at :99),
* the recursion call counts 19 (n=rho=2) and 5 (full history n=2) that the committed
  cProfile dumps record (results/**/Grad_Dependent_Nonlinear_rho_2.prof),
* the statistical band of the logged relative-L2 errors (results/**/*.log),
* Random123 known-answer vectors for Philox4x32-10.

Bit-level parity with the reference's own random stream is "parity unpinned" by
construction (threefry vs Philox); everything deterministic is pinned as above.
"""
