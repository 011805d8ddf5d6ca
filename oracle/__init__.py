"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the tf-mpc v0.7.0 hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / timed CPU baseline.  The product path
(``tf-mpc_amd/``) never imports this package and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):

* The reference's arithmetic lives in TensorFlow 2 (``tensorflow-cpu``, version
  unpinned in the reference's ``setup.py:48-58``).  TensorFlow, gym and
  tuneconfig are not installed in the build container and there is no network,
  so the reference cannot be imported or run here.
* PINNED against the reference's own known answers: the box-QP solutions of
  ``tests/test_utils_optimization.py:7-15``, every env derivative closed form in
  ``tests/test_env_*.py``, the LQR forward-consistency / value-function
  identities of ``tests/test_lqr.py:51-86`` and the ``navlin`` table in
  ``README.md:75-90`` (which was produced with a zero terminal value function,
  see ``lqr_ref.backward(terminal="zero")``).
* PARITY UNPINNED for numeric iLQR trajectories (K, k, accepted step sizes,
  iteration counts, final cost): the reference's tests hold no numeric answer
  for them (``tests/test_ilqr.py:114`` has its only such assert commented out).
  For those the fp64 restatement here is the oracle of record and reports must
  say "vs own fp64 restatement of the tf-mpc v0.7.0 equations".
"""
