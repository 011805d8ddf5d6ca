"""Solvers of the hot path: :mod:`tfmpc.solvers.lqr` (Riccati sweep + rollout) and :mod:`tfmpc.solvers.ilqr`
(control-limited iLQR), both thin ctypes front ends of ``tfmpc/_lib/libtfmpc_hip.so``."""
