"""Receding-horizon controllers over the batched HIP solvers (caller side of the hot path)."""
from .mpc import MPC

__all__ = ["MPC"]
