"""MI355X-native drop-in for the solver path of tf-mpc (thiagopbueno/tf-mpc v0.7.0).

Same import paths and class / method names as the reference for the hot path:

    tfmpc.solvers.lqr.LQR              (reference: tfmpc/solvers/lqr.py)
    tfmpc.solvers.ilqr.iLQR            (reference: tfmpc/solvers/ilqr.py)
    tfmpc.envs.make_lqr, make_lqr_linear_navigation, make_env
    tfmpc.envs.{diffenv, lqr.navigation, navigation, hvac, reservoir}
    tfmpc.utils.trajectory.Trajectory, tfmpc.utils.optimization.projected_newton_qp

Tensors are torch (ROCm) / numpy instead of ``tf.Tensor`` and every argument may
carry one extra leading batch axis ``B`` of independent problem instances.  All
arithmetic of ``backward`` / ``forward`` / ``solve`` runs in hand-written gfx950
HIP kernels behind the C ABI in ``include/tfmpc_hip.h``; there is no CPU fallback:
calling a solver without the built library or without a GPU raises.
"""

__version__ = "0.1.0"
