from tfmpc.agents.mpc import MPC  # noqa: F401
