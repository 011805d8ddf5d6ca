"""Closed-loop episode driver: the env is stepped under an agent for the horizon configured by
``env.setup`` and the visited (state, action, cost) triples come back as a ``Trajectory``.
Same surface as the reference's ``tfmpc/runners/__init__.py:8-49`` (``Runner(env, agent).run()``
and the ``with runner(x0, T) as r`` form); here ``B`` episodes advance together when the
initial state is ``[B,n,1]`` -- one fused iLQR launch per control step for all of them."""

import contextlib

import torch

from tfmpc.utils.trajectory import Trajectory


class Runner:

    def __init__(self, env, agent):
        self.env, self.agent = env, agent

    @contextlib.contextmanager
    def __call__(self, initial_state, horizon):
        self.env.setup(initial_state, horizon)
        try:
            yield self
        finally:
            self.env.close()

    def run(self, mode=None):
        return Trajectory(*self.run_device(mode))

    def run_device(self, mode=None):
        """The episode as device tensors (states [.., T+1, n, 1], actions [.., T, m, 1], costs [.., T+1]); nothing is
        read back to the host."""
        env, agent = self.env, self.agent
        x = env.reset()
        reset_agent = getattr(agent, "reset", None)
        if callable(reset_agent):
            reset_agent()
        batched = x.dim() == 3
        time_axis = 1 if batched else 0
        visited, applied, paid = [x], [], []
        if env.horizon < 1:
            raise ValueError("Runner.run needs a horizon of at least one step")
        finished = False
        while not finished:
            u = agent(x, env._t)                       # plan over the remaining horizon, apply the first action
            x, stage_cost, finished, _ = env.step(u)
            if mode is not None:
                env.render(mode)
            visited.append(x)
            applied.append(u)
            paid.append(stage_cost)
        paid.append(env.final_cost(x, batch=batched))
        return (torch.stack(visited, dim=time_axis), torch.stack(applied, dim=time_axis), torch.stack(paid, dim=time_axis))

    def capture(self, initial_state, horizon, noise):
        """The whole closed-loop episode as ONE hipGraph (``torch.cuda.CUDAGraph``): every control step of an MPC episode
        is a handful of small launches (the fused iLQR solve over the remaining horizon, the env's transition / cost
        kernels, a few element-wise ops) with Python in between, and at small batches the loop is bound by launch and
        interpreter overhead, not by the GPU.  Captured once for a batch shape and horizon, an episode is then a copy of
        its inputs into static buffers and one graph launch.

        ``initial_state``: ``[B,n,1]`` (or ``[n,1]``); ``noise``: the env's raw noise draws for the T steps, as for
        ``GymEnv.inject_noise`` (a captured episode cannot draw from a host-seeded generator per step; inject the draws).
        Returns ``episode(initial_state, noise) -> (Trajectory, iterations[T])``.  The agent must be an ``agents.MPC``;
        its cold-start actions are generated up front (same values as the eager loop draws step by step), so a
        captured episode reproduces the eager one bit for bit (tests/test_mpc_graph_gpu.py)."""
        return CapturedEpisode(self, initial_state, horizon, noise)


class CapturedEpisode:

    def __init__(self, runner, initial_state, horizon, noise):
        import copy
        if getattr(getattr(runner.agent, "solver", None), "_generic_env", False):
            raise ValueError("Runner.capture: a solver on a TorchEnv drives its iterations from the host (it reads "
                             "results back between launches) and cannot run inside a graph capture")
        # the capture works on PRIVATE shallow copies of the env and the agent: device mode, the start actions of this
        # batch shape and the injected static noise buffers would otherwise leak into a later eager run() of the
        # caller's objects
        env, agent = copy.copy(runner.env), copy.copy(runner.agent)
        dev = env._device()
        self.runner, self.horizon = Runner(env, agent), int(horizon)
        self.x0 = torch.as_tensor(initial_state, dtype=torch.float32).to(dev).clone()
        self.noise = [torch.as_tensor(s, dtype=torch.float32).to(dev).clone() for s in noise]
        if len(self.noise) != self.horizon:
            raise ValueError("capture needs one noise draw per control step")
        agent.on_device = True
        agent.reset()
        agent.prepare_start_actions(self.x0.shape[0] if self.x0.dim() == 3 else None)
        env._injected = self.noise                      # the static buffers themselves: replays read what __call__ copied in
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                   # warm-up outside the capture (lazy initialisation, allocator pools)
            for _ in range(2):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.tensors, self.iterations = self._run()

    def _run(self):
        with self.runner(self.x0, self.horizon) as r:
            tensors = r.run_device()
        return tensors, torch.stack([i.reshape(-1) for i in r.agent.iterations])

    def __call__(self, initial_state=None, noise=None):
        if initial_state is not None:
            self.x0.copy_(torch.as_tensor(initial_state, dtype=torch.float32), non_blocking=True)
        if noise is not None:
            noise = list(noise)
            if len(noise) != len(self.noise):
                raise ValueError(f"a captured episode of {len(self.noise)} steps needs {len(self.noise)} noise draws, got {len(noise)}")
            for dst, src in zip(self.noise, noise):
                src = torch.as_tensor(src, dtype=torch.float32)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"noise draw of shape {tuple(src.shape)}, captured with {tuple(dst.shape)}")
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return Trajectory(*self.tensors), self.iterations.cpu().numpy()         # (the conversions synchronise)
