"""Host-side result container -- same surface as the reference's
``tfmpc/utils/trajectory.py:11-90`` (``Trajectory``, ``Transition``), accepting
torch / numpy arrays and an optional leading batch axis.

Shapes after construction: ``states[..., T+1, n]``, ``actions[..., T, m]`` (the
trailing singleton axis of the column vectors is squeezed as in
``trajectory.py:13-14``) and ``costs[..., T+1]`` (the reference leaves LQR costs as
``[T+1,1,1]`` and iLQR costs as ``[T+1]``; both are flattened here -- quirk Q12).
"""

import os
from collections import namedtuple

import numpy as np

Transition = namedtuple("Transition", "state action cost")


def _to_numpy(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


class Trajectory:
    def __init__(self, states, actions, costs):
        states, actions, costs = _to_numpy(states), _to_numpy(actions), _to_numpy(costs)
        self.states = np.squeeze(states, axis=-1) if states.shape[-1] == 1 else states
        self.actions = np.squeeze(actions, axis=-1) if actions.shape[-1] == 1 else actions
        while costs.ndim > self.states.ndim - 1 and costs.shape[-1] == 1:
            costs = np.squeeze(costs, axis=-1)
        self.costs = costs

    @property
    def batched(self):
        return self.states.ndim == 3

    @property
    def initial_state(self):
        return self.states[..., 0, :]

    @property
    def final_state(self):
        return self.states[..., -1, :]

    @property
    def total_cost(self):
        return np.sum(self.costs, axis=-1)

    @property
    def cumulative_cost(self):
        return np.cumsum(self.costs, axis=-1)

    @property
    def cost_to_go(self):
        return np.cumsum(self.costs[..., ::-1], axis=-1)[..., ::-1]

    def __len__(self):
        return self.actions.shape[-2]

    def instance(self, b):
        """The b-th trajectory of a batched result."""
        if not self.batched:
            raise IndexError("not a batched trajectory")
        return Trajectory(self.states[b][..., None], self.actions[b][..., None], self.costs[b])

    def __getitem__(self, t):
        if t >= len(self) or t < -len(self):
            raise IndexError(t)
        return Transition(self.states[..., t + 1, :], self.actions[..., t, :], self.costs[..., t])

    def __repr__(self):
        if self.batched:
            return (f"Trajectory(batch={self.states.shape[0]}, horizon={len(self)}, "
                    f"mean_total={float(np.mean(self.total_cost)):.4f})")
        return f"Trajectory(init={self.initial_state}, final={self.final_state}, total={self.total_cost:.4f})"

    def __str__(self):
        if self.batched:
            return repr(self)
        rows = [("Steps", "States", "Actions", "Costs")]
        for t, (state, action, cost) in enumerate(self):
            state = "[" + ", ".join(f"{x:8.4f}" for x in state) + "]"
            action = "[" + ", ".join(f"{u:8.4f}" for u in action) + "]"
            rows.append((str(t), state, action, f"{cost:8.4f}"))
        sizes = [max(map(len, col)) for col in zip(*rows)]
        out = " | ".join(h.center(sz) for h, sz in zip(rows[0], sizes)) + "\n"
        out += " | ".join("=" * sz for sz in sizes) + "\n"
        for row in rows[1:]:
            out += " | ".join(col.center(sz) for col, sz in zip(row, sizes)) + "\n"
        return out

    def save(self, filepath):
        """CSV in the reference's format (``trajectory.py:71-90``): index ``Timestep``,
        columns ``x[i]`` (state AFTER the step), ``u[i]``, ``costs`` (final cost dropped)."""
        import pandas as pd

        if self.batched:
            raise ValueError("save() writes one trajectory; use .instance(b).save(path)")
        df = pd.DataFrame()
        for i, x_i in enumerate(np.transpose(self.states[1:])):
            df[f"x[{i+1}]"] = x_i
        for i, u_i in enumerate(np.transpose(self.actions)):
            df[f"u[{i+1}]"] = u_i
        df["costs"] = self.costs[:-1]
        dirname = os.path.dirname(filepath)
        if dirname and not os.path.exists(dirname):
            os.makedirs(dirname)
        df.to_csv(filepath, index=True, index_label="Timestep")
