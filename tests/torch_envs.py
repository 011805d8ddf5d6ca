"""Three of the reference's envs written as PLAIN torch functions of one instance (x[n], u[m]) -- what a user of the reference has: Python
methods, not device templates (/root/reference/tfmpc/envs/navigation/__init__.py:34-74, reservoir/__init__.py:47-105,
hvac/__init__.py:69-149, restated on torch with the vector shapes of tfmpc.envs.torchenv.TorchEnv).  tests/test_fxenv_gpu.py hands them to
TorchEnv(...).to_device_env() and holds the result against the built-in kernels and the oracle; bench.py's `deviceenv_from_python` line times
the Navigation one.  Written with the ordinary torch vocabulary on purpose (broadcasting, matmul, boolean masks, relu / abs / where)."""

import numpy as np
import torch

from tfmpc.envs.torchenv import TorchEnv


def _t(a, device):
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=device)


def navigation(cfg, device="cpu"):
    goal = _t(cfg["goal"], device).reshape(-1)
    centers = _t(cfg["deceleration"]["center"], device).reshape(-1, 2)
    decay = _t(cfg["deceleration"]["decay"], device).reshape(-1)

    def transition(x, u):
        dist = torch.linalg.norm(x[None, :] - centers, dim=-1)                  # distance to every zone centre
        lam = torch.prod(2.0 / (1.0 + torch.exp(-decay * dist)) - 1.0)          # joint deceleration factor
        return x + lam * u

    def cost(x, u):
        return torch.sum((x - goal) ** 2)

    def final_cost(x):
        return torch.sum((x - goal) ** 2)

    return TorchEnv(transition, cost, final_cost, 2, 2, np.asarray(cfg["low"]).reshape(2, 1), np.asarray(cfg["high"]).reshape(2, 1), device=device)


def reservoir(cfg, device="cpu"):
    col = lambda k: _t(cfg[k], device).reshape(-1)
    cap, lo, hi = col("max_res_cap"), col("lower_bound"), col("upper_bound")
    low_pen, high_pen, sp_pen = -col("low_penalty"), -col("high_penalty"), -col("set_point_penalty")      # (the configs hold them negative)
    rain = col("rain_shape") * col("rain_scale")                                                           # mean rainfall (cec=True)
    downstream = _t(cfg["downstream"], device)
    n = cap.numel()

    def transition(x, u):
        outflow = u * x
        inflow = downstream.T @ outflow
        vaporated = 0.5 * torch.sin(x / cap) * x
        return x + rain + inflow - vaporated - outflow

    def cost(x, u):
        mid = (lo + hi) / 2
        penalty = low_pen * torch.relu(lo - x) + high_pen * torch.relu(x - hi) + sp_pen * torch.abs(mid - x)
        return penalty.sum()

    return TorchEnv(transition, cost, lambda x: cost(x, None), n, n, 0.0, 1.0, device=device)


def hvac(cfg, device="cpu"):
    CAP_AIR, COST_AIR, TEMP_AIR, TIME_DELTA, PENALTY, SET_POINT_PENALTY = 1.006, 1.0, 40.0, 1.0, 20000.0, 10.0
    col = lambda k: _t(cfg[k], device).reshape(-1)
    t_out, t_hall, lo, hi = col("temp_outside"), col("temp_hall"), col("temp_lower_bound"), col("temp_upper_bound")
    r_out, r_hall, cap, air_max = col("R_outside"), col("R_hall"), col("capacity"), col("air_max")
    r_wall = _t(cfg["R_wall"], device)
    adj = torch.as_tensor(np.asarray(cfg["adj"], dtype=bool), device=device)
    adj = (adj | adj.T).float()
    adj_out = torch.as_tensor(np.asarray(cfg["adj_outside"], dtype=bool), device=device).reshape(-1).float()
    adj_hall = torch.as_tensor(np.asarray(cfg["adj_hall"], dtype=bool), device=device).reshape(-1).float()
    n = lo.numel()

    def transition(x, u):
        air = u * air_max
        heating = air * CAP_AIR * (TEMP_AIR - x)
        between = torch.sum(-adj / r_wall * (x[:, None] - x[None, :]), dim=-1)      # heat exchanged with the adjacent rooms
        outside = adj_out / r_out * (t_out - x)
        hall = adj_hall / r_hall * (t_hall - x)
        return x + TIME_DELTA / cap * (heating + between + outside + hall)

    def penalties(x):
        out_of_bounds = PENALTY * (torch.relu(lo - x) + torch.relu(x - hi))
        set_point = SET_POINT_PENALTY * torch.abs((lo + hi) / 2 - x)
        return out_of_bounds + set_point

    def cost(x, u):
        return torch.sum(COST_AIR * (u * air_max) + penalties(x))

    def final_cost(x):
        return torch.sum(penalties(x))

    return TorchEnv(transition, cost, final_cost, n, n, 0.0, 1.0, device=device)


def pendulum(device="cpu", dt=0.05, g_over_l=9.81, damping=0.1, torque=4.0):
    """An env that is none of the reference's: a damped pendulum with a torque limit, x = [angle, angular velocity], u = [torque] (the same model as
    tests/deviceenv_sources.py: PENDULUM) -- three plain torch functions, as a user would write them."""
    def transition(x, u):
        theta, omega = x[0], x[1]
        return torch.stack([theta + dt * omega, omega + dt * (u[0] - g_over_l * torch.sin(theta) - damping * omega)])

    def cost(x, u):
        return x[0] ** 2 + 0.1 * x[1] ** 2 + 0.01 * u[0] ** 2

    def final_cost(x):
        return 10.0 * (x[0] ** 2 + 0.1 * x[1] ** 2)

    return TorchEnv(transition, cost, final_cost, 2, 1, np.full((1, 1), -torque), np.full((1, 1), torque), device=device)


def bicycle(device="cpu", dt=0.1, wheelbase=2.5, goal=(8.0, 3.0, 0.4)):
    """Another env that is none of the reference's: a kinematic bicycle -- x = [px, py, heading, speed], u = [acceleration, steering angle] -- with a
    heading error wrapped through atan2: tan / atan2 / cos / sin, the vocabulary of a vehicle model."""
    gx, gy, gh = goal

    def transition(x, u):
        px, py, th, v = x[0], x[1], x[2], x[3]
        return torch.stack([px + dt * v * torch.cos(th), py + dt * v * torch.sin(th), th + dt * v / wheelbase * torch.tan(u[1]), v + dt * u[0]])

    def heading_error(th):
        return torch.atan2(torch.sin(th - gh), torch.cos(th - gh))

    def cost(x, u):
        return 0.1 * ((x[0] - gx) ** 2 + (x[1] - gy) ** 2) + 0.5 * heading_error(x[2]) ** 2 + 0.01 * x[3] ** 2 + 0.05 * u[0] ** 2 + 0.5 * u[1] ** 2

    def final_cost(x):
        return 5.0 * ((x[0] - gx) ** 2 + (x[1] - gy) ** 2) + 5.0 * heading_error(x[2]) ** 2 + x[3] ** 2

    return TorchEnv(transition, cost, final_cost, 4, 2, np.array([[-2.0], [-0.5]]), np.array([[2.0], [0.5]]), device=device)


def cartpole(device="cpu", dt=0.02, m_cart=1.0, m_pole=0.1, half_length=0.5, gravity=9.81, force=10.0):
    """The textbook cart-pole (Barto, Sutton & Anderson's equations): x = [position, velocity, angle from upright, angular velocity], u = [force]."""
    total, pml = m_cart + m_pole, m_pole * half_length

    def transition(x, u):
        pos, vel, th, om = x[0], x[1], x[2], x[3]
        s, c = torch.sin(th), torch.cos(th)
        temp = (u[0] + pml * om ** 2 * s) / total
        th_acc = (gravity * s - c * temp) / (half_length * (4.0 / 3.0 - m_pole * c ** 2 / total))
        acc = temp - pml * th_acc * c / total
        return torch.stack([pos + dt * vel, vel + dt * acc, th + dt * om, om + dt * th_acc])

    def cost(x, u):
        return 0.1 * x[0] ** 2 + 0.01 * x[1] ** 2 + 1.0 - torch.cos(x[2]) + 0.01 * x[3] ** 2 + 0.001 * u[0] ** 2

    def final_cost(x):
        return 10.0 * (x[0] ** 2 + 0.1 * x[1] ** 2 + 2.0 * (1.0 - torch.cos(x[2])) + 0.1 * x[3] ** 2)

    return TorchEnv(transition, cost, final_cost, 4, 1, np.full((1, 1), -force), np.full((1, 1), force), device=device)
