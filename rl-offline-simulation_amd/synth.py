"""Synthetic logged-experience generators (host side, NumPy).

The GPU box has no datasets, gym or h5py, so bench.py, smoke() and the tests make their
logged transitions here.  Every generator returns the `OfflineDataset.experience` schema of
the reference (offsim4rl/data.py:46-58; written by offsim4rl/utils/dataset_utils.py:103-113):
observations, actions, action_distributions, rewards, next_observations, terminals, steps,
episode_ids -- plus, where the latent state is known in closed form, `z` / `z_next`.

Shapes and distributions follow SURVEY.md section 8(d).
"""
import numpy as np

__all__ = ["synth_iid", "cartpole_log", "grid_log", "grid_coords_log", "dirichlet_policy"]


def synth_iid(N, nS=162, nA=2, seed=20221107, p_done=1.0 / 50, p_init=1.0 / 50):
    """S-iid workload (headline / C4): iid states, softmax(N(0,1)) logging policy stored f32."""
    g = np.random.default_rng(seed)
    z = g.integers(0, nS, N, dtype=np.int64)
    z_next = g.integers(0, nS, N, dtype=np.int64)
    logits = g.standard_normal((N, nA))
    logits -= logits.max(axis=1, keepdims=True)
    p = np.exp(logits)
    p /= p.sum(axis=1, keepdims=True)
    p32 = p.astype(np.float32)
    # a ~ Categorical(p_log) by inverse CDF on the f64 probabilities
    u = g.random(N)
    a = (u[:, None] > np.cumsum(p, axis=1)).sum(axis=1).clip(0, nA - 1).astype(np.int64)
    r = g.random(N).astype(np.float32)
    done = g.random(N) < p_done
    t0 = g.random(N) < p_init
    if N > 0:
        t0[0] = True
    return dict(
        observations=z, next_observations=z_next, z=z, z_next=z_next, actions=a, action_distributions=p32,
        rewards=r, terminals=done, steps=np.where(t0, 0, 1).astype(np.int64),
        episode_ids=np.cumsum(t0).astype(np.int64) - 1)


def dirichlet_policy(n_rows, nA, seed=9):
    """Target policy of section 8(d): default_rng(9).dirichlet(1_nA, n_rows), f64."""
    return np.random.default_rng(seed).dirichlet(np.ones(nA), n_rows)


# ---------------------------------------------------------------------------------------
# CartPole-v1 dynamics (public equations: Barto, Sutton & Anderson 1983; Euler, tau = 0.02)
# ---------------------------------------------------------------------------------------
_G, _MC, _MP, _L, _F, _TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
_TH_LIM, _X_LIM = 12 * 2 * np.pi / 360, 2.4


def cartpole_log(N, seed=0, n_envs=None, max_steps=500, p_left=0.5):
    """Uniform-random logging policy on CartPole dynamics (configs C1/C2).

    `n_envs` independent environments are stepped in lock-step (vectorised); rows are emitted
    env-major, so each env contributes a run of consecutive complete-or-truncated episodes.
    Observations are float32, as gym returns them.
    """
    if n_envs is None:
        n_envs = int(min(4096, max(1, N // 64)))
    T = -(-N // n_envs)
    g = np.random.default_rng(seed)
    s = g.uniform(-0.05, 0.05, (n_envs, 4))
    step = np.zeros(n_envs, np.int64)
    obs = np.empty((T, n_envs, 4), np.float32)
    nobs = np.empty((T, n_envs, 4), np.float32)
    act = np.empty((T, n_envs), np.int64)
    term = np.empty((T, n_envs), bool)
    trunc = np.empty((T, n_envs), bool)
    steps = np.empty((T, n_envs), np.int64)
    pm = _MP * _L
    tm = _MC + _MP
    for t in range(T):
        obs[t] = s
        steps[t] = step
        a = (g.random(n_envs) >= p_left).astype(np.int64)
        act[t] = a
        x, xd, th, thd = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
        f = np.where(a == 1, _F, -_F)
        ct, st = np.cos(th), np.sin(th)
        tmp = (f + pm * thd * thd * st) / tm
        tha = (_G * st - ct * tmp) / (_L * (4.0 / 3.0 - _MP * ct * ct / tm))
        xa = tmp - pm * tha * ct / tm
        s = np.stack([x + _TAU * xd, xd + _TAU * xa, th + _TAU * thd, thd + _TAU * tha], axis=1)
        nobs[t] = s
        step = step + 1
        d = (s[:, 0] < -_X_LIM) | (s[:, 0] > _X_LIM) | (s[:, 2] < -_TH_LIM) | (s[:, 2] > _TH_LIM)
        tr = (~d) & (step >= max_steps)
        term[t] = d
        trunc[t] = tr
        rs = d | tr
        if rs.any():
            s[rs] = g.uniform(-0.05, 0.05, (int(rs.sum()), 4))
            step[rs] = 0

    def em(x):  # env-major flatten, cut to N rows
        return np.ascontiguousarray(np.swapaxes(x, 0, 1)).reshape((n_envs * T,) + x.shape[2:])[:N]

    steps_f = em(steps)
    return dict(
        observations=em(obs), next_observations=em(nobs), actions=em(act),
        action_distributions=np.tile(np.array([p_left, 1 - p_left], np.float32), (N, 1)),
        rewards=np.ones(N, np.float32), terminals=em(term), truncateds=em(trunc), steps=steps_f,
        episode_ids=np.cumsum(steps_f == 0).astype(np.int64) - 1)


# ---------------------------------------------------------------------------------------
# 5x5 grid world of the reference's tests (dynamics: offsim4rl/envs/gridworld.py:72-124)
# ---------------------------------------------------------------------------------------
def _grid_rollout(n_episodes, num_cells, num_steps, goal, seed, coords):
    g = np.random.default_rng(seed)
    rows = []
    nA = 5
    for ep in range(n_episodes):
        x, y, cnt = 0, 0, 0
        done = False
        d0 = g.uniform(-0.1, 0.1, 2)
        while not done:
            a = int(g.integers(0, nA))
            was_goal = (x == goal[0] and y == goal[1])
            ox, oy = x, y
            if a == 1:
                y = min(y + 1, num_cells - 1)
            elif a == 2:
                x = min(x + 1, num_cells - 1)
            elif a == 3:
                y = max(y - 1, 0)
            elif a == 4:
                x = max(x - 1, 0)
            cnt += 1
            done = was_goal or cnt >= num_steps
            if done:
                r = 0.0
            elif x == goal[0] and y == goal[1]:
                r, done = 1.0, True
            else:
                r = -0.1
            d1 = g.uniform(-0.1, 0.1, 2)
            rows.append((ep, cnt - 1, ox, oy, a, r, x, y, done, d0[0], d0[1], d1[0], d1[1]))
            d0 = d1
    R = np.array(rows, np.float64).reshape(-1, 13)
    z = (R[:, 2] + num_cells * R[:, 3]).astype(np.int64)
    zn = (R[:, 6] + num_cells * R[:, 7]).astype(np.int64)
    out = dict(
        z=z, z_next=zn, actions=R[:, 4].astype(np.int64), rewards=R[:, 5].copy(), terminals=R[:, 8] != 0,
        steps=R[:, 1].astype(np.int64), episode_ids=R[:, 0].astype(np.int64),
        action_distributions=np.full((R.shape[0], nA), 0.2, np.float64))
    if coords:  # MyGridNaviCoords.external_state  (gridworld.py:205-211)
        out["observations"] = np.stack([R[:, 2] / 5 + 0.1 + R[:, 9], R[:, 3] / 5 + 0.1 + R[:, 10]], 1).astype(np.float32)
        out["next_observations"] = np.stack([R[:, 6] / 5 + 0.1 + R[:, 11], R[:, 7] / 5 + 0.1 + R[:, 12]], 1).astype(np.float32)
    else:
        out["observations"], out["next_observations"] = z.copy(), zn.copy()
    return out


def grid_log(n_episodes=10, num_cells=5, num_steps=10, goal=(4, 4), seed=0):
    """Discrete-observation grid log shaped like tests/test_psrs.py:18-23 of the reference."""
    return _grid_rollout(n_episodes, num_cells, num_steps, goal, seed, coords=False)


def grid_coords_log(n_episodes, num_cells=5, num_steps=15, goal=(4, 4), seed=0):
    """Continuous-observation grid log (config C3; examples/continuous_grid/random_agent_rollout.py:62-84)."""
    return _grid_rollout(n_episodes, num_cells, num_steps, goal, seed, coords=True)


def grid_coords_log_fast(N, num_cells=5, num_steps=15, goal=(4, 4), seed=0, n_envs=4096):
    """Vectorised variant of grid_coords_log for multi-million-row logs (env-major rows)."""
    T = -(-N // n_envs)
    g = np.random.default_rng(seed)
    x = np.zeros(n_envs, np.int64)
    y = np.zeros(n_envs, np.int64)
    cnt = np.zeros(n_envs, np.int64)
    d0 = g.uniform(-0.1, 0.1, (n_envs, 2))
    O = np.empty((T, n_envs, 2), np.float32)
    NO = np.empty((T, n_envs, 2), np.float32)
    Z = np.empty((T, n_envs), np.int64)
    ZN = np.empty((T, n_envs), np.int64)
    A = np.empty((T, n_envs), np.int64)
    Rw = np.empty((T, n_envs), np.float64)
    D = np.empty((T, n_envs), bool)
    S = np.empty((T, n_envs), np.int64)
    for t in range(T):
        a = g.integers(0, 5, n_envs)
        was_goal = (x == goal[0]) & (y == goal[1])
        O[t, :, 0] = x / 5 + 0.1 + d0[:, 0]
        O[t, :, 1] = y / 5 + 0.1 + d0[:, 1]
        Z[t] = x + num_cells * y
        S[t] = cnt
        A[t] = a
        y = np.where(a == 1, np.minimum(y + 1, num_cells - 1), y)
        x = np.where(a == 2, np.minimum(x + 1, num_cells - 1), x)
        y = np.where(a == 3, np.maximum(y - 1, 0), y)
        x = np.where(a == 4, np.maximum(x - 1, 0), x)
        cnt = cnt + 1
        done = was_goal | (cnt >= num_steps)
        at_goal = (x == goal[0]) & (y == goal[1])
        r = np.where(done, 0.0, np.where(at_goal, 1.0, -0.1))
        done = done | at_goal
        d1 = g.uniform(-0.1, 0.1, (n_envs, 2))
        NO[t, :, 0] = x / 5 + 0.1 + d1[:, 0]
        NO[t, :, 1] = y / 5 + 0.1 + d1[:, 1]
        ZN[t] = x + num_cells * y
        Rw[t] = r
        D[t] = done
        d0 = d1
        x = np.where(done, 0, x)
        y = np.where(done, 0, y)
        cnt = np.where(done, 0, cnt)
        if done.any():
            d0[done] = g.uniform(-0.1, 0.1, (int(done.sum()), 2))

    def em(v):
        return np.ascontiguousarray(np.swapaxes(v, 0, 1)).reshape((n_envs * T,) + v.shape[2:])[:N]

    steps = em(S)
    return dict(observations=em(O), next_observations=em(NO), z=em(Z), z_next=em(ZN), actions=em(A),
                rewards=em(Rw), terminals=em(D), steps=steps, episode_ids=np.cumsum(steps == 0) - 1,
                action_distributions=np.full((N, 5), 0.2, np.float32))


def grid_cell_encoder_weights(num_cells=5, hidden=64, seed=0, sharpness=200.0):
    """Weights of a 2-`hidden`-num_cells^2 MLP (Linear, LeakyReLU, Linear) that sends a continuous_grid observation to its cell:
    no trained HOMER checkpoint travels, and nn.Linear's default init leaves most of the 25 abstract states empty.  Hidden units 0
    and 1 copy the (non-negative) coordinates; the readout scores cell k by sharpness * (2 c_k . x - |c_k|^2), the nearest cell centre
    c_k = ((i + 0.5) / num_cells, (j + 0.5) / num_cells), k = i + num_cells * j (ContinuousGridEnv's state index,
    continuous_grid.py:62-66); every other weight keeps nn.Linear's default init, so the product is still a dense 2-64-25 MLP whose
    argmax differs from the true cell only within the noise of those units next to a cell boundary.  Returns (W1, b1, W2, b2), f32."""
    g = np.random.default_rng(seed)
    nz = num_cells * num_cells
    lin = lambda o, i: (((g.random((o, i)) * 2 - 1) / i ** 0.5).astype(np.float32), ((g.random(o) * 2 - 1) / i ** 0.5).astype(np.float32))
    (W1, b1), (W2, b2) = lin(hidden, 2), lin(nz, hidden)
    W1[0], W1[1], b1[0], b1[1] = (1.0, 0.0), (0.0, 1.0), 0.0, 0.0
    k = np.arange(nz)
    c = np.stack([(k % num_cells + 0.5) / num_cells, (k // num_cells + 0.5) / num_cells], 1)
    W2[:, 0:2] = (sharpness * 2.0 * c).astype(np.float32)
    b2 += (-sharpness * (c * c).sum(1)).astype(np.float32)
    return W1, b1, W2, b2
