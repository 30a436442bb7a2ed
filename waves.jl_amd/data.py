"""Episode rollouts -- host mirror of the rollout loop of the reference's src/data.jl:3-33 (the caller of the hot path)."""
from __future__ import annotations

import numpy as np


class Episode:
    """src/data.jl:3-10: states, actions, tspans, signals."""

    def __init__(self, s, a, t, y):
        self.s, self.a, self.t, self.y = s, a, t, y

    def __len__(self):
        return len(self.a)


def generate_episode(policy, env, *, reset: bool = True, with_states: bool = False, verbose: bool = False) -> Episode:
    """generate_episode!(policy, env)  src/data.jl:12-33.  `with_states=False` skips the per-action `state(env)`
    download (an observation path the north star leaves for later: SURVEY 8f)."""
    s, a, t, y = [], [], [], []
    if reset:
        env.reset()
    keep = env.return_fields
    env.return_fields = False  # the rollout discards the returned fields (src/data.jl:27)
    try:
        while not env.is_terminated():
            if with_states:
                s.append(env.state())
            action = policy(env)
            a.append(action)
            t.append(env.build_tspan())
            env(action)
            y.append(np.array(env.signal))
            if verbose:
                print(env.time_step)
    finally:
        env.return_fields = keep
    return Episode(s, a, t, y)
