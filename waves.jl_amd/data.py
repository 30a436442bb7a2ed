"""Episode rollouts -- host mirror of the reference's src/data.jl (the caller of the hot path): the rollout loop
(:12-33), the Episode container (:3-10), prepare_data windowing (:35-62, with flatten_repeated_last_dim of
src/utils.jl:20-35) and an on-disk episode format (:64-75 -- the reference writes BSON, which is Julia-specific: here
one .npz of plain arrays plus a JSON manifest inside it; nothing is pickled)."""
from __future__ import annotations

import json

import numpy as np

from .designs import (AdjustablePositionScatterers, AdjustableRadiiScatterers, Cloak, Cylinders, NoDesign)
from .dims import TwoDim


class Episode:
    """src/data.jl:3-10: states, actions, tspans, signals."""

    def __init__(self, s, a, t, y):
        self.s, self.a, self.t, self.y = s, a, t, y

    def __len__(self):
        return len(self.a)

    # ---- on-disk format (FileIO.save(episode, path) / Episode(path = ...)  src/data.jl:64-75)
    def save(self, path: str):
        """One .npz (numpy.load(..., allow_pickle=False) reads it): y (A, steps+1, 3), t (A, steps+1), the observations
        s_wave (A, rx, ry, 4) + their tspans and dim vectors when the episode carries states, and a JSON manifest with
        the designs (actions and state designs) as nested {type, arrays}."""
        man = {"format": "waves_amd.episode/1", "actions": [design_to_dict(a) for a in self.a],
               "has_states": len(self.s) > 0}
        arrs = {"y": np.stack(self.y).astype(np.float32), "t": np.stack(self.t).astype(np.float32)}
        if self.s:
            arrs["s_wave"] = np.stack([st.wave for st in self.s]).astype(np.float32)
            arrs["s_tspan"] = np.stack([st.tspan for st in self.s]).astype(np.float32)
            arrs["dim_x"] = np.asarray(self.s[0].dim.x, np.float32)
            arrs["dim_y"] = np.asarray(self.s[0].dim.y, np.float32)
            man["state_designs"] = [design_to_dict(st.design) for st in self.s]
        arrs["manifest"] = np.frombuffer(json.dumps(man).encode(), dtype=np.uint8)
        with open(path, "wb") as fh:
            np.savez(fh, **arrs)

    @classmethod
    def load(cls, path: str) -> "Episode":
        from .env import WaveEnvState
        z = np.load(path, allow_pickle=False)
        man = json.loads(bytes(z["manifest"]).decode())
        if man.get("format") != "waves_amd.episode/1":
            raise ValueError(f"{path}: not a waves_amd episode file")
        a = [design_from_dict(d) for d in man["actions"]]
        y = [np.array(v) for v in z["y"]]
        t = [np.array(v) for v in z["t"]]
        s = []
        if man["has_states"]:
            dim = TwoDim(np.array(z["dim_x"]), np.array(z["dim_y"]))
            for k, d in enumerate(man["state_designs"]):
                s.append(WaveEnvState(dim, np.array(z["s_tspan"][k]), np.asfortranarray(z["s_wave"][k]), design_from_dict(d)))
        return cls(s, a, t, y)


_DESIGN_TYPES = {"NoDesign": NoDesign, "Cylinders": Cylinders, "AdjustableRadiiScatterers": AdjustableRadiiScatterers,
                 "AdjustablePositionScatterers": AdjustablePositionScatterers, "Cloak": Cloak}


def design_to_dict(d) -> dict:
    """A design as plain JSON (float32 values survive the round trip exactly: repr of the float64 they widen to)."""
    if isinstance(d, NoDesign):
        return {"type": "NoDesign"}
    if isinstance(d, Cylinders):
        return {"type": "Cylinders", "pos": np.asarray(d.pos, np.float64).tolist(), "r": np.asarray(d.r, np.float64).tolist(),
                "c": np.asarray(d.c, np.float64).tolist()}
    if isinstance(d, Cloak):
        return {"type": "Cloak", "config": design_to_dict(d.config), "core": design_to_dict(d.core)}
    if isinstance(d, (AdjustableRadiiScatterers, AdjustablePositionScatterers)):
        return {"type": type(d).__name__, "cylinders": design_to_dict(d.cylinders)}
    raise TypeError(f"cannot serialise design of type {type(d).__name__}")


def design_from_dict(m: dict):
    t = m["type"]
    if t == "NoDesign":
        return NoDesign()
    if t == "Cylinders":
        return Cylinders(np.asarray(m["pos"], np.float32).reshape(-1, 2), np.asarray(m["r"], np.float32),
                         np.asarray(m["c"], np.float32))
    if t == "Cloak":
        return Cloak(design_from_dict(m["config"]), design_from_dict(m["core"]))
    if t in ("AdjustableRadiiScatterers", "AdjustablePositionScatterers"):
        return _DESIGN_TYPES[t](design_from_dict(m["cylinders"]))
    raise ValueError(f"unknown design type {t!r}")


def flatten_repeated_last_dim(x):
    """src/utils.jl:20-35.  x: (..., n, k) -- k consecutive segments of n samples whose first sample repeats the last one
    of the segment before.  Returns (..., n + (n-1)(k-1)): the first segment whole, then every later segment without its
    first sample, in order (Julia's column-major reshape: sample index fastest, then segment).  A list of such arrays is
    flattened one by one and `hcat`ed (the Vector{<:AbstractMatrix} method)."""
    if isinstance(x, (list, tuple)):  # hcat(flatten_repeated_last_dim.(x)...): a matrix flattens to a vector = one column
        cols = [flatten_repeated_last_dim(v) for v in x]
        return np.stack(cols, axis=1) if cols[0].ndim == 1 else np.concatenate(cols, axis=1)
    x = np.asarray(x, np.float32)
    first = x[..., :, 0]
    rest = x[..., 1:, 1:]
    rest = np.swapaxes(rest, -1, -2).reshape(rest.shape[:-2] + (-1,))
    return np.concatenate([first, rest], axis=-1)


def prepare_data(ep, horizon: int):
    """prepare_data(ep::Episode, horizon)  src/data.jl:35-57: every window of `horizon` consecutive actions becomes one
    sample: the state before the window, the window's actions, its time axis and its signal with the repeated boundary
    samples removed -- t: (steps*horizon + 1,), y: (steps*horizon + 1, 3).  A list of episodes is handled episode by
    episode and concatenated (:59-62)."""
    if isinstance(ep, (list, tuple)):
        parts = [prepare_data(e, horizon) for e in ep]
        return tuple(sum((list(p[k]) for p in parts), []) for k in range(4))
    s, a, t, y = [], [], [], []
    n = horizon - 1
    for i in range(len(ep) - n):
        b = i + n
        if ep.s:
            s.append(ep.s[i])
        a.append(list(ep.a[i:b + 1]))
        t.append(flatten_repeated_last_dim(np.stack(ep.t[i:b + 1], axis=1)))            # hcat -> (steps+1, h)
        sig = np.stack(ep.y[i:b + 1], axis=2)                                              # (steps+1, 3, h)
        sig = flatten_repeated_last_dim(np.transpose(sig, (1, 0, 2)))                      # (3, T)
        y.append(np.ascontiguousarray(sig.T))
    return s, a, t, y


def generate_episode(policy, env, *, reset: bool = True, with_states: bool = False, verbose: bool = False,
                     in_flight: int | None = None, per_launch: int | None = None) -> Episode:
    """generate_episode!(policy, env)  src/data.jl:12-33.  `with_states=True` records `state(env)` before every action
    like the reference does (the observation is resized on the device: 256 KB per action); the default skips it, which is
    all the energy-trace benchmarks need.  `in_flight=2` keeps two actions in flight (the host prepares action k+1 while
    action k runs): policy(env) for action k+1 is then called while action k is still pending, so it is only valid for
    policies that do not look at the wave state.  The default (`in_flight=None`) therefore is the reference's strictly
    sequential loop (src/data.jl:22-27) UNLESS the policy itself says `reads_state = False`, as RandomDesignPolicy (which
    only samples its action space, src/env.jl:151-157) does; `in_flight=1` forces the sequential loop.
    `per_launch=n` hands n actions at a time to ONE device call (WaveEnv.steps_begin, wv_set_design_sequence; again for
    policies that do not read the wave state): the launch start-up and the gap between launches are paid once per n
    actions, and with `with_states=True` the frames of every action are kept on the device, so that the episode holds
    the very states the plain loop records (state(env) in front of every action)."""
    s, a, t, y = [], [], [], []
    if reset:
        env.reset()
    if per_launch:
        keep = env.return_fields
        env.return_fields = False
        try:
            while not env.is_terminated():
                left = env.actions - env.time_step // env.integration_steps
                m = max(1, min(int(per_launch), left))
                if with_states:
                    s.append(env.state())
                acts = [policy(env) for _ in range(m)]
                a += acts
                t += env.steps_begin(acts, keep_frames=with_states)
                y += [np.array(v) for v in env.steps_end()]
                if with_states:
                    s += [env.state_after(k) for k in range(m - 1)]
                if verbose:
                    print(env.time_step)
        finally:
            env.return_fields = keep
        return Episode(s, a, t, y)
    keep = env.return_fields
    env.return_fields = False  # the rollout discards the returned fields (src/data.jl:27)
    if in_flight is None:
        in_flight = 2 if getattr(policy, "reads_state", True) is False else 1
    depth = 1 if with_states else max(1, min(2, int(in_flight)))
    pending = 0
    try:
        while not env.is_terminated():
            if with_states:
                s.append(env.state())
            action = policy(env)
            a.append(action)
            t.append(env.build_tspan())
            env.step_begin(action)
            pending += 1
            if pending >= depth:
                env.step_end()
                pending -= 1
                y.append(np.array(env.signal))
            if verbose:
                print(env.time_step)
        while pending:
            env.step_end()
            pending -= 1
            y.append(np.array(env.signal))
    finally:
        env.return_fields = keep
        while pending:   # (an exception in the policy or in a step: nothing stays pending in the env or the context)
            pending -= 1
            try:
                env.step_end()
            except Exception:
                if getattr(env, "_pending", None):
                    env._pending.clear()
                break
    return Episode(s, a, t, y)
