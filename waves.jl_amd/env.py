"""WaveEnv -- host mirror of the reference's src/env.jl: the RL "gym" around the device integrator."""
from __future__ import annotations

import numpy as np

from .designs import DesignInterpolator, DesignSpace, WATER, build_action_space, rand
from .dims import TwoDim, get_dx, get_dy
from .dynamics import AcousticDynamics, Integrator, build_tspan, runge_kutta
from .sources import NoSource

f32 = np.float32
FRAMESKIP = 10  # src/env.jl:90


class WaveEnvState:
    """src/env.jl:5-12."""

    def __init__(self, dim, tspan, wave, design):
        self.dim, self.tspan, self.wave, self.design = dim, tspan, wave, design


class WaveEnv:
    """src/env.jl:14-67.  Same constructor keywords and defaults; `device`, `impl` and `rng` are additions
    (device replaces `Flux.device!(n)`; rng replaces Julia's global RNG)."""

    def __init__(self, dim: TwoDim, *, design_space: DesignSpace, action_speed=250.0, source=None, c0=WATER,
                 pml_width=2.0, pml_scale=20000.0, resolution=(128, 128), dt=1e-5, integration_steps=100, actions=10,
                 device=0, impl="auto", rng=None, return_fields=True, trajectory_stride=1):
        if not all(s > r for s, r in zip(dim.size(), resolution)):  # src/env.jl:52
            raise AssertionError("Resolution must be less than finite element grid.")
        self.rng = rng if rng is not None else np.random.default_rng()
        self.dim = dim
        self.design_space = design_space
        self.design = rand(design_space, self.rng)                      # :55
        self.source = source if source is not None else NoSource()
        dyn = AcousticDynamics(dim, c0, pml_width, pml_scale, device=device, impl=impl)   # :58
        self.iter = Integrator(runge_kutta, dyn, dt)                    # :59
        self.ctx = self.iter.ctx                                        # env.wave lives on the device: zeros(nx,ny,12,3)
        self.source.attach(self.ctx)
        self.signal = np.zeros(integration_steps + 1, dtype=np.float32)  # :56
        self.time_step = 0
        self.resolution = tuple(resolution)
        self.action_speed = f32(action_speed)
        self.dt = f32(dt)
        self.integration_steps = int(integration_steps)
        self.actions = int(actions)
        self.return_fields = return_fields        # False / True (src/env.jl:120) / "stream" (planes streamed to pinned memory)
        self.trajectory_stride = int(trajectory_stride)   # u_tot / u_inc of env(action) hold every k-th saved time
        if self.trajectory_stride != 1:
            self.ctx.set_trajectory_stride(self.trajectory_stride)

    # --- src/env.jl:69-79
    def time(self):
        return f32(f32(self.time_step) * self.dt)

    def build_tspan(self):
        return build_tspan(self.time(), self.dt, self.integration_steps)

    def is_terminated(self):
        return self.time_step >= self.actions * self.integration_steps

    @property
    def wave(self):
        """env.wave (nx, ny, 12, 3), downloaded from the device."""
        return self.ctx.get_frames()

    @wave.setter
    def wave(self, w):
        self.ctx.set_frames(w)

    def reset(self):
        """RLBase.reset!  src/env.jl:81-88."""
        self.time_step = 0
        self.ctx.reset()                                    # env.wave *= 0f0
        self.design = rand(self.design_space, self.rng)
        self.signal = self.signal * f32(0.0)
        self.source.reset(self.rng)
        return None

    def __call__(self, action):
        """(env::WaveEnv)(action)  src/env.jl:91-121."""
        self.step_begin(action)
        return self.step_end()

    def step_begin(self, action):
        """First half of env(action): enqueue the whole action on the env's HIP stream and return.  Several envs on one
        GPU overlap this way (`for e in envs: e.step_begin(a)` then `for e in envs: e.step_end()`): while one env's
        tiles are in their load phase another's are computing (BASELINE config 3: 8 episodes per GPU).
        One env may also have TWO actions in flight (step_begin, step_begin, step_end, step_begin, step_end, ...): the
        host work of action k+1 (design algebra, coefficient tables, tile culling, uploads) then overlaps the device work
        of action k; only the wave state is sequentially dependent and the stream orders that.  The bookkeeping that
        does not depend on the device (env.design, env.time_step, src/env.jl:115,117) advances here; env.signal (:114)
        in step_end."""
        tspan = self.build_tspan()
        ti = self.time()
        current_design = self.design
        next_design = self.design_space(current_design, action)
        interp = DesignInterpolator(current_design, next_design, ti, tspan[-1])
        if self.integration_steps < 2 * FRAMESKIP:
            raise IndexError("BoundsError: sol[:, :, :, end-20:10:end] needs integration_steps >= 20 (src/env.jl:116)")
        self.ctx.set_design(*interp.abi_args())             # C = t -> speed(interp(t), grid, c0)
        self.ctx.integrate_begin(tspan, capture_frames=True, want_signal=True, want_fields=self.return_fields)
        if not hasattr(self, "_pending") or self._pending is None:
            self._pending = []
        self._pending.append((tspan, interp))
        self.design = next_design
        self.time_step += self.integration_steps
        # The NEXT action's tspan (a Julia Float32 range evaluated in twice precision: ~35 us of numpy on its first use) is
        # tabulated now, while the device is busy with this action; build_tspan memoises it.
        build_tspan(self.time(), self.dt, self.integration_steps)

    def steps_begin(self, actions, keep_frames=False):
        """`for a in actions; env(a); end` (src/data.jl:22-27) enqueued as ONE device call: valid when the actions do not
        depend on the wave state in between (RandomDesignPolicy, src/env.jl:151-157).  The launch start-up and the gap
        between two launches are paid once for all of them (wv_set_design_sequence).  env.design / env.time_step advance
        here as in step_begin; steps_end returns every action's signal; env.wave holds the frames of the LAST action.
        keep_frames=True keeps the frames of every action on the device (state_after(k): what state(env) returned after
        action k in the plain loop); such a call is not overlapped with another one."""
        if self.return_fields:
            raise ValueError("steps_begin: trajectories are per action -- use step_begin / step_end")
        if self.integration_steps < 2 * FRAMESKIP:
            raise IndexError("BoundsError: sol[:, :, :, end-20:10:end] needs integration_steps >= 20 (src/env.jl:116)")
        if keep_frames and self.integration_steps <= 2 * FRAMESKIP:
            # (the first kept frame of an action would be the state the action STARTS from, which no step of the one launch
            # writes; the per-action loop copies it -- wv_integrate_begin says the same for capture_frames == 2)
            raise ValueError("steps_begin(keep_frames=True) needs integration_steps > 20: use the per-action loop (step_begin / step_end)")
        tspans, interps, designs = [], [], [self.design]
        for action in actions:
            tspan = self.build_tspan()
            current_design = self.design
            next_design = self.design_space(current_design, action)
            interps.append(DesignInterpolator(current_design, next_design, self.time(), tspan[-1]))
            tspans.append(tspan)
            designs.append(next_design)
            self.design = next_design
            self.time_step += self.integration_steps
        stk = [d.stacked() for d in designs]
        if stk[0] is None:
            raise ValueError("steps_begin: NoDesign has nothing to move -- use step_begin")
        self.ctx.set_design_sequence([(c.pos, c.r, c.c) for c in stk], [(i.ti, i.tf) for i in interps], self.integration_steps)
        self.ctx.integrate_sequence_begin(np.stack(tspans), capture_frames="all" if keep_frames else True, want_signal=True)
        if not hasattr(self, "_pending") or self._pending is None:
            self._pending = []
        self._pending.append((tspans, interps))
        self._seq_designs = designs[1:] if keep_frames else None   # env.design after every action
        return tspans

    def state_after(self, k):
        """state(env) (src/env.jl:132-137) as the plain loop would have returned it after action k of the last
        steps_begin(..., keep_frames=True): the frames of that action, resized on the device, and the design then in force."""
        if not getattr(self, "_seq_designs", None):
            raise RuntimeError("state_after: the last steps_begin did not keep every action's frames")
        x = self.ctx.observation_action(k, *self.resolution)
        n = self.integration_steps
        t0 = f32(f32(self.time_step - (len(self._seq_designs) - 1 - k) * n) * self.dt)
        return WaveEnvState(self.dim, build_tspan(t0, self.dt, n), x, self._seq_designs[k])

    def steps_end(self):
        """Second half of steps_begin: the list of every action's env.signal (src/env.jl:114); env.signal is the last."""
        tspans, interps = self._pending.pop(0)
        sig, _, _ = self.ctx.integrate_end()
        n = self.integration_steps
        sigs = [sig[k * n:(k + 1) * n + 1].copy() for k in range(len(tspans))]
        self.signal = sigs[-1]
        return sigs

    def step_end(self):
        """Second half of env(action): wait for the device work of the oldest action in flight (src/env.jl:114,120)."""
        tspan, interp = self._pending.pop(0)
        sig, u_tot, u_inc = self.ctx.integrate_end()
        self.signal = sig                                   # hcat(tot_energy, inc_energy, sc_energy)
        return tspan, interp, u_tot, u_inc

    def state(self):
        """RLBase.state  src/env.jl:132-137: the three U_tot frames and the source shape, `imresize`d to
        `env.resolution` -- on the device (wv_observation; 3 x 64 KB leave the GPU instead of 7.8 MB).  imresize is
        Images.jl's (third-party, unpinned): the rule implemented is stated in csrc/kernels_aux.hip (k_observation)."""
        x = self.ctx.observation(*self.resolution)
        return WaveEnvState(self.dim, self.build_tspan(), x, self.design)

    def action_space(self):
        """RLBase.action_space  src/env.jl:143-145."""
        scale = f32(f32(self.action_speed * self.dt) * f32(self.integration_steps))
        return build_action_space(rand(self.design_space, self.rng), scale)

    def reward(self):
        """RLBase.reward  src/env.jl:147-149."""
        return np.sum(self.signal)


def rollout_pipelined(env, policy, n_actions):
    """`for _ in 1:n; env(policy(env)); end` with two actions in flight.  Valid for policies that do not look at the wave
    state (RandomDesignPolicy, src/env.jl:151-157: it only samples its action space); returns the list of env.signal."""
    sigs = []
    for k in range(n_actions):
        env.step_begin(policy(env))
        if k > 0:
            env.step_end()
            sigs.append(env.signal)
    if n_actions > 0:
        env.step_end()
        sigs.append(env.signal)
    return sigs


def rollout_batched(env, policy, n_actions, per_launch=None):
    """The same loop with `per_launch` actions per device call (default: all of them in one), two calls in flight.  For
    policies that do not look at the wave state; returns the list of env.signal, action by action."""
    per = n_actions if per_launch is None else max(1, int(per_launch))
    sigs, inflight = [], 0
    k = 0
    while k < n_actions:
        m = min(per, n_actions - k)
        env.steps_begin([policy(env) for _ in range(m)])
        inflight += 1
        k += m
        if inflight == 2:
            sigs += env.steps_end()
            inflight -= 1
    while inflight:
        sigs += env.steps_end()
        inflight -= 1
    return sigs


MAX_IN_FLIGHT = 4


def step_all(envs, actions, max_in_flight: int = MAX_IN_FLIGHT):
    """env(action) for several environments that live on one GPU, overlapped on their HIP streams: step_begin on a group,
    step_end on the group, next group.  At most `max_in_flight` actions are in flight at a time: the aggregate rate of
    700^2 environments on one MI355X peaks at 4 (48 Gcell-updates/s) and halves at 8, because every step kernel wants
    the whole device for one round of tiles.  Returns the list of step_end results."""
    out = []
    envs, actions = list(envs), list(actions)
    for g in range(0, len(envs), max_in_flight):
        grp = range(g, min(g + max_in_flight, len(envs)))
        for k in grp:
            envs[k].step_begin(actions[k])
        for k in grp:
            out.append(envs[k].step_end())
    return out


# RLBase-style free functions, as the reference's scripts call them
def reset(env):
    return env.reset()


def is_terminated(env):
    return env.is_terminated()


def state(env):
    return env.state()


def action_space(env):
    return env.action_space()


def reward(env):
    return env.reward()


class RandomDesignPolicy:
    """src/env.jl:151-157."""

    reads_state = False   # it only samples its action space: rollouts may keep two of its actions in flight (generate_episode)

    def __init__(self, a_space: DesignSpace, rng=None):
        self.a_space = a_space
        self.rng = rng if rng is not None else np.random.default_rng()

    def __call__(self, env):
        return rand(self.a_space, self.rng)
