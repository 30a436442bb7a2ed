"""MI355X-native WaveEnv integrator: host-side mirror of the gladisor/Waves.jl operator API (TwoDim / WaveEnv /
Integrator) over libwaves_amd.so (hand-written HIP kernels for gfx950 behind the C ABI of include/waves_amd.h).

The directory is called `waves.jl_amd`; import it as `waves_jl_amd` (see waves_jl_amd.py at the repository root).
"""
from . import _ffi
from ._ffi import WavesAmdError, build, device_count
from .data import (Episode, design_from_dict, design_to_dict, flatten_repeated_last_dim, generate_episode,
                   prepare_data)
from .designs import (AIR, ALUMINIUM, BRASS, COPPER, WATER, AdjustablePositionScatterers, AdjustableRadiiScatterers, Cloak,
                      Cylinders, DesignInterpolator, DesignSpace, NoDesign, build_action_space,
                      build_radii_design_space, build_simple_radii_design_space, build_triple_ring_design_space, rand,
                      stack)
from .dims import TwoDim, build_dirichlet, build_grid, build_wave, get_dx, get_dy
from .dynamics import AcousticDynamics, Integrator, UniformSpeed, build_tspan, runge_kutta
from .env import (FRAMESKIP, RandomDesignPolicy, WaveEnv, WaveEnvState, action_space, is_terminated, reset, reward,
                  rollout_batched, rollout_pipelined, state, step_all)
from .latent import LatentIntegrator, LatentSource, LinearInterpolation, OneDim, compute_latent_energy
from .sources import NoSource, RandomPosGaussianSource, Source

__all__ = [n for n in dir() if not n.startswith("_")]
