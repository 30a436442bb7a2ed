"""BASELINE config 5 / config 3 harness: full WaveEnv episodes (RandomDesignPolicy) sharded over the visible ranks.

  python tools/rollout.py --episodes 8 --actions 20                      # 1 GPU
  python -m torch.distributed.run --nproc-per-node N tools/rollout.py    # N GPUs, episodes sharded contiguously

Prints one JSON line on rank 0: aggregate Mcell-updates/s, per-episode reward and final scattered-energy."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=8)
    ap.add_argument("--actions", type=int, default=20)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--grid", type=int, default=700)
    ap.add_argument("--per-launch", type=int, default=0,
                    help="actions per device call (wv_set_design_sequence; 0 = one call per action, two in flight)")
    args = ap.parse_args()
    import torch
    import waves_jl_amd as w
    from waves_jl_amd import dist as wd
    rank, local_rank, world = wd.init()
    dev = local_rank if torch.cuda.device_count() > local_rank else 0
    ds = wd.broadcast_design_space(w.build_triple_ring_design_space() if rank == 0 else None)
    dim = w.TwoDim(15.0, args.grid)
    mine = list(wd.shard_episodes(args.episodes, world, rank))
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0)
    env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=args.steps, actions=args.actions, device=dev,
                    return_fields=False)
    import gc
    gc.collect()
    gc.freeze()   # (as bench.py does: a collection that walks torch's objects in mid-episode stalls the host past a job's length)
    wd.barrier()
    t0 = time.perf_counter()
    rows = []
    for e in mine:
        env.rng = np.random.default_rng(e)               # episode e is the same whichever rank runs it
        src.rng = np.random.default_rng(10_000 + e)
        pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(20_000 + e))
        ep = w.generate_episode(pol, env, per_launch=args.per_launch or None)
        y = np.stack(ep.y)                               # (actions, steps+1, 3)
        rows.append([e, float(y.sum()), float(y[-1, -1, 0]), float(y[-1, -1, 2])])
    torch.cuda.synchronize()
    wd.barrier()
    dt = wd.max_over_ranks(time.perf_counter() - t0)
    pad = np.full((max(len(wd.shard_episodes(args.episodes, world, r)) for r in range(world)), 4), np.nan, np.float32)
    pad[:len(rows)] = np.array(rows, np.float32).reshape(-1, 4)
    allrows = np.concatenate(wd.gather_signals(pad))
    if rank == 0:
        allrows = allrows[~np.isnan(allrows[:, 0])]
        cu = args.episodes * args.actions * args.steps * args.grid * args.grid
        print(json.dumps({"episodes": args.episodes, "actions": args.actions, "steps_per_action": args.steps,
                          "grid": args.grid, "n_gpus": world, "actions_per_launch": args.per_launch or 1, "seconds": round(dt, 3),
                          "Mcell_updates_per_s": round(cu / dt / 1e6, 1),
                          "episodes_table[id,reward,tot_energy_end,sc_energy_end]": allrows.round(4).tolist()}))
    wd.finalize()


if __name__ == "__main__":
    main()
