#!/usr/bin/env python3
"""SURVEY 8(d) CPU baseline (i)/(ii): the REFERENCE's search and this build's CPU port, both timed on the build container's
host cores on the same kind of work, so that the port's rate measured on the GPU box (bench.py cpu_baseline) can be related
to the reference.  Runs only where /root/reference exists; writes profiles/r02/cpu_ratio_container.json.

  reference : main.run_mcts (main.py:99-125), 60 simulations, the reference's own Network (seed 1261) at batch 1 and one
              torch thread per process (main.py:2), 8 processes side by side (the reference starts cpu_count()-2 actors, main.py:32)
  port      : oracle/mz_oracle.c tree search (one OpenMP call per simulation over all trees) + PyTorch CPU fp32 inference
              on the whole leaf batch, 8 threads -- exactly bench.py's cpu_baseline leg
Roots: positions after (g mod 60) random moves of game g, Dirichlet noise at the root, as in the bench.
"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
SIMS = 60


def ref_worker(args):
    wid, gids = args
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as G      # registers the in-process `ray` stub and imports the reference's main.py
    import numpy as np
    import torch
    M = G.M
    M.config.num_simulations = SIMS
    torch.manual_seed(1261)
    net = M.Network()
    net.eval()
    np.random.seed(1261 + wid)
    sims, secs = 0, 0.0
    for k, g in enumerate(gids):
        game = G.reach_position(g, g % 60)
        root = M.Node(0, game.to_play())
        with torch.no_grad():
            out0 = net.initial_inference(torch.from_numpy(game.make_image(game.environment.steps)))
            M.expand_node(root, game.legal_actions(), out0)
            M.add_exploration_noise(root)
            t0 = time.perf_counter()
            M.run_mcts(root, game.action_history, net)
            dt = time.perf_counter() - t0
        if k > 0:     # the first search of a process warms the allocator / thread pool
            sims += SIMS
            secs += dt
    return sims, secs


def port_leg(n, threads):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    from oracle import mz_oracle as O
    from muzero_amd.network import Network   # importing the package needs the built .so; the port itself never calls it
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    torch.manual_seed(1261)
    net = Network().eval()
    # roots by the oracle's own env: game g after g % 60 random moves
    to_play, legal, img = np.zeros(n, np.int8), np.zeros(n, np.uint64), np.zeros((n, 9, 8, 8), np.float32)
    for g in range(n):
        e = O.EnvBatch(1)
        hw, hb = [e.white[0]], [e.black[0]]
        for t in range(g % 60):
            a = O.lib().mzo_playout_action(1261, g, t, int(e.legal[0]))
            e.step(np.array([a], np.uint8))
            hw.append(e.white[0]); hb.append(e.black[0])
        k = g % 60
        pad = lambda h: np.array(h + [0] * (61 - len(h)), np.uint64)  # noqa: E731
        img[g] = O.make_image(pad(hw), pad(hb), k)
        to_play[g], legal[g] = e.to_play[0], e.legal[0]
    rng = np.random.default_rng(0)
    noise = np.zeros((n, 65))
    for g in range(n):
        acts = [a for a in range(64) if (int(legal[g]) >> a) & 1] or [64]
        noise[g, acts] = rng.dirichlet([0.03] * len(acts))
    tb = O.TreeBatch(n, SIMS)
    pool = np.zeros((SIMS + 1, n, 512), np.float32)
    x = np.zeros((n, 10, 8, 8), np.float32)
    xt = torch.from_numpy(x)
    t0 = time.perf_counter()
    with torch.no_grad():
        v, p, h = net.initial_inference(torch.from_numpy(img))
        pool[0] = h.reshape(n, 512).numpy()
        tb.root_init(to_play, legal, p.numpy(), noise)
        for s in range(SIMS):
            tb.select(pool, x)
            v, p, h = net.recurrent_inference_packed(xt)
            pool[s + 1] = h.reshape(n, 512).numpy()
            tb.expand_backup(v.numpy().reshape(-1), p.numpy())
    return n * SIMS / (time.perf_counter() - t0)


def main():
    procs = 8
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    jobs = [(w, [w * 1000 + i * 7 for i in range(per)]) for w in range(procs)]
    t0 = time.perf_counter()
    with mp.Pool(procs) as pool:
        res = pool.map(ref_worker, jobs)
    ref_rate = sum(s / t for s, t in res)
    print("reference: %.0f sims/s on %d processes (%.0f per core), %.0f s wall" % (ref_rate, procs, ref_rate / procs, time.perf_counter() - t0))
    port_rate = port_leg(1024, procs)
    print("port: %.0f sims/s on %d threads" % (port_rate, procs))
    out = {"reference_sims_per_sec": ref_rate, "port_sims_per_sec": port_rate, "cores": procs, "ratio_port_over_reference": port_rate / ref_rate,
           "workload": "%d-simulation searches, seed-1261 random-init network, roots after 0..59 random moves with Dirichlet noise; reference = "
                       "main.run_mcts at batch 1 in %d processes of 1 torch thread (main.py:2,32); port = C oracle tree search + PyTorch CPU "
                       "fp32 inference on a 1024-tree leaf batch, %d threads; build container, 8 cores" % (SIMS, procs, procs),
           "reference_searches_timed": sum(s for s, _ in res) // SIMS}
    with open(os.path.join(ROOT, "profiles", "r02", "cpu_ratio_container.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
