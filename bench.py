#!/usr/bin/env python3
"""bench.py -- MCTS simulations/s (+ env steps/s, self-play games/s) on N MI355X GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): 4096 game trees per GPU x 50 simulations, random-init MuZero
network (seed 1261), bf16 inference once per simulation on the whole leaf batch, roots =
positions reached after 0..59 random moves.  One "step" = one complete search of every tree on the
rank: initial inference + root expansion with Dirichlet noise + 50 x (recurrent inference + fused
expand/backup/select kernel) + root statistics.  Inputs are resident in HBM before the timed region.
Games shard across ranks with no data-path collective (weak scaling: 4096 trees per GPU).

Extra legs, reported in the same JSON line and NOT part of `value`:
  env        configs[1]: 1,048,576 random-play games x 60 steps; fused in-register playout kernel
             and the HBM-resident step kernel (roofline_env is for the latter)
  selfplay   --selfplay-games G: G whole games of 60 moves through the batched engine -> games/s
  roofline   dominant kernel of the timed region: k_tower (conv tower + heads, MFMA bound), HIP-event timed;
             roofline_tree = k_backup_select and roofline_env = k_env_step (both HBM bound)
  cpu_baseline  the C oracle (tree search) + PyTorch CPU inference on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA
F32_PEAK_TFLOPS = 157.3
NET_MFLOP = 116.0            # MFLOP per inference sample (rounded)
# exact algorithmic FLOPs of one recurrent_inference sample (network.py:113-135, 79-111), real channel counts:
# conv 10->128 + 6 x conv 128->128 + conv 128->8 (3x3 on 64 cells, 2 FLOP per MAC) + heads (2 x 1x1 conv 8->2, 128x128, 128x65, 128x1)
TOWER_FLOP = 2 * 64 * 9 * (10 * 128 + 6 * 128 * 128 + 128 * 8) + 2 * (2 * 2 * 8 * 64 + 128 * 128 + 128 * 65 + 128)
ENV_BYTES_PER_STEP = 46


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trees", type=int, default=4096, help="game trees per GPU")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tower", default="auto", choices=["auto", "hip", "miopen"], help="inference path (auto = hip for bf16)")
    ap.add_argument("--env-games", type=int, default=1 << 20)
    ap.add_argument("--env-games-hbm", type=int, default=1 << 24, help="second env point whose state exceeds the caches (0 = skip)")
    ap.add_argument("--fp32-steps", type=int, default=2, help="searches of the fp32-inference companion leg (0 = skip)")
    ap.add_argument("--selfplay-games", type=int, default=4096)
    ap.add_argument("--selfplay-sims", type=int, default=50)
    ap.add_argument("--skip-extras", action="store_true", help="only the timed MCTS leg")
    ap.add_argument("--cpu-trees", type=int, default=4096)
    ap.add_argument("--train-batch", type=int, default=512, help="config.py:34 batch_size for the training leg (0 = skip)")
    ap.add_argument("--train-unroll", type=int, default=5, help="K (BASELINE configs[4]: 5; config.py:36: 3)")
    ap.add_argument("--config4-searches", type=int, default=3, help="searches of the 8192 x 100 leg (0 = skip)")
    ap.add_argument("--loop-train-steps", type=int, default=20, help="training steps of the whole-loop leg (one generation)")
    return ap.parse_args()


def make_roots(n, seed, game0, device):
    """positions after (g mod 60) random moves of game g, with their observation history"""
    from muzero_amd.env import BatchedEnv
    e = BatchedEnv(n, device)
    hw = torch.zeros((61, n), dtype=torch.int64, device=device)
    hb = torch.zeros((61, n), dtype=torch.int64, device=device)
    hw[0], hb[0] = e.white, e.black
    act = torch.empty(n, dtype=torch.uint8, device=device)
    for t in range(60):
        e.random_action(seed, game0, t, out=act)
        e.step(act)
        hw[t + 1], hb[t + 1] = e.white, e.black
    idx = (torch.arange(n, device=device) % 60).to(torch.int32)
    gi = torch.arange(n, device=device)
    r = BatchedEnv(n, device)
    r.white.copy_(hw[idx.long(), gi])
    r.black.copy_(hb[idx.long(), gi])
    r.to_play.copy_(torch.where(idx % 2 == 0, -1, 1).to(torch.int8))  # every step switches the mover
    r.steps.copy_(idx.to(torch.uint8))
    r.legal.copy_(r.legal_mask())
    return r, hw, hb, idx


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/README.md): FETCH_SIZE
    (doubled, MI355X_MICROARCH.md gfx950 correction) + WRITE_SIZE at this bench's launch shape.
    rocprofv3 cannot run inside this process, so the number is read from the newest committed profile of this
    same command (profiles/rNN/pmc_summary.json, regenerated every round); None when no profile is present."""
    return pmc_traffic_src(kernel)[0]


def newest_profile(name):
    """path of the newest committed profiles/rNN/<name>, or None"""
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, r, name)
        if os.path.exists(f):
            return f
    return None


def pmc_traffic_src(kernel, name="pmc_summary.json"):
    """(bytes per launch, the committed file they were read from) -- the number is NOT measured in this run"""
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, r, name)
        if os.path.exists(f):
            rec = json.load(open(f)).get(kernel)
            if rec and rec.get("hbm_bytes_per_launch"):
                return rec["hbm_bytes_per_launch"], "profiles/%s/%s (rocprofv3 --pmc passes of %s, committed; not measured in this run)" % (
                    r, name, "this command" if name == "pmc_summary.json" else "tools/x3_engine_prof.py")
    return None, None


def event_time_ms(fn, iters):
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(min(10, iters)):   # untimed: the launch sequence at the clock the chip holds under it
        fn()
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def kernel_isa(kernel):
    """static VALU census of the kernel's gfx950 ISA, generated from the CURRENT sources by __graft_entry__.build()
    (tools/isa_stats.py -> muzero_amd/lib/kernel_isa.json); None when the file is absent (never a literal)"""
    f = os.path.join(ROOT, "muzero_amd", "lib", "kernel_isa.json")
    return json.load(open(f)).get(kernel) if os.path.exists(f) else None


def bench_env(args, device):
    from muzero_amd.env import BatchedEnv
    n = args.env_games
    e = BatchedEnv(n, device)
    e.random_playout(1261, 0, 60, 0)
    torch.cuda.synchronize()
    ms_fused = event_time_ms(lambda: e.random_playout(1261, 0, 60, 0), 5)

    def step_kernel_ms(e, n, game0, reps):
        """HBM-resident path: one k_env_step launch per step over states in HBM (actions precomputed)"""
        e.reset()
        acts = torch.empty((60, n), dtype=torch.uint8, device=device)
        for t in range(60):
            e.random_action(1261, game0, t, out=acts[t])
            e.step(acts[t])
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(reps):
            e.reset()
            start.record()
            for t in range(60):
                e.step(acts[t])
            end.record()
            end.synchronize()
            tot += start.elapsed_time(end)
        return tot / (reps * 60)

    def packed_step_ms(e, n, game0, reps):
        """the same 60 steps through mz_env_step_packed.  The action travels inside the per-game u32 word, so between two
        steps one torch op patches the next action into the words; it sits OUTSIDE the timed region: every launch is timed
        between its own pair of events"""
        from muzero_amd import _lib as L
        e.reset()
        acts = torch.empty((60, n), dtype=torch.uint8, device=device)
        for t in range(60):
            e.random_action(1261, game0, t, out=acts[t])
            e.step(acts[t])
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(reps):
            e.reset()
            meta = e.pack_meta(acts[0])
            for t in range(60):
                start.record()
                L.call("mz_env_step_packed", L.dptr(e.black), L.dptr(e.white), L.dptr(e.legal), L.dptr(meta), n, L.current_stream())
                end.record()
                end.synchronize()
                tot += start.elapsed_time(end)
                if t + 1 < 60:
                    meta = ((meta & 0x1FF) | (acts[t + 1].to(torch.int32) << 9)).contiguous()
        return tot / (reps * 60)

    ms_step = step_kernel_ms(e, n, 0, 3)
    gbps = n * ENV_BYTES_PER_STEP / (ms_step * 1e-3) / 1e9
    del e
    # the same kernel on a state that cannot sit in the caches: 16 M games = 336 MB of state arrays + 16 MB of actions per
    # step against 256 MB of Infinity Cache + 32 MB of L2, so THIS fraction is an HBM fraction
    big = args.env_games_hbm
    roof_big = None
    if big > 0:
        eb = BatchedEnv(big, device)
        ms_big = step_kernel_ms(eb, big, 1 << 24, 1)
        g2 = big * ENV_BYTES_PER_STEP / (ms_big * 1e-3) / 1e9
        roof_big = {"kernel": "k_env_step", "games": big, "bound": "hbm", "achieved": g2, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": g2 / HBM_PEAK_GBPS, "traffic": pmc_traffic("k_env_step_hbm"), "traffic_source": pmc_traffic_src("k_env_step_hbm")[1], "algorithmic_bytes_per_launch": big * ENV_BYTES_PER_STEP,
                    "us_per_launch": ms_big * 1e3, "env_steps_per_sec": big / (ms_big * 1e-3),
                    "working_set_bytes": big * 21, "note": "state exceeds L2 + Infinity Cache: served from HBM"}
        # the packed record (mz_env_step_packed): one u32 word instead of six byte-wide streams; 20 B in + 28 B out per step
        ms_pk = packed_step_ms(eb, big, 1 << 24, 1)
        gp = big * ENV_BYTES_PER_STEP / (ms_pk * 1e-3) / 1e9
        roof_big["packed"] = {"kernel": "k_env_step_packed", "achieved": gp, "frac": gp / HBM_PEAK_GBPS, "us_per_launch": ms_pk * 1e3,
                              "env_steps_per_sec": big / (ms_pk * 1e-3), "bytes_moved_per_step": 48,
                              "note": "same algorithmic 46 B per step as the line above; each launch timed between its own events"}
        del eb
    isa = kernel_isa("k_env_step")
    valu = None
    if isa:   # the env's floor is its integer ALU work, so the VALU issue rate is reported beside the GB/s
        waves = n / 64
        valu = {"instructions_per_wave_step": isa["valu_instructions"], "half_rate_64bit_shifts": isa["half_rate_64bit_shifts"],
                "issue_cycles_per_wave_step": isa["issue_cycles_per_wave"], "source": "muzero_amd/lib/kernel_isa.json (build-time census of the gfx950 ISA)",
                "wave_instructions_per_sec": waves * isa["valu_instructions"] / (ms_step * 1e-3),
                "busy_frac_at_2.4GHz": waves * isa["issue_cycles_per_wave"] / 1024 / 2.4e9 / (ms_step * 1e-3)}
    return {
        "env_games": n,
        "env_steps_per_sec_fused_playout": n * 60 / (ms_fused * 1e-3),
        "env_steps_per_sec_step_kernel": n / (ms_step * 1e-3),
        "roofline_env": {"kernel": "k_env_step", "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": gbps / HBM_PEAK_GBPS, "traffic": pmc_traffic("k_env_step"), "traffic_source": pmc_traffic_src("k_env_step")[1], "bytes_per_step": ENV_BYTES_PER_STEP,
                         "algorithmic_bytes_per_launch": n * ENV_BYTES_PER_STEP, "us_per_launch": ms_step * 1e3,
                         "note": "nominal: the 1,048,576-game state (22 MB + actions) stays resident in L2 + Infinity Cache, so this is "
                                 "algorithmic bytes over time, not HBM traffic; the kernel is VALU-issue bound (valu) -- roofline_env_hbm is the HBM point",
                         "valu": valu},
        "roofline_env_hbm": roof_big,
    }


def bench_train(args, device, rec):
    """replay window -> sample_batch (HIP) -> update_weights (PyTorch fp32, flat gradient bucket)"""
    import random
    from muzero_amd import config, train as T
    from muzero_amd.network import Network
    from muzero_amd.replay import ReplayBuffer
    B, K = args.train_batch, args.train_unroll
    config.num_unroll_steps = K
    T.LOG = None   # stdout carries the one JSON line only
    rb = ReplayBuffer(device, window_size=max(rec.n, B), batch_size=B)
    while rb.buffer_size < B:
        rb.save_games(rec)
    random.seed(1261)
    torch.manual_seed(1261)
    net = Network().to(device)
    net.train()
    opt = T.make_optimizer(net)
    bucket = T.FlatGradBucket(net.parameters())
    net.steps = 1  # no loss print inside the timed loop
    for _ in range(3):
        T.update_weights(opt, net, rb.sample_batch(K), bucket)
    torch.cuda.synchronize()
    ms_sample = event_time_ms(lambda: rb.sample_batch(K), 20)
    t0 = time.perf_counter()
    iters = 10
    for _ in range(iters):
        p_loss, v_loss = T.update_weights(opt, net, rb.sample_batch(K), bucket)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    t0 = time.perf_counter()
    for _ in range(3 + iters):
        if _ == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        T.update_weights(opt, net, rb.sample_batch(K), bucket, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    dt_amp = (time.perf_counter() - t0) / iters
    # arithmetic of one step: K + 1 network passes over B boards (one Representation, K Dynamics: the same tower FLOPs), each
    # forward + gradient w.r.t. its input + gradient w.r.t. its weights = 3 x the forward FLOPs (heads and loss are < 1 %)
    flop = 3.0 * (K + 1) * B * TOWER_FLOP

    def hip_legs():
        # the towers' 128 -> 128 convolutions (forward, data gradient, weight gradient) on the hand-written matrix-core kernels
        for _ in range(3 + iters):
            if _ == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            T.update_weights(opt, net, rb.sample_batch(K), bucket, hip_convs=True)
        torch.cuda.synchronize()
        dt_hip = (time.perf_counter() - t0) / iters
        # the same kernels in split precision (three bf16 MFMA products per fp32 product: ~1e-5 of the fp32 arithmetic)
        for _ in range(3 + iters):
            if _ == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            T.update_weights(opt, net, rb.sample_batch(K), bucket, hip_convs="x3")
        torch.cuda.synchronize()
        dt_x3 = (time.perf_counter() - t0) / iters
        # the two hand-written training kernels alone, at the step's own sizes (HIP events over 30 launches each, torch's current
        # stream = the stream they are launched on): one convolution of one use (B boards), one weight gradient over the K uses
        from muzero_amd import convgrad as CG
        gen = torch.Generator(device=device).manual_seed(7)
        wconv = torch.randn(128, 128, 3, 3, device=device, generator=gen) * 0.03
        xs = [torch.randn(B, 128, 8, 8, device=device, generator=gen).relu() for _ in range(K)]
        gs = [torch.randn(B, 128, 8, 8, device=device, generator=gen) for _ in range(K)]
        pf, _pb = CG.pack_conv(wconv)
        yb, dwb, dbb = torch.empty_like(xs[0]), torch.empty_like(wconv), torch.empty(128, device=device)
        us_conv = event_time_ms(lambda: CG.conv3x3(xs[0], pf, out=yb), 30) * 1e3
        us_wgrad = event_time_ms(lambda: CG.conv3x3_wgrad(xs, gs, dwb, dbb), 30) * 1e3
        conv_flop = 2.0 * B * 64 * 128 * 128 * 9
        hip_kernels = {
            "k_conv3x3": {"boards": B, "us_per_launch": us_conv, "tflops": conv_flop / us_conv / 1e6, "frac_of_bf16_peak": conv_flop / us_conv / 1e6 / BF16_PEAK_TFLOPS,
                          "hbm_gbps_fp32_io": B * 65536 / us_conv / 1e3, "frac_of_hbm_peak": B * 65536 / us_conv / 1e3 / HBM_PEAK_GBPS,
                          "note": "forward / data gradient of one 128->128 convolution, fp32 NCHW in and out (64 KB per board = 295 FLOP/B: as HBM-bound as "
                                  "MFMA-bound); at this batch 256 workgroups of 2 boards (4- and 8-board tiles from 1024 / 2048 boards on), one tile each: load, K loop and store phases do not overlap"},
            "k_conv_wgrad+k_wgrad_reduce": {"boards": B * K, "us_per_launch": us_wgrad, "tflops": K * conv_flop / us_wgrad / 1e6,
                                            "frac_of_bf16_peak": K * conv_flop / us_wgrad / 1e6 / BF16_PEAK_TFLOPS,
                                            "hbm_gbps_fp32_operands": B * K * 98304 / us_wgrad / 1e3, "frac_of_hbm_peak": B * K * 98304 / us_wgrad / 1e3 / HBM_PEAK_GBPS,
                                            "note": "weight + bias gradient over the K uses, fp32 NCHW operands by LDS-DMA (each co-half workgroup pulls 48 KB per "
                                                    "board: 197 FLOP/B, HBM-bound), slice partial sums (75 MB written and re-read) reduced in a fixed order"}}
        del xs, gs
        return {"hip_convs_ms_per_step": dt_hip * 1e3, "hip_convs_samples_per_sec": B / dt_hip, "hip_convs_tflops": flop / dt_hip / 1e12,
                "hip_convs_frac_of_bf16_peak": flop / dt_hip / 1e12 / BF16_PEAK_TFLOPS,
                "hip_kernels": hip_kernels, "hip_x3_ms_per_step": dt_x3 * 1e3, "hip_x3_samples_per_sec": B / dt_x3}

    try:
        hip = hip_legs()
    except Exception as e:   # noqa: BLE001  (optional legs: reported, not fatal to the fp32 numbers)
        hip = {"hip_convs_error": "%s: %s" % (type(e).__name__, e)}
    config.num_unroll_steps = 3
    res = {"batch": B, "unroll_steps": K, "dtype": "fp32", "ms_per_step": dt * 1e3, "samples_per_sec": B / dt,
            "sample_batch_ms": ms_sample, "p_loss": float(p_loss), "v_loss": float(v_loss),
            "bf16_autocast_ms_per_step": dt_amp * 1e3, "algorithmic_flop_per_step": flop,
            "tflops": flop / dt / 1e12, "frac_of_fp32_matrix_peak": flop / dt / 1e12 / F32_PEAK_TFLOPS,
            "bf16_autocast_tflops": flop / dt_amp / 1e12, "bf16_autocast_frac_of_bf16_peak": flop / dt_amp / 1e12 / BF16_PEAK_TFLOPS,
            "note": "ms_per_step: MIOpen convolutions through PyTorch autograd, fp32 as the reference; bf16_autocast: torch.autocast over the same "
                    "(fp32 master weights); hip_convs: the towers' 128 -> 128 convolutions -- forward, data gradient, weight gradient -- on the "
                    "hand-written HIP kernels (mz_conv3x3_forward / mz_conv3x3_wgrad: bf16 operands, fp32 sums, fp32 master weights; "
                    "gradients closer to the fp32 step's than autocast's, tests/test_gpu_convgrad.py), everything else as the fp32 step; "
                    "hip_x3: those kernels in split precision (hi = bf16(v), lo = bf16(v - hi), a b = a_hi b_hi + a_lo b_hi + a_hi b_lo: a convolution "
                    "within 3e-6 of fp64, against 2e-3 for one bf16 product).  All steps eager: the step replayed as one hipGraph is 4.1 ms but is not "
                    "shipped -- replays intermittently compute a wrong gradient on this stack, with or without these kernels (DESIGN.md 5)"}
    res.update(hip)
    return res


def bench_loop(args, device, dtype):
    """BASELINE configs[4] on one rank: a whole generation of muzero_amd.loop (self-play -> replay window -> training
    steps -> weights published into the serving engine).  With N ranks the same code adds the trajectory all_gather and
    the one gradient all-reduce per step."""
    from muzero_amd import config, train as T
    from muzero_amd.loop import MuZeroLoop
    T.LOG = None
    g, steps = args.selfplay_games, args.loop_train_steps
    config.num_unroll_steps = args.train_unroll
    lp = MuZeroLoop(g, device=device, infer_dtype=dtype, num_simulations=args.selfplay_sims, window_size=g,
                    batch_size=args.train_batch, use_graph=not args.no_graph)
    lp.selfplay.play_games(max_moves=2)   # warm-up: graph capture
    lp.generation(2)                      # warm-up: optimiser state, kernel selection
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = lp.generation(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same generation with the training steps on a second stream while the games are played (main.py:32-37)
    lp.generation(2, overlap=True)        # warm-up of the second stream's allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp.generation(steps, overlap=True)
    torch.cuda.synchronize()
    dto = time.perf_counter() - t0
    # and with the training steps' tower convolutions on the HIP kernels (train_convs="hip")
    lp.hip_convs = True
    lp.generation(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp.generation(steps)
    torch.cuda.synchronize()
    dth = time.perf_counter() - t0
    lp.generation(2, overlap=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp.generation(steps, overlap=True)
    torch.cuda.synchronize()
    dtho = time.perf_counter() - t0
    config.num_unroll_steps = 3
    return {"games": g, "simulations": args.selfplay_sims, "train_steps": steps, "batch": args.train_batch,
            "unroll_steps": args.train_unroll, "seconds_per_generation": dt, "games_per_sec": g / dt,
            "overlapped_seconds_per_generation": dto, "overlapped_games_per_sec": g / dto,
            "hip_convs_note": "train_convs=hip", "hip_convs_seconds_per_generation": dth, "hip_convs_games_per_sec": g / dth,
            "hip_convs_overlapped_seconds_per_generation": dtho, "hip_convs_overlapped_games_per_sec": g / dtho,
            "p_loss": float(losses[-1][0]), "v_loss": float(losses[-1][1])}


def bench_cpu_baseline(args, roots, image, noise):
    """The CPU port on the SAME roots as the GPU leg: C oracle tree search (one call per simulation over all trees, OpenMP)
    + PyTorch CPU fp32 inference on the host cores; a bounded sample (one search of every tree)."""
    import numpy as np
    from oracle import mz_oracle as O
    from muzero_amd.network import Network
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    torch.manual_seed(1261)
    net = Network().eval()
    n, S = min(args.cpu_trees, roots.n), args.sims
    to_play = roots.to_play[:n].cpu().numpy()
    legal = roots.legal[:n].cpu().numpy().view(np.uint64)
    img = image[:n].float().cpu()
    nz = noise[:n].cpu().numpy()
    tb = O.TreeBatch(n, S)
    pool = np.zeros((S + 1, n, 512), np.float32)
    x = np.zeros((n, 10, 8, 8), np.float32)
    xt = torch.from_numpy(x)
    t0 = time.perf_counter()
    with torch.no_grad():
        v, p, h = net.initial_inference(img)
        pool[0] = h.reshape(n, 512).numpy()
        tb.root_init(to_play, legal, p.numpy(), nz)
        for s in range(S):
            tb.select(pool, x)
            v, p, h = net.recurrent_inference_packed(xt)
            pool[s + 1] = h.reshape(n, 512).numpy()
            tb.expand_backup(v.numpy().reshape(-1), p.numpy())
    dt = time.perf_counter() - t0
    out = {"value": n * S / dt, "unit": "simulations/s", "cores": threads, "kind": "port",
           "sample": "one search of %d trees x %d sims on the GPU leg's roots (positions after 0..59 moves, same noise): C oracle "
                     "(oracle/mz_oracle.c, one OpenMP call per simulation) + PyTorch CPU fp32 inference at batch %d, %.1f s" % (n, S, n, dt)}
    # the reference cannot run on the GPU box; its ratio to this port was measured with both on the
    # build container's 8 cores (tools/cpu_ratio.py -> profiles/rNN/cpu_ratio_container.json, the newest round's)
    f = newest_profile("cpu_ratio_container.json")
    if f:
        r = json.load(open(f))
        out["ratio_to_reference"] = r["port_sims_per_sec"] / r["reference_sims_per_sec"]
        out["reference_container"] = {k: r[k] for k in ("reference_sims_per_sec", "port_sims_per_sec", "cores", "workload")}
        out["reference_container"]["source"] = os.path.relpath(f, ROOT)
    return out, tb.root_visits()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # rehearsal of the N > 1 protocol on a one-GPU box: MZ_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    # (RCCL needs one device per rank); the numbers of such a run mean nothing, the code path is the real one
    rehearsal = os.environ.get("MZ_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("MZ_BENCH_FORCE_DIST") == "1":   # (forced: the RCCL protocol with a single rank, tests/test_gpu_rccl.py)
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from muzero_amd.env import make_image
    from muzero_amd.network import InferenceEngine, Network
    from muzero_amd.selfplay import SelfPlayEngine

    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[args.dtype]
    torch.manual_seed(1261)  # main.py:18: identical random-init weights on every rank
    net = InferenceEngine(Network(), args.trees, dtype, tower=args.tower)
    n, S = args.trees, args.sims
    eng = SelfPlayEngine(n, net, num_simulations=S, device=device, use_graph=not args.no_graph, seed=1261 + rank)
    from muzero_amd.shard import game_block, max_over_ranks
    game0, _ = game_block(n * world, rank, world)  # weak scaling: n trees per GPU, ids partitioned by rank
    roots, hw, hb, idx = make_roots(n, 1261, game0, device)
    eng.env = roots
    image = make_image(hw, hb, idx, n, dtype)

    if eng._image.dtype == image.dtype:   # the engine's own image buffer: what play_games fills (no copy in front of the graph)
        eng._image.copy_(image)
        image = eng._image

    def step():
        return eng.search(image, copy=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, device)
    sims = args.steps * n * S * world
    out = {
        "metric": "mcts_simulations_per_sec", "value": sims / dt, "unit": "simulations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        # the arithmetic the path computes in: the tree is the reference's (double priors / pb_c, fp32 value sums); the network
        # inference that feeds it runs in args.dtype (the reference's network is fp32: see "fp32_inference" and "bf16_vs_fp32_search")
        "dtype": "f64/f32 tree + %s inference" % args.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: %d game trees per GPU x %d simulations per search; random-init MuZero net "
                               "(seed 1261), %s inference on the whole leaf batch once per simulation (%s); roots = positions after "
                               "0..59 random moves; tree arithmetic f64/f32 as the reference" %
                               (n, S, args.dtype, "one HIP launch: LDS-resident MFMA conv tower + heads" if getattr(net, "tower", "") == "hip"
                                else "PyTorch-ROCm / MIOpen"),
                   "trees_per_gpu": n, "simulations": S, "hipgraph": not args.no_graph, "parallelism": "games sharded x%d" % world},
    }

    def tower_roofline():
        # ---- the dominant kernel of the timed region: the inference launch (k_tower: conv tower + heads, ~97 % of a
        # simulation step), HIP-event timed back to back on the launch stream; MFMA bound
        # (into one reused set of output buffers, as the timed search runs it)
        net_out = net.output_buffers(n) if getattr(net, "tower", "") == "hip" else None
        if net_out is not None and not args.no_graph:
            # 25 launches captured in one hipGraph, as the timed search replays them (eager launches from Python leave a gap of
            # ~20 us between dependent kernels that the graph does not have)
            sg = torch.cuda.Stream(device=device)
            sg.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sg):
                net.recurrent(eng.mcts.net_in, out=net_out)
            torch.cuda.current_stream().wait_stream(sg)
            torch.cuda.synchronize()
            tg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(tg):
                for _ in range(25):
                    net.recurrent(eng.mcts.net_in, out=net_out)
            ms_net = event_time_ms(tg.replay, 4) / 25
        else:
            ms_net = event_time_ms((lambda: net.recurrent(eng.mcts.net_in, out=net_out)) if net_out is not None else (lambda: net.recurrent(eng.mcts.net_in)), 50)
        peak = F32_PEAK_TFLOPS if dtype == torch.float32 else BF16_PEAK_TFLOPS
        hip = getattr(net, "tower", "") == "hip"
        flop = n * (TOWER_FLOP if hip else NET_MFLOP * 1e6)
        tf = flop / (ms_net * 1e-3) / 1e12
        return {"kernel": "k_tower (mz_inference_forward: LDS-resident conv tower + prediction heads)" if hip else
                           "recurrent_inference (MIOpen/hipBLASLt via PyTorch)", "bound": "mfma", "achieved": tf, "peak": peak,
                           "unit": "TFLOP/s", "frac": tf / peak, "traffic": pmc_traffic("k_tower") if hip else None, "traffic_source": pmc_traffic_src("k_tower")[1] if hip else None,
                           "us_per_launch": ms_net * 1e3, "boards_per_launch": n, "flop_per_board": TOWER_FLOP if hip else NET_MFLOP * 1e6,
                           "algorithmic_flop_per_launch": flop, "share_of_step": ms_net / (dt / args.steps / S * 1e3)}

    if rank == 0 and world > 1:   # the dominant kernel's roofline on rank 0 (the other legs are N = 1 only); the other ranks wait at the last barrier
        try:
            out["roofline"] = tower_roofline()
        except Exception as e:   # noqa: BLE001
            out["roofline_error"] = "%s: %s" % (type(e).__name__, e)
    if rank == 0 and world == 1 and not args.skip_extras:
        # ---- roofline of the dominant hand-written kernel: k_backup_select, HIP-event timed on the launch stream
        from muzero_amd import _lib as L
        from muzero_amd import mcts as MC
        noise = eng.dirichlet_noise(roots.legal)
        v0, p0, h0 = net.initial(image)
        h0 = h0.reshape(n, 8, 8, 8).contiguous()
        # pass 1: one real search, recording the network outputs of every simulation and the path depths
        eng.mcts.root_init(roots.to_play, roots.legal, p0, h0, noise)
        eng.mcts.select()
        rec, depth_sum = [], 0.0
        for s in range(S - 1):
            v, p, h = net.recurrent(eng.mcts.net_in)
            rec.append((v.clone(), p.clone(), h.reshape(n, 8, 8, 8).clone()))
            depth_sum += float(eng.mcts.meta()[:, MC.META_PATH_LEN].float().mean())
            eng.mcts.backup_select(*rec[-1])
        # pass 2: replay the same search with the kernel launched back to back (identical trees,
        # nothing but k_backup_select between the two events on the launch stream)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot_ms, launches = 0.0, 0
        routed = eng._whole_graph_ok()
        for rep in range(5):
            eng.mcts.root_init(roots.to_play, roots.legal, p0, h0, noise)
            eng.mcts.select()
            torch.cuda.synchronize()
            start.record()
            for s in range(S - 1):
                if routed:   # the form the timed search runs: no hidden-state traffic in the tree kernel
                    eng.mcts.backup_select_routed(rec[s][0], rec[s][1])
                else:
                    eng.mcts.backup_select(*rec[s])
            end.record()
            end.synchronize()
            tot_ms += start.elapsed_time(end)
            launches += S - 1
        del rec
        d = depth_sum / (S - 1)
        eb = 2 if dtype != torch.float32 else 4
        # algorithmic bytes per simulation: descent d x 65 x 16 B; expansion 260+4 B in, 65 x 16 B out;
        # backup (d+1) x 16 B r/w; path/leaf records 16 B; then what the kernel moves for the network input: routed = two action
        # planes + two slot words (the hidden states stay in the pool); rows = hidden copy-in 2 x 512 x eb, gather 512 x eb in + 640 x eb out
        tree_bytes = 1040.0 * d + 1040 + 264 + 16.0 * (d + 1) + 16
        hid_bytes = (128 * eb + 8) if routed else (2 * 512 + 512 + 640) * eb
        survey_bytes = 796.0 * d + 1076     # 12-byte child records, hidden traffic accounted to the inference step
        us = tot_ms / launches * 1e3
        gbps = n * (tree_bytes + hid_bytes) / (us * 1e-6) / 1e9
        out["roofline_tree"] = {"kernel": "k_backup_select", "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": gbps / HBM_PEAK_GBPS, "traffic": pmc_traffic("k_backup_select"), "traffic_source": pmc_traffic_src("k_backup_select")[1], "us_per_launch": us,
                                "mean_depth": d, "bytes_per_sim_tree": tree_bytes, "bytes_per_sim_hidden": hid_bytes, "routed": bool(routed),
                                "bytes_per_sim_survey": survey_bytes, "frac_survey_bytes": n * survey_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                "trees_per_launch": n, "algorithmic_bytes_per_launch": n * (tree_bytes + hid_bytes),
                                "share_of_step": us * 1e-3 / (dt / args.steps / S * 1e3),
                                "note": "frac = this build's byte count (16-byte child records with float64 priors) over time; frac_survey_bytes = SURVEY "
                                        "8(d)'s 796 d + 1076.  The kernel is a chain of dependent scattered 1 KB reads per tree (latency, not bandwidth): "
                                        "profiles/r02/tree_experiment_latency.txt",
                                "valu_static": {k: (kernel_isa("k_backup_select") or {}).get(k) for k in
                                                ("valu_instructions", "fp64_valu_instructions", "salu_instructions", "vmem_instructions", "has_loop")}}
        out["roofline"] = tower_roofline()
        out.update(bench_env(args, device))
        if args.selfplay_games > 0:
            g = args.selfplay_games
            net_sp = InferenceEngine(Network(), g, dtype, tower=args.tower)
            sp = SelfPlayEngine(g, net_sp, num_simulations=args.selfplay_sims, device=device, use_graph=not args.no_graph)
            sp.play_games(max_moves=2)  # warm-up: graph capture, kernel selection
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rec = sp.play_games()
            torch.cuda.synchronize()
            dts = time.perf_counter() - t1
            out["selfplay"] = {"games": g, "simulations": args.selfplay_sims, "moves": rec.length, "seconds": dts,
                               "games_per_sec": g / dts, "simulations_per_sec": g * rec.length * args.selfplay_sims / dts}
            if args.train_batch > 0:
                # optional legs: a failure is reported in the line, it does not take the headline with it
                for leg, fn in (("train", lambda: bench_train(args, device, rec)), ("loop", lambda: bench_loop(args, device, dtype))):
                    try:
                        out[leg] = fn()
                    except Exception as e:   # noqa: BLE001
                        out[leg + "_error"] = "%s: %s" % (type(e).__name__, e)
                    if leg == "train":
                        del sp, net_sp, rec
        # ---- BASELINE configs[3] per-GPU size: 8192 trees x 100 simulations (what each of 8 GPUs runs in the sharded config)
        if args.config4_searches > 0 and getattr(net, "tower", "") == "hip":
            n4, S4 = 8192, 100
            torch.manual_seed(1261)
            net4 = InferenceEngine(Network(), n4, dtype, tower=args.tower)
            eng4 = SelfPlayEngine(n4, net4, num_simulations=S4, device=device, use_graph=not args.no_graph, seed=1261 + rank)
            roots4, hw4, hb4, idx4 = make_roots(n4, 1261, game0, device)
            eng4.env = roots4
            make_image(hw4, hb4, idx4, n4, dtype, out=eng4._image)
            eng4.search(eng4._image, copy=False)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(args.config4_searches):
                eng4.search(eng4._image, copy=False)
            torch.cuda.synchronize()
            dt4 = (time.perf_counter() - t4) / args.config4_searches
            out["config4_per_gpu"] = {"trees": n4, "simulations": S4, "ms_per_search": dt4 * 1e3, "simulations_per_sec": n4 * S4 / dt4,
                                      "note": "BASELINE configs[3] (65,536 games x 100 simulations over 8 GPUs) as one GPU's share"}
            del eng4, net4, roots4, hw4, hb4
        # ---- reference-precision companion (network.py:146-163 is fp32): the same searches with fp32 inference through
        # PyTorch-ROCm/MIOpen and an fp32 hidden-state pool, and what bf16 serving does to the search's OUTPUT
        bf16_visits, _ = eng.search(image, noise)
        bf16_visits = bf16_visits.clone()
        del eng
        if args.fp32_steps > 0 and dtype != torch.float32:
            from muzero_amd.metrics import visit_agreement
            torch.manual_seed(1261)
            net32 = InferenceEngine(Network(), n, torch.float32, tower="miopen")
            eng32 = SelfPlayEngine(n, net32, num_simulations=S, device=device, use_graph=not args.no_graph, seed=1261 + rank)
            eng32.env = roots
            image32 = image.float()
            fp32_visits, _ = eng32.search(image32, noise)   # warm-up + the agreement sample (same roots, same noise)
            fp32_visits = fp32_visits.clone()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.fp32_steps):
                eng32.search(image32, noise)
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t1) / args.fp32_steps
            ms32 = event_time_ms(lambda: net32.recurrent(eng32.mcts.net_in), 10)
            tf32 = n * NET_MFLOP * 1e6 / (ms32 * 1e-3) / 1e12
            out["fp32_inference"] = {"value": n * S / dt32, "unit": "simulations/s", "ms_per_step": dt32 * 1e3, "steps": args.fp32_steps,
                                     "inference": "PyTorch-ROCm / MIOpen fp32 (+ fused bias/ReLU epilogue), fp32 hidden-state pool",
                                     "roofline": {"bound": "mfma", "achieved": tf32, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                  "frac": tf32 / F32_PEAK_TFLOPS, "us_per_launch_sequence": ms32 * 1e3, "traffic": None}}
            out["bf16_vs_fp32_search"] = dict(visit_agreement(fp32_visits, bf16_visits),
                                              note="same %d roots, same Dirichlet noise, %d simulations: bf16 HIP tower vs fp32 MIOpen inference" % (n, S))
            del eng32, net32
            # the same fp32 search with the towers' 128 -> 128 convolutions on the hand-written kernels in split precision
            try:
                torch.manual_seed(1261)
                netx = InferenceEngine(Network(), n, torch.float32, channels_last=False, tower="hip_x3")
                engx = SelfPlayEngine(n, netx, num_simulations=S, device=device, use_graph=not args.no_graph, seed=1261 + rank)
                engx.env = roots
                x3_visits, _ = engx.search(image32, noise)
                x3_visits = x3_visits.clone()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.fp32_steps):
                    engx.search(image32, noise)
                torch.cuda.synchronize()
                dtx = (time.perf_counter() - t1) / args.fp32_steps
                msx = event_time_ms(lambda: netx.recurrent(engx.mcts.net_in), 10)
                tfx = n * NET_MFLOP * 1e6 / (msx * 1e-3) / 1e12
                out["fp32_split_precision_inference"] = dict(
                    value=n * S / dtx, unit="simulations/s", ms_per_step=dtx * 1e3, steps=args.fp32_steps,
                    inference="fp32 serving: each tower (first convolution, three residual blocks, last convolution) as ONE LDS-resident launch in "
                              "split precision (k_restower_x3: hi / lo bf16 activation images, three bf16 MFMA products per fp32 product, fp32 sums, "
                              "skip path in fp32 registers) with the prediction heads in plain fp32 behind it (mz_inference_x3_forward)",
                    roofline={"bound": "mfma", "achieved": 3 * tfx, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 3 * tfx / BF16_PEAK_TFLOPS,
                              "us_per_launch_sequence": msx * 1e3, "fp32_grade_tflops": tfx,
                              "traffic": pmc_traffic_src("k_restower_x3", "x3_pmc_summary.json")[0],
                              "traffic_source": pmc_traffic_src("k_restower_x3", "x3_pmc_summary.json")[1],
                              "note": "bf16 MFMA work = 3 x the network's FLOPs, one launch per inference (tower + heads)"},
                    agreement_with_fp32_search=visit_agreement(fp32_visits, x3_visits))
                del engx, netx
            except Exception as e:   # noqa: BLE001  (optional leg)
                out["fp32_split_precision_inference_error"] = "%s: %s" % (type(e).__name__, e)
        else:
            fp32_visits = None
        cpu, cpu_visits = bench_cpu_baseline(args, roots, image, noise)
        out["cpu_baseline"] = cpu
        if fp32_visits is not None:
            from muzero_amd.metrics import visit_agreement
            k = cpu_visits.shape[0]
            out["cpu_baseline"]["search_agreement_with_gpu_fp32"] = visit_agreement(torch.from_numpy(cpu_visits), fp32_visits[:k].cpu())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
