"""muzero() -- the reference's top-level loop (main.py:25-65) as one process per GPU.

Reference: Ray starts cpu_count()-2 self-play actors that push finished games into a ReplayBuffer
actor while the trainer samples from it (main.py:26-39, 55-63, 201-227).  Here every rank of a
torch.distributed job (RCCL on the GPUs; none needed for one GPU) alternates, per generation:

  1. self-play: `games_per_rank` lock-step games on the rank's own game-id block (no collective);
  2. trajectory gather: one all_gather_into_tensor per record field (shard.gather_game_records; 5.2 KB per game with
     uint8 visit counts) so every rank's replay window holds the same games in the same (game-id) order;
  3. training: `train_steps` x (sample_batch on the rank's shard of the global batch ->
     update_weights with ONE flat-bucket gradient all-reduce);
  4. publish: the new weights are copied in place into the rank's InferenceEngine (the parameter
     storage a captured hipGraph replays against) -- SharedStorage.save_network / run_selfplay's
     load_state_dict (main.py:60-61, 216-221) without Ray.

`generation(train_steps, overlap=True)` runs 1 and 3 AT THE SAME TIME, as the reference's actors and trainer do (main.py:32-37):
the training steps are enqueued on a second HIP stream, a few per real move, against the window as it stood before this
generation (its games are saved, and the weights published, at the generation boundary -- the serving engine holds its own copy
of the weights, so nothing a search replays is written under it).  The inference kernel fills every CU, so the two streams
share the GPU kernel by kernel: what the overlap buys is the training step's idle time between its short dependent kernels.

All ranks draw the same (game, position) pairs from `random` (seeded identically, main.py:19) and
take disjoint slices of them, so an N-GPU run trains on exactly the batch a 1-GPU run would.
"""
import random

import torch
import torch.distributed as dist

from . import config
from .network import InferenceEngine, Network
from .replay import ReplayBuffer
from .selfplay import SelfPlayEngine
from .shard import gather_game_records, owner_of_global_game
from . import train as T


class MuZeroLoop:
    def __init__(self, games_per_rank, device="cuda", infer_dtype=torch.bfloat16, num_simulations=None, window_size=None,
                 batch_size=None, seed=1261, use_graph=True, group=None, replay_mode="replicated", train_convs="fp32", infer_tower="auto"):
        """replay_mode "replicated": trajectories are all_gathered and every rank holds the whole window (what
        BASELINE configs[3] describes); "sharded": no gather at all -- every rank keeps only the games it played (1/N of the
        window), all ranks make the same global draws and each trains on the draws that hit its own games, with the
        gradient all-reduce weighted by the local batch sizes.  Both train on exactly the batch one process would draw.
        train_convs "fp32": the reference's arithmetic (PyTorch/MIOpen); "hip": the towers' 128 -> 128 convolutions and their
        gradients on the hand-written matrix-core kernels (bf16 operands, fp32 sums and master weights: train.update_weights
        hip_convs); "hip_x3": those kernels in split precision (three bf16 products per fp32 product: close to fp32).
        infer_tower: InferenceEngine's tower ("auto": the bf16 HIP kernel for infer_dtype bf16; "hip_x3" with infer_dtype float32:
        self-play at the reference's fp32 precision on the split-precision tower kernel)."""
        self.group = group
        self.world = dist.get_world_size(group) if T._dist_on(group) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.device = torch.device(device)
        torch.manual_seed(seed)          # main.py:18: identical initial weights on every rank
        random.seed(seed)                # main.py:19: identical replay draws on every rank
        self.network = Network().to(self.device)
        self.network.train()
        self.optimizer = T.make_optimizer(self.network)
        self.bucket = T.FlatGradBucket(self.network.parameters())
        self.storage = T.SharedStorage(self.network.state_dict())
        self.infer = InferenceEngine(self.network, games_per_rank, infer_dtype, device=self.device, tower=infer_tower,
                                     channels_last=infer_tower != "hip_x3")   # serves its own copy
        self.selfplay = SelfPlayEngine(games_per_rank, self.infer, num_simulations=num_simulations, device=self.device,
                                       use_graph=use_graph, seed=seed + 7919 * self.rank)
        self.batch_size = int(batch_size if batch_size is not None else config.batch_size)
        if self.batch_size % self.world:
            raise ValueError("global batch %d is not divisible by %d ranks" % (self.batch_size, self.world))
        if replay_mode not in ("replicated", "sharded"):
            raise ValueError("replay_mode must be 'replicated' or 'sharded'")
        self.replay_mode = replay_mode
        if train_convs not in ("fp32", "hip", "hip_x3"):
            raise ValueError("train_convs must be 'fp32', 'hip' or 'hip_x3'")
        self.hip_convs = {"fp32": False, "hip": True, "hip_x3": "x3"}[train_convs]
        self.games_per_rank = int(games_per_rank)
        window = int(window_size if window_size is not None else config.window_size)
        if replay_mode == "sharded":
            if window % (self.games_per_rank * self.world):
                raise ValueError("a sharded window of %d games must hold whole generations of %d x %d games" % (window, self.games_per_rank, self.world))
            window //= self.world
        self.replay = ReplayBuffer(self.device, window_size=window, batch_size=self.batch_size if replay_mode == "replicated" else 1)
        self.losses = []
        self.train_stream = None

    def _n_global(self):
        return self.replay.buffer_size * (self.world if self.replay_mode == "sharded" else 1)

    def _save(self, rec):
        if self.replay_mode == "replicated":
            for part in gather_game_records(rec, self.group):   # one dict per rank, in game-id order (this rank's own records when alone)
                self.replay.save_games(part, lengths=rec.length)
        else:
            self.replay.save_games(rec)                          # no collective: the window is sharded by game id

    def _train_step(self):
        """one sample_batch + update_weights on the current stream and the current window; returns the two losses"""
        K = config.num_unroll_steps
        n_global = self._n_global()
        shard = self.batch_size // self.world
        idx = random.sample(range(n_global), self.batch_size)                      # utilities.py:76 (same draws on every rank)
        if self.replay_mode == "replicated":
            pos = [self.replay.sample_position(i) for i in idx]                    # utilities.py:78
            sl = slice(self.rank * shard, (self.rank + 1) * shard)
            batch, scale = self.replay.sample_batch(K, idx[sl], pos[sl]), 1.0
        else:
            # every game of this engine lasts exactly max_moves moves (selfplay.py), so all ranks can make the
            # reference's position draw (utilities.py:104) for games they do not hold
            pos = [random.randint(0, int(self.selfplay.cfg.max_moves) - K) for _ in idx]
            mine = [(li, p) for (r, li), p in ((owner_of_global_game(i, self.games_per_rank, self.world), p) for i, p in zip(idx, pos))
                    if r == self.rank]
            batch = self.replay.sample_batch(K, [li for li, _ in mine], [p for _, p in mine]) if mine else None
            scale = len(mine) * self.world / self.batch_size
        return T.update_weights(self.optimizer, self.network, batch, self.bucket, self.group, loss_scale=scale, hip_convs=self.hip_convs)

    def _publish(self):
        self.storage.save_network(self.network.state_dict())
        self.infer.load_weights(self.storage.get_network())

    def generation(self, train_steps, overlap=False, order="selfplay_first"):
        """one self-play round + `train_steps` updates; returns the losses of this generation.
        overlap=False, order="selfplay_first": self-play, save the games, train on the window that holds them, publish.
        overlap=True: the training steps run on a second stream WHILE the games are played, on the window as it stood before
        this generation; games saved and weights published at the boundary.  order="train_first" (overlap=False) executes that
        same dataflow one after the other -- what the overlapped form is tested against."""
        out = []
        ready = self._n_global() >= self.batch_size              # ReplayBuffer.ready(), utilities.py:65-69, on the global window
        if overlap:
            if self.train_stream is None:
                self.train_stream = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream()
            self.train_stream.wait_stream(main)                  # the window and the weights as the previous generation left them
            done = [0]
            T_moves = int(self.selfplay.cfg.max_moves)

            def on_move(t):
                # spread the steps over the game's moves; enqueued behind the move's own launches, on the other stream
                want = train_steps * (t + 1) // T_moves if ready else 0
                with torch.cuda.stream(self.train_stream):
                    while done[0] < want:
                        out.append(self._train_step())
                        done[0] += 1

            rec = self.selfplay.play_games(on_move=on_move)
            main.wait_stream(self.train_stream)
            self._save(rec)
            if out:
                self._publish()
        elif order == "train_first":
            if ready:
                out = [self._train_step() for _ in range(train_steps)]
            rec = self.selfplay.play_games()
            self._save(rec)
            if out:
                self._publish()
        else:
            rec = self.selfplay.play_games()
            self._save(rec)
            if self._n_global() >= self.batch_size:
                out = [self._train_step() for _ in range(train_steps)]
                self._publish()
        self.losses.extend(out)
        return out


def muzero(generations=1, games_per_rank=None, train_steps=None, **kw):
    """main.py:25-39 for this rank; returns the MuZeroLoop (weights, replay window, losses)."""
    loop = MuZeroLoop(games_per_rank or config.batch_size, **kw)
    for _ in range(generations):
        loop.generation(train_steps if train_steps is not None else config.training_steps)
    return loop
