"""Batched self-play: the per-move composition of reference main.play_game (main.py:69-92) for n
games in lockstep on one GPU.

Every game advances one real move per outer iteration.  In the reference every action (passes
included) consumes a step and ``done`` is only set by the 60th step, so all games last exactly
``config.max_moves`` = 60 iterations -- lockstep loses nothing.

    image  = make_image(step)                    HIP   (Game.make_image, reversi.py:301-312)
    v,p,h  = net.initial(image)                  torch (network.initial_inference)
    root_init(to_play, legal, p, h, noise)       HIP   (expand_node + add_exploration_noise)
    S x [ v,p,h = net.recurrent(net_in) ;        torch (network.recurrent_inference)
          backup_select(v,p,h) ]                 HIP   (expand_node + backpropagate + next descent)
    visits, root_value = root_stats()            HIP   (Game.store_search_statistics inputs)
    action = select_action(visits, step, u)      HIP   (select_action / softmax_sample)
    env.step(action)                             HIP   (Game.apply -> Environment.step)

The S-loop body (network + fused tree kernel) is captured once into a hipGraph and replayed.
Randomness (Dirichlet noise, sampling uniforms) comes from a torch generator on the device, or is
supplied by the caller (parity tests replay the reference's own draws).
"""
import os

import torch

from . import _lib as L
from . import config
from .env import BatchedEnv, make_image
from .mcts import META_ERROR, BatchedMCTS, select_action


class NetworkAdapter:
    """serve a reference-API network (initial_inference / recurrent_inference(hidden, action))"""

    def __init__(self, net):
        self.net = net
        self.dtype = torch.float32

    @torch.no_grad()
    def initial(self, image):
        v, p, h = self.net.initial_inference(image)
        return v.float().reshape(-1).contiguous(), p.float().contiguous(), h.contiguous()

    @torch.no_grad()
    def recurrent(self, x10):
        v, p, h = self.net.recurrent_inference(x10[:, :8], x10[:, 8:])
        return v.float().reshape(-1).contiguous(), p.float().contiguous(), h.contiguous()


class GameRecords:
    """Trajectories of n games (device tensors, step-major) -- the fields of reference ``Game``
    (reversi.py:257-266): state_history, action_history, rewards, child_visits (as raw visit
    counts, 0 for an action that is no root child), root_values."""

    def __init__(self, n, device, max_moves, max_visits=None):
        """max_visits: the largest count a root child can reach (= simulations per search).  Counts are kept in the
        narrowest type that holds it -- uint8 up to 255 simulations, int16 up to 32,767, else int32 -- with 0 for an action
        that is no root child (it normalises to 0 either way, reversi.py:292-298): 65 B instead of 260 B per move, which is
        most of a trajectory record (5.2 KB per game on the wire at S <= 255)."""
        self.n, self.max_moves = n, max_moves
        vis_dt = torch.int32 if max_visits is None else (torch.uint8 if max_visits <= 255 else torch.int16 if max_visits <= 32767 else torch.int32)
        self.hist_white = torch.zeros((max_moves + 1, n), dtype=torch.int64, device=device)
        self.hist_black = torch.zeros((max_moves + 1, n), dtype=torch.int64, device=device)
        self.actions = torch.zeros((max_moves, n), dtype=torch.uint8, device=device)
        self.rewards = torch.zeros((max_moves, n), dtype=torch.int8, device=device)
        self.visits = torch.zeros((max_moves, n, 65), dtype=vis_dt, device=device)
        self.root_values = torch.zeros((max_moves, n), dtype=torch.float32, device=device)
        self.length = 0

    def child_visits(self):
        """Game.store_search_statistics (reversi.py:292-299): N_a / sum N, 0 for non-children (float64)"""
        v = self.visits[:self.length].clamp(min=0).double()
        return v / v.sum(-1, keepdim=True)


class SelfPlayEngine:
    def __init__(self, n_games, net, num_simulations=None, device="cuda", use_graph=True, seed=1261, cfg=None, graph_steps=None):
        """graph_steps: simulation steps captured per hipGraph (default: all S - 1 of a search in ONE graph)"""
        self.cfg = cfg or config
        self.n = int(n_games)
        self.S = int(num_simulations if num_simulations is not None else self.cfg.num_simulations)
        self.device = torch.device(device)
        self.net = net
        self.dtype = getattr(net, "dtype", torch.float32)
        self.env = BatchedEnv(self.n, device)
        self.mcts = BatchedMCTS(self.n, self.S, self.dtype, device, cfg=self.cfg)
        self.use_graph = use_graph
        self.graph_steps = graph_steps
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.seed = int(seed)
        self._graph = None
        self._search_graphs = {}   # whole-search hipGraphs by noise mode ("device" / "given" / "none")
        self._g_noise = None
        self._g_ptrs = None
        self._out = None
        self._image = torch.zeros((self.n, 9, 8, 8), dtype=self.dtype, device=self.device)
        self.sims_run = 0

    # ---- randomness ------------------------------------------------------------------------------
    def dirichlet_noise(self, legal):
        """np.random.dirichlet([alpha]*k) over each game's legal actions (main.py:193), by action:
        normalised Gamma(alpha) draws on the legal entries (pass column when nothing is legal)."""
        bits = (legal.unsqueeze(1) >> torch.arange(64, device=self.device, dtype=torch.int64)) & 1
        mask = torch.cat([bits, (legal == 0).long().unsqueeze(1)], 1).double()
        alpha = torch.full((self.n, 65), float(self.cfg.root_dirichlet_alpha), dtype=torch.float64, device=self.device)
        g = torch._standard_gamma(alpha, generator=self.gen).clamp_(min=1e-300) * mask
        return (g / g.sum(1, keepdim=True)).contiguous()

    # ---- one search for every game (main.py:75-87) -------------------------------------------------
    def _net_out(self):
        """one reused set of network outputs when the engine offers it (InferenceEngine on the HIP path): every result is
        consumed by the tree kernel that follows before the next inference is issued on the same stream"""
        if self._out is None and getattr(self.net, "tower", None) == "hip" and hasattr(self.net, "output_buffers") and \
                os.environ.get("MZ_NO_OUT_REUSE") != "1":
            self._out = self.net.output_buffers(self.n)
        return self._out

    def _recurrent(self):
        out = self._net_out()
        return self.net.recurrent(self.mcts.net_in) if out is None else self.net.recurrent(self.mcts.net_in, out=out)

    def _sim_step(self):
        v, p, h = self._recurrent()
        h = h.reshape(self.n, 8, 8, 8)
        L.call("mz_mcts_backup_select", *self.mcts._args(), L.dptr(v), L.dptr(p), L.dptr(h), L.dptr(self.mcts.net_in),
               self.n, L.current_stream())
        return v, p, h

    def _ensure_graph(self):
        if self._graph is not None or not self.use_graph:
            return
        # warm up on a side stream (allocator, MIOpen kernel selection), then capture one simulation
        # step: network + fused tree kernel.  Capture executes nothing, so the trees are untouched;
        # the captured outputs live in the graph's private pool and are kept alive by self._cap.
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                self.net.recurrent(self.mcts.net_in)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # one graph = `_graph_steps` consecutive simulation steps (every step reads what the previous one wrote, so the
        # chain is just kernel nodes in sequence); replaying one big graph instead of S-1 small ones removes the host
        # launch gap between steps
        self._graph_steps = max(1, min(self.S - 1, int(self.graph_steps or self.S - 1)))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._cap = [self._sim_step() for _ in range(self._graph_steps)]
        self._graph = g

    # ---- the WHOLE search as one hipGraph (HIP inference path): initial inference, root expansion with the Dirichlet noise drawn
    # inside k_root_init, first descent, S - 1 x [inference + fused tree kernel], last inference, expansion + backup, root
    # statistics -- 2 S + 5 kernel nodes and nothing else on the stream
    def _whole_graph_ok(self):
        return self.use_graph and self.S > 1 and self._net_out() is not None and os.environ.get("MZ_STEP_GRAPH") != "1"

    def _search_body(self, mode):
        """routed: hidden states never leave the node pool -- the inference launch reads the leaf's parent from its slot and
        writes the new node's state into its slot (mz_inference_forward_routed); the tree kernels copy nothing"""
        v, p, _ = self._net_out()
        m, env, net = self.mcts, self.env, self.net
        pool_h = m.hidden_pool()
        m.route()
        net.infer_routed("repr", self._image, pool_h, v, p, dst_stride=m.layout.max_nodes)
        if mode == "device":
            m.root_init(env.to_play, env.legal, p, None, None, device_noise_seed=self.seed)
        else:
            m.root_init(env.to_play, env.legal, p, None, self._g_noise if mode == "given" else None)
        m.select_routed()
        for s in range(self.S):
            net.infer_routed("dyn", m.act_planes, pool_h, v, p, src_pool=pool_h, src_slot=m.src_slot, dst_slot=m.dst_slot)
            if s + 1 < self.S:
                L.call("mz_mcts_backup_select_routed", *m._args(), L.dptr(v), L.dptr(p), L.C.byref(m.route()), self.n, L.current_stream())
            else:
                L.call("mz_mcts_expand_backup", *m._args(), L.dptr(v), L.dptr(p), None, self.n, L.current_stream())
        return m.root_stats()

    def _search_graph(self, image, noise, mode, copy):
        env = self.env
        ptrs = (env.legal.data_ptr(), env.to_play.data_ptr())
        if self._g_ptrs != ptrs:          # the roots live in other tensors than the captured ones: capture again
            self._search_graphs.clear()
            self._g_ptrs = ptrs
        if image.data_ptr() != self._image.data_ptr():
            self._image.copy_(image)
        if mode == "given":
            if self._g_noise is None:
                self._g_noise = torch.empty((self.n, 65), dtype=torch.float64, device=self.device)
            self._g_noise.copy_(noise)
        ent = self._search_graphs.get(mode)
        if ent is None:
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):          # warm-up outside the capture (allocator, lazy initialisation)
                self._search_body(mode)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                res = self._search_body(mode)
            ent = self._search_graphs[mode] = (g, res)
        ent[0].replay()
        self.mcts.sims_done = self.S
        self.sims_run += self.S * self.n
        visits, root_value = ent[1]
        return (visits.clone(), root_value.clone()) if copy else (visits, root_value)

    def search(self, image, noise="auto", copy=True):
        """root set-up + S simulations for all games. Returns (visits (n,65) i32, root_value (n,) f32).
        noise: (n,65) f64 by action, None (no exploration noise), or "auto" = drawn by the engine (on the device inside the
        root expansion when the search runs as one hipGraph, else from the engine's torch generator).
        copy=False (hipGraph path): the returned tensors are the graph's own output buffers, overwritten by the next search."""
        if self._whole_graph_ok():
            return self._search_graph(image, noise, "device" if isinstance(noise, str) else ("none" if noise is None else "given"), copy)
        out = self._net_out()
        v0, p0, h0 = self.net.initial(image) if out is None else self.net.initial(image, out=out)
        legal, to_play = self.env.legal, self.env.to_play
        if isinstance(noise, str):
            noise = self.dirichlet_noise(legal)
        self.mcts.root_init(to_play, legal, p0, h0.reshape(self.n, 8, 8, 8).contiguous(), noise)
        S = self.S
        if S > 0:
            self.mcts.select()
            if self.use_graph and S > 1:
                self._ensure_graph()
                full, rest = divmod(S - 1, self._graph_steps)
                for _ in range(full):
                    self._graph.replay()
                for _ in range(rest):   # only when graph_steps was set below S - 1
                    v, p, h = self._recurrent()
                    self.mcts.backup_select(v, p, h.reshape(self.n, 8, 8, 8))
                self.mcts.sims_done += full * self._graph_steps
            else:
                for _ in range(S - 1):
                    v, p, h = self._recurrent()
                    self.mcts.backup_select(v, p, h.reshape(self.n, 8, 8, 8))
            v, p, h = self._recurrent()
            self.mcts.expand_backup(v, p, h.reshape(self.n, 8, 8, 8).contiguous())
            self.sims_run += S * self.n
        return self.mcts.root_stats()

    # ---- whole games (main.py:69-92) -----------------------------------------------------------------
    def play_games(self, noise=None, u=None, max_moves=None, on_move=None):
        """Play n games from the opening. noise: optional (moves, n, 65) f64 by action; u: optional
        (moves, n) f64 uniforms for the sampled moves (both default to the engine's generator).
        on_move(t): called on the host once move t's work is enqueued (MuZeroLoop enqueues training steps on another stream there)."""
        T = int(max_moves if max_moves is not None else self.cfg.max_moves)
        rec = GameRecords(self.n, self.device, T, max_visits=self.S)
        env = self.env
        env.reset()
        bad = torch.zeros(self.n, dtype=torch.bool, device=self.device)
        err = torch.zeros(self.n, dtype=torch.bool, device=self.device)
        rec.hist_white[0].copy_(env.white)
        rec.hist_black[0].copy_(env.black)
        for t in range(T):
            make_image(rec.hist_white, rec.hist_black, t, self.n, self.dtype, out=self._image)
            visits, root_value = self.search(self._image, noise[t].contiguous() if noise is not None else "auto", copy=False)
            ut = u[t].contiguous() if u is not None else torch.rand(self.n, dtype=torch.float64, device=self.device,
                                                                     generator=self.gen)
            action = select_action(visits, t, ut)  # len(action_history) == t for every game
            reward, status = env.step(action)
            bad |= status == L.ST_ILLEGAL
            err |= self.mcts.meta()[:, META_ERROR] != 0   # root_init clears the flag, so collect it per move
            rec.actions[t].copy_(action)
            rec.rewards[t].copy_(reward)
            rec.visits[t].copy_(visits.clamp_min(0))   # -1 (not a root child) -> 0; narrowed to the record's count type
            rec.root_values[t].copy_(root_value)
            rec.hist_white[t + 1].copy_(env.white)
            rec.hist_black[t + 1].copy_(env.black)
            rec.length = t + 1
            if on_move is not None:
                on_move(t)
        rec.done = env.done.clone()
        # one sync per call: a move the env rejected (the reference would assert, reversi.py:143-144) or a tree that ran
        # out of pool must not flow into the replay window as a silently corrupt trajectory
        n_bad = int(bad.sum())
        n_err = int(err.sum())
        if n_bad or n_err:
            raise L.MzError("self-play: %d games were handed an illegal action, %d trees flagged an error "
                            "(non-finite network outputs?)" % (n_bad, n_err))
        return rec
