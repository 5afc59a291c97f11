"""The reference's Python-facing API on top of the HIP engine (batch-of-1 views).

    from muzero_amd.compat import *      # instead of `from reversi import *` + main.py's helpers

gives the names main.py's self-play loop uses, with the reference's signatures, argument meaning
and error behaviour, every one of them executing on the GPU through the C ABI:

    reversi.py   Player, Action, Environment, Node, Game                 (reversi.py:14-36, 39-205, 234-344)
    utilities.py MinMaxStats                                             (utilities.py:114-129)
    main.py      play_game, run_mcts, select_action, select_child, ucb_score, expand_node,
                 backpropagate, add_exploration_noise, softmax_sample    (main.py:69-196, 275-285)

`Environment` is a 1-game BatchedEnv; a search tree is a 1-tree BatchedMCTS pool and `Node`
objects are views of it.  Host-side work is limited to glue the reference also does on the host:
drawing from ``np.random`` (so a seeded run consumes the same random stream as the reference),
formatting bitboards as numpy arrays, and bookkeeping lists.  There is no CPU fallback: without a
GPU these classes raise.

Conventions that differ from the reference, both forced by SURVEY.md H3:
  * ``Environment.legal_actions`` / ``Node.children`` are in ASCENDING action order (the reference's
    order is Python-set iteration order);
  * this is a convenience layer -- the throughput path is muzero_amd.selfplay.SelfPlayEngine.
"""
from dataclasses import dataclass, field
from enum import Enum
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from . import config
from . import mcts as MC
from .env import BatchedEnv
from .env import make_image as _make_image

__all__ = ["Player", "Action", "Environment", "Node", "Game", "MinMaxStats", "play_game", "run_mcts", "select_action",
           "select_child", "ucb_score", "expand_node", "backpropagate", "add_exploration_noise", "softmax_sample", "config"]


class Player(Enum):  # reversi.py:14-16
    WHITE = 1
    BLACK = -1


def _other(p):
    return Player.WHITE if p is Player.BLACK else Player.BLACK


def _dev():
    if not torch.cuda.is_available():
        raise L.MzError("muzero_amd.compat needs a GPU (the reference-shaped API runs on the HIP kernels; no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


@dataclass
class Action:  # reversi.py:19-36
    index: int
    player: Player
    position: Optional[Tuple[int, int]] = field(init=False)

    def __post_init__(self):
        assert self.index >= 0 and self.index <= 64
        self.position = divmod(self.index, 8) if self.index < 64 else None

    def encode(self):
        """(2,8,8) bool planes, computed by mz_action_encode (incl. the reference's row/column and
        whole-plane-pass behaviour)."""
        d = _dev()
        a = torch.tensor([self.index], dtype=torch.uint8, device=d)
        p = torch.tensor([self.player.value], dtype=torch.int8, device=d)
        from .env import action_encode
        return action_encode(a, p)[0].cpu().numpy().astype(bool)


def _bits_to_plane(x):
    return np.unpackbits(np.array([x], dtype="<u8").view(np.uint8), bitorder="little").reshape(8, 8).astype(bool)


class Environment:
    """reversi.py:39-205 on a 1-game device batch."""

    def __init__(self):
        self._env = BatchedEnv(1, _dev())
        self._sync()

    def _sync(self):
        e = self._env
        packed = torch.stack([e.black, e.white, e.legal, e.to_play.long(), e.steps.long(), e.done.long()]).cpu().numpy()
        m = (1 << 64) - 1
        self._black, self._white, self._legal = int(packed[0, 0]) & m, int(packed[1, 0]) & m, int(packed[2, 0]) & m
        self.player = Player(int(packed[3, 0]))
        self.steps = int(packed[4, 0])
        self.done = bool(packed[5, 0])
        idx = [i for i in range(64) if (self._legal >> i) & 1]
        self.legal_actions = [Action(i, self.player) for i in idx] or [Action(64, self.player)]

    @property
    def board(self):
        return (_bits_to_plane(self._white).astype(np.int8) - _bits_to_plane(self._black).astype(np.int8))

    @property
    def possible_pos(self):  # frontier of reversi.py:49-50,188-194: empty squares next to a disc
        occ = self._white | self._black
        out = set()
        for i in range(64):
            if (occ >> i) & 1:
                continue
            r, c = divmod(i, 8)
            if any(0 <= r + dr <= 7 and 0 <= c + dc <= 7 and (occ >> ((r + dr) * 8 + c + dc)) & 1
                   for dr in (-1, 0, 1) for dc in (-1, 0, 1) if dr or dc):
                out.add((r, c))
        return out

    def get_legal_actions(self):  # reversi.py:109-117
        return [a.position for a in self.legal_actions if a.index != 64]

    def update_legal_actions(self):  # reversi.py:119-126
        self._env.legal.copy_(self._env.legal_mask())
        self._sync()

    def check_position(self, position):  # reversi.py:96-106
        return bool((int(self._env.legal_mask().cpu()[0]) >> (position[0] * 8 + position[1])) & 1)

    def step(self, action: Action, player: Player = None):  # reversi.py:130-180
        assert action.player is self.player
        a = torch.tensor([action.index], dtype=torch.uint8, device=self._env.device)
        reward, status = self._env.step(a)
        st = int(status.cpu()[0])
        if st == L.ST_ILLEGAL:
            raise AssertionError("{} out of {}".format(action, self.legal_actions))  # reversi.py:143-144
        if st == L.ST_DONE:
            raise RuntimeError("step() on a finished game")
        r = int(reward.cpu()[0])
        self._sync()
        return self.observe(), r, self.done

    def observe(self):  # reversi.py:182-186
        return np.stack((_bits_to_plane(self._white), _bits_to_plane(self._black)), axis=0)

    def check_win(self):  # reversi.py:197-205
        w, b = bin(self._white).count("1"), bin(self._black).count("1")
        return 0 if w == b else (self.player.value if w > b else -self.player.value)

    def print_board(self):  # reversi.py:208-231 (compact form)
        for row in self.board:
            print(" ".join("O" if v == 1 else "X" if v == -1 else "." for v in row))


class MinMaxStats:  # utilities.py:114-129
    """A class that holds the min-max values of the tree."""

    def __init__(self, known_bounds):
        self.minimum = known_bounds[0] if known_bounds else float("inf")
        self.maximum = known_bounds[1] if known_bounds else -float("inf")

    def update(self, value):
        self.minimum = min(self.minimum, value)
        self.maximum = max(self.maximum, value)

    def normalize(self, value):
        if self.maximum > self.minimum:
            return (value - self.minimum) / (self.maximum - self.minimum)
        return value


class _Tree:
    """one device search tree (1-tree BatchedMCTS) + a lazily refreshed host snapshot"""

    def __init__(self, max_sims):
        self.m = MC.BatchedMCTS(1, max_sims, torch.float32, _dev())
        lay = self.m.layout
        pool = self.m.pool
        nrec = lay.max_nodes * 65
        st = pool[lay.off_stats:lay.off_stats + nrec * 16]
        self.st_i32 = st.view(torch.int32).view(lay.max_nodes, 65, 4)
        self.st_f32 = st.view(torch.float32).view(lay.max_nodes, 65, 4)
        self.st_f64 = st.view(torch.float64).view(lay.max_nodes, 65, 2)
        self.meta_i32 = self.m.meta()[0]
        self.meta_f32 = self.meta_i32.view(torch.float32)
        self.meta_f64 = self.meta_i32.view(torch.float64)
        self.hidden = pool[lay.off_hidden:lay.off_hidden + lay.max_nodes * 2048].view(torch.float32).view(lay.max_nodes, 8, 8, 8)
        self.path_u8 = self.m.path()[0]
        self.snap = None
        self.root_inputs = None

    def dirty(self):
        self.snap = None

    def snapshot(self):
        if self.snap is None:
            meta = self.meta_i32.cpu().numpy().copy()
            n = int(meta[MC.META_N_NODES])
            self.snap = dict(meta=meta, meta_f32=meta.view(np.float32), meta_f64=meta.view(np.float64), n=n,
                             nc=self.st_i32[:n, :, 0].cpu().numpy().view(np.uint32), w=self.st_f32[:n, :, 1].cpu().numpy(),
                             p=self.st_f64[:n, :, 1].cpu().numpy())
        return self.snap

    # MinMaxStats <-> mz_tree_meta
    def push_minmax(self, mm):
        flags = 0
        for bit, v, col in ((1, mm.minimum, 2), (2, mm.maximum, 3)):
            if torch.is_tensor(v):
                flags |= bit
                v = float(v)
            self.meta_f64[col] = float(v)
        self.meta_i32[MC.META_MM_FLAGS] = flags
        self.dirty()

    def pull_minmax(self, mm):
        s = self.snapshot()
        flags = int(s["meta"][MC.META_MM_FLAGS])
        lo, hi = float(s["meta_f64"][2]), float(s["meta_f64"][3])
        if flags & 1:
            mm.minimum = torch.tensor([[lo]], dtype=torch.float32)
        if flags & 2:
            mm.maximum = torch.tensor([[hi]], dtype=torch.float32)


class Node:
    """reversi.py:234-251.  `Node(prior, player)` is detached (plain fields) until `expand_node`
    makes it the root of a device tree; nodes reached through `.children` are views of that tree."""

    def __init__(self, prior: float, player: Player):
        self._tree = None
        self._ref = None        # None = root of its tree, else (parent expanded id, action)
        self._player = player
        self._prior = prior
        self.reward = 0

    @classmethod
    def _view(cls, tree, ref, player):
        n = cls.__new__(cls)
        n._tree, n._ref, n._player, n._prior, n.reward = tree, ref, player, None, 0
        return n

    def __eq__(self, other):
        return isinstance(other, Node) and self._tree is other._tree and self._ref == other._ref and \
            (self._tree is not None or self is other)

    def __hash__(self):
        return hash((id(self._tree), self._ref))

    @property
    def player(self):
        return self._player

    def _rec(self):
        s = self._tree.snapshot()
        q, a = self._ref
        return int(s["nc"][q, a]), float(s["w"][q, a]), float(s["p"][q, a])

    def _id(self):
        """expanded-node id in the pool, or -1"""
        if self._tree is None:
            return -1
        if self._ref is None:
            return 0
        return (self._rec()[0] >> 16) - 1

    @property
    def visit_count(self):
        if self._tree is None:
            return 0
        if self._ref is None:
            return int(self._tree.snapshot()["meta"][MC.META_ROOT_N])
        return self._rec()[0] & 0xFFFF

    @property
    def value_sum(self):
        n = self.visit_count
        if n == 0:
            return 0
        w = float(self._tree.snapshot()["meta_f32"][MC.META_ROOT_W]) if self._ref is None else self._rec()[1]
        return torch.tensor([[w]], dtype=torch.float32)  # the reference accumulates an fp32 (1,1) tensor

    @property
    def prior(self):
        return self._prior if self._ref is None or self._tree is None else self._rec()[2]

    @property
    def hidden_state(self):
        i = self._id()
        return None if i < 0 else self._tree.hidden[i:i + 1].clone()

    @property
    def children(self):
        i = self._id()
        if i < 0:
            return {}
        s = self._tree.snapshot()
        child_player = _other(self._player)
        return {a: Node._view(self._tree, (i, a), child_player) for a in range(65) if s["nc"][i, a] != 0xFFFFFFFF}

    def expanded(self) -> bool:
        return self._id() >= 0

    def value(self) -> float:
        if self.visit_count == 0:
            return 0
        return self.value_sum / self.visit_count


# ---------------------------------------------------------------------------------------------
# search entry points (main.py)
# ---------------------------------------------------------------------------------------------
def _net_out(network_output, d):
    value, policy, hidden = network_output
    return (torch.as_tensor(value, dtype=torch.float32).reshape(1).to(d).contiguous(),
            torch.as_tensor(policy, dtype=torch.float32).reshape(1, 65).to(d).contiguous(),
            torch.as_tensor(hidden, dtype=torch.float32).reshape(1, 8, 8, 8).to(d).contiguous())


def expand_node(node: Node, actions: List[Action], network_output):
    """main.py:165-175.  A detached node becomes the root of a fresh device tree (priors over
    `actions` only); a tree node must be an unexpanded child and gets all 65 actions."""
    d = _dev()
    value, policy, hidden = _net_out(network_output, d)
    if node._tree is None:
        tree = _Tree(max(int(config.num_simulations), 1))
        idx = sorted(a.index for a in actions)
        legal = sum(1 << i for i in idx if i < 64)
        legal = legal - (1 << 64) if legal >= (1 << 63) else legal
        tree.root_inputs = dict(to_play=torch.tensor([node._player.value], dtype=torch.int8, device=d),
                                legal=torch.tensor([legal], dtype=torch.int64, device=d),
                                pass_ok=torch.tensor([int(64 in idx)], dtype=torch.uint8, device=d),
                                policy=policy, hidden=hidden, actions=idx, order=[a.index for a in actions])
        node._tree = tree
        _root_init(tree, None)
        return
    tree = node._tree
    if node._ref is None or node.expanded() or [a.index for a in actions] != list(range(65)):
        raise NotImplementedError("expand_node on a tree node is supported for an unexpanded child with all %d actions "
                                  "(main.py:123)" % config.action_space_size)
    # mz_mcts_expand expands the node the stored path ends in: point a one-edge path at this child
    q, a = node._ref
    tree.path_u8[0] = torch.tensor([q & 0xFF, q >> 8, a, 0], dtype=torch.uint8, device=tree.m.device)
    tree.meta_i32[MC.META_PATH_LEN] = 1
    L.call("mz_mcts_expand", L.dptr(tree.m.pool), L.C.byref(tree.m.layout), L.dptr(policy), L.dptr(hidden), 1, L.current_stream())
    tree.dirty()


def _root_init(tree, noise):
    r = tree.root_inputs
    tree.m.cfg = MC.search_cfg(config)
    L.call("mz_pool_init", L.dptr(tree.m.pool), L.C.byref(tree.m.layout), L.C.byref(tree.m.cfg), L.current_stream())
    tree.m.root_init(r["to_play"], r["legal"], r["policy"], r["hidden"], noise, r["pass_ok"])
    tree.dirty()


def add_exploration_noise(node: Node):
    """main.py:191-196: np.random.dirichlet over the root's children in children order."""
    tree = node._tree
    if tree is None or node._ref is not None or node.visit_count != 0:
        raise NotImplementedError("add_exploration_noise applies to a freshly expanded root (main.py:82-83)")
    # the i-th draw goes to the i-th action of the list expand_node was given (= root.children order in the reference,
    # main.py:192-196), so a caller that passes the reference's own (set-ordered) legal list consumes the np.random stream
    # exactly as the reference does; the prior normalisation still sums in ascending action order
    acts = tree.root_inputs["order"]
    drawn = np.random.dirichlet([config.root_dirichlet_alpha] * len(acts))
    noise = np.zeros((1, 65), np.float64)
    noise[0, acts] = drawn
    _root_init(tree, torch.from_numpy(noise).to(tree.m.device))


def _net_device(network, default):
    try:
        return next(network.parameters()).device
    except (AttributeError, StopIteration, TypeError):
        return default


def run_mcts(root: Node, action_history: List, network):
    """main.py:99-125: config.num_simulations simulations from `root` (expanded, optionally noised)."""
    tree = root._tree
    if tree is None or root._ref is not None:
        raise ValueError("run_mcts needs a root prepared by expand_node (main.py:75-83)")
    S = int(config.num_simulations)
    if tree.m.max_sims - tree.m.sims_done < S:
        raise L.MzError("tree sized for %d simulations" % tree.m.max_sims)
    m = tree.m
    m.cfg = MC.search_cfg(config)
    nd = _net_device(network, torch.device("cpu"))
    for _ in range(S):
        m.select()
        hidden, action = m.hidden_in.to(nd), m.action_in.to(nd)
        with torch.no_grad():
            out = network.recurrent_inference(hidden, action)
        v, p, h = _net_out(out, m.device)
        m.expand_backup(v, p, h)
    tree.dirty()


def _scores(node, min_max_stats, want_scores):
    tree = node._tree
    i = node._id()
    if tree is None or i < 0:
        raise ValueError("node is not expanded")
    tree.m.cfg = MC.search_cfg(config)
    tree.push_minmax(min_max_stats)
    d = tree.m.device
    nid = torch.tensor([i], dtype=torch.int32, device=d)
    act = torch.empty(1, dtype=torch.int32, device=d)
    score = torch.empty((1, 65), dtype=torch.float64, device=d) if want_scores else None
    kind = torch.empty((1, 65), dtype=torch.uint8, device=d) if want_scores else None
    L.call("mz_mcts_node_scores", L.dptr(tree.m.pool), L.C.byref(tree.m.layout), L.C.byref(tree.m.cfg), L.dptr(nid), L.dptr(act),
           L.dptr(score), L.dptr(kind), 1, L.current_stream())
    return int(act.cpu()[0]), (score.cpu().numpy()[0] if want_scores else None), (kind.cpu().numpy()[0] if want_scores else None)


def select_child(node: Node, min_max_stats: MinMaxStats):
    """main.py:144-148 -> (action_idx, child)"""
    a, _, _ = _scores(node, min_max_stats, False)
    return a, node.children[a]


def ucb_score(parent: Node, child: Node, min_max_stats: MinMaxStats):
    """main.py:153-160: a Python float for an unvisited child, an fp32 (1,1) tensor otherwise"""
    if child._tree is not parent._tree or child._ref is None or child._ref[0] != parent._id():
        raise ValueError("child is not a child of parent")
    _, score, kind = _scores(parent, min_max_stats, True)
    a = child._ref[1]
    return torch.tensor([[score[a]]], dtype=torch.float32) if kind[a] == 1 else float(score[a])


def backpropagate(search_path: List[Node], value, player: Player, discount: float, min_max_stats: MinMaxStats):
    """main.py:180-186 along search_path (root .. leaf)."""
    tree = search_path[0]._tree
    if tree is None or search_path[0]._ref is not None:
        raise ValueError("search_path must start at a tree root")
    recs = []
    for n in search_path[1:]:
        if n._tree is not tree or n._ref is None:
            raise ValueError("search_path nodes must belong to one tree")
        recs.append(n._ref)
    path = np.zeros((max(len(recs), 1), 4), np.uint8)
    for k, (q, a) in enumerate(recs):
        path[k] = (q & 0xFF, q >> 8, a, 0)
    if recs:
        tree.path_u8[:len(recs)] = torch.from_numpy(path[:len(recs)]).to(tree.m.device)
    leaf_player = search_path[-1].player
    tree.meta_i32[MC.META_PATH_LEN] = len(recs)
    tree.meta_i32[MC.META_LEAF_PLAYER] = leaf_player.value
    tree.push_minmax(min_max_stats)
    cfg = MC.search_cfg(config)
    cfg.discount = float(discount)
    v = torch.as_tensor(value, dtype=torch.float32).reshape(1).to(tree.m.device)
    if player is not leaf_player:  # signs are relative to `player` (main.py:182)
        v = -v
    L.call("mz_mcts_backup", L.dptr(tree.m.pool), L.C.byref(tree.m.layout), L.C.byref(cfg), L.dptr(v.contiguous()), 1,
           L.current_stream())
    tree.dirty()
    tree.pull_minmax(min_max_stats)


def softmax_sample(distribution, temperature: float):
    """main.py:275-285 on the device kernel; the uniform comes from np.random like np.random.choice's"""
    if temperature not in (0, 1):
        raise NotImplementedError
    visits = np.full((1, 65), -1, np.int32)
    for n, a in distribution:
        visits[0, a] = n
    d = _dev()
    u = np.random.random_sample() if temperature == 1 else 0.0
    out = MC.select_action(torch.from_numpy(visits).to(d), 0 if temperature == 1 else 20,
                           torch.tensor([u], dtype=torch.float64, device=d))
    return int(out.cpu()[0])


def select_action(num_moves: int, node: Node):
    """main.py:128-140 (temperature 1 below 20 moves, then greedy)"""
    visit_counts = [(child.visit_count, action) for action, child in node.children.items()]
    return softmax_sample(visit_counts, 1.0 if num_moves < 20 else 0.0)


# ---------------------------------------------------------------------------------------------
# Game (reversi.py:254-344) and play_game (main.py:69-92)
# ---------------------------------------------------------------------------------------------
class Game:
    """A single episode of interaction with the environment."""

    def __init__(self):
        self.environment = Environment()
        self.state_history = [self.environment.observe()]
        self.action_history = []
        self.rewards = []
        self.child_visits = []
        self.root_values = []
        self.action_space_size = 65
        self.discount = 1
        d = self.environment._env.device
        self._hw = torch.zeros((config.max_moves + 1, 1), dtype=torch.int64, device=d)
        self._hb = torch.zeros((config.max_moves + 1, 1), dtype=torch.int64, device=d)
        self._hw[0], self._hb[0] = self.environment._env.white, self.environment._env.black

    def terminal(self) -> bool:
        return self.environment.done

    def legal_actions(self) -> List[Action]:
        return self.environment.legal_actions

    def apply(self, action: Action):
        state, reward, _ = self.environment.step(action)
        self.rewards.append(reward)
        self.state_history.append(state)
        self.action_history.append(action)
        t = len(self.action_history)
        if t < self._hw.shape[0]:
            self._hw[t], self._hb[t] = self.environment._env.white, self.environment._env.black

    def store_search_statistics(self, root: Node):
        children = root.children
        sum_visits = sum(child.visit_count for child in children.values())
        self.child_visits.append([children[a].visit_count / sum_visits if a in children else 0
                                  for a in range(self.action_space_size)])
        self.root_values.append(root.value())

    def make_image(self, state_index: int):
        """reversi.py:301-312 through mz_make_image -> np.float32 (9,8,8)"""
        return _make_image(self._hw, self._hb, int(state_index), 1)[0].cpu().numpy()

    def make_target(self, state_index: int, num_unroll_steps: int):
        """reversi.py:314-341: value = final reward signed by whether the mover at that index is the
        last mover; zero policy past the end."""
        target_value, target_policy = [], []
        for current_index in range(state_index, state_index + num_unroll_steps + 1):
            if current_index < len(self.root_values):
                same = self.action_history[current_index].player is self.action_history[-1].player
                target_value.append(self.rewards[-1] if same else -self.rewards[-1])
                target_policy.append(self.child_visits[current_index])
            else:
                target_value.append(0)
                target_policy.append([0 for _ in range(config.action_space_size)])
        return target_value, target_policy

    def to_play(self) -> Player:
        return self.environment.player


def play_game(network) -> Game:
    """main.py:69-92, line for line, on the shims above."""
    game = Game()
    while not game.terminal() and len(game.action_history) < config.max_moves:
        root = Node(0, game.to_play())
        current_observation = torch.from_numpy(game.make_image(game.environment.steps))
        with torch.no_grad():
            net_output = network.initial_inference(current_observation.to(_net_device(network, torch.device("cpu"))))
        expand_node(root, game.legal_actions(), net_output)
        add_exploration_noise(root)
        run_mcts(root, game.action_history, network)
        action = select_action(len(game.action_history), root)
        game.apply(Action(action, game.to_play()))
        game.store_search_statistics(root)
    return game
