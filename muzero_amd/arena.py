"""Arena: checkpoint-vs-checkpoint evaluation, reference test.py:12-111, batched on the HIP engine.

The reference plays 10 games with `base` as BLACK / `target` as WHITE (test1) and 10 with the colours
swapped (test2), each move chosen by a 100-simulation search WITHOUT exploration noise (test.py:140-166)
from the Controller's observation (test.py:16-29) and sampled with temperature 0.25 (test.py:169-177),
and prints how many of the 20 games `target` won.  Here all games of a colour assignment advance in
lock-step on one GPU: every step is one batched search by the mover's network.

Differences, both deliberate: the reference's ``while not env.done`` never terminates when the 60th
step is a pass -- games here stop after ``config.max_moves`` steps like play_game; and
legal actions are taken in ascending order.
"""
import torch

from . import _lib as L
from . import config
from .env import DT, BatchedEnv
from .selfplay import SelfPlayEngine

NUM_SIMULATIONS = 100   # test.py:143
TEMPERATURE = 0.25      # test.py:174


def controller_image(hist_white, hist_black, state_index, n, dtype=torch.float32, out=None):
    """Controller.step's observation (test.py:19-29) for n lock-step games; arguments as env.make_image"""
    out = torch.empty((n, 9, 8, 8), dtype=dtype, device=hist_white.device) if out is None else out
    per_game = state_index if torch.is_tensor(state_index) else None
    L.call("mz_controller_image", L.dptr(hist_white), L.dptr(hist_black), hist_white.shape[1], L.dptr(per_game),
           0 if per_game is not None else int(state_index), L.dptr(out), DT[dtype], n, L.current_stream())
    return out


def softmax_sample(visits, temperature, u=None):
    """test.py:228-243 for n visit vectors (int32 (n,65), -1 = not a child); u: float64 (n,) uniforms"""
    n = visits.shape[0]
    out = torch.empty(n, dtype=torch.uint8, device=visits.device)
    L.call("mz_softmax_sample", L.dptr(visits), float(temperature), L.dptr(u), L.dptr(out), n, L.current_stream())
    return out


class ArenaResult:
    def __init__(self, actions, white_score, black_score, done):
        self.actions, self.white_score, self.black_score, self.done = actions, white_score, black_score, done


def play_games(black_net, white_net, n_games, num_simulations=NUM_SIMULATIONS, temperature=TEMPERATURE, u=None, max_steps=None,
               device="cuda", use_graph=True, seed=1261, engines=None):
    """n_games lock-step games, `black_net` moving for BLACK and `white_net` for WHITE (the loop of
    test.py:45-65).  Networks follow the engine protocol (InferenceEngine / NetworkAdapter).
    u: optional (steps, n) float64 uniforms for the sampled moves.  Returns ArenaResult with actions
    (steps, n) uint8 and the final disc counts (test.py:67-68)."""
    n, T = int(n_games), int(max_steps if max_steps is not None else config.max_moves)
    dev = torch.device(device)
    if engines is None:
        engines = {}
    for net in (black_net, white_net):
        if id(net) not in engines:
            engines[id(net)] = SelfPlayEngine(n, net, num_simulations=num_simulations, device=dev, use_graph=use_graph, seed=seed)
    env = BatchedEnv(n, dev)
    hw = torch.zeros((T + 1, n), dtype=torch.int64, device=dev)
    hb = torch.zeros((T + 1, n), dtype=torch.int64, device=dev)
    hw[0], hb[0] = env.white, env.black
    actions = torch.zeros((T, n), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    bad = torch.zeros(n, dtype=torch.bool, device=dev)
    for t in range(T):
        eng = engines[id(black_net if t % 2 == 0 else white_net)]  # BLACK moves on even steps (reversi.py:54,137,176)
        image = controller_image(hw, hb, t, n, eng.dtype)
        eng.env = env
        visits, _ = eng.search(image, noise=None)   # expand_node + run_mcts, no add_exploration_noise (test.py:37-41)
        ut = u[t].contiguous() if u is not None else torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
        a = softmax_sample(visits, temperature, ut)
        _, status = env.step(a)   # finished games report MZ_ST_DONE and stay untouched
        bad |= status == L.ST_ILLEGAL
        actions[t].copy_(a)
        hw[t + 1], hb[t + 1] = env.white, env.black
    if int(bad.sum()):   # the reference would assert (reversi.py:143-144)
        raise L.MzError("arena: %d games were handed an illegal action" % int(bad.sum()))
    pop = lambda x: ((x.unsqueeze(-1) >> torch.arange(64, device=dev)) & 1).sum(-1).to(torch.int32).cpu()  # noqa: E731
    return ArenaResult(actions, pop(env.white), pop(env.black), env.done.clone().cpu())


def test(base, target, n_games=10, **kw):
    """test.py:105-111: n games with target as WHITE (test1: win iff white > black) + n with target as
    BLACK (test2: win iff black > white).  Returns (wins of target out of 2n, result1, result2)."""
    engines = {}
    r1 = play_games(base, target, n_games, engines=engines, **kw)
    r2 = play_games(target, base, n_games, engines=engines, **kw)
    wins = int((r1.white_score > r1.black_score).sum()) + int((r2.white_score < r2.black_score).sum())
    print(wins)  # test.py:111
    return wins, r1, r2
