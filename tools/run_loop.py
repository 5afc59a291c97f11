"""The whole MuZero loop on one GPU, end to end: G generations of (self-play -> replay window -> training -> publish),
then the arena: trained network vs the initial one.
python tools/run_loop.py [generations] [train_steps] [games] [lr weight_decay] [negamax] [fp32|hip|hip_x3] [bf16|x3]
Without lr/weight_decay the optimiser is the reference's (main.py:206: lr 1e-3, weight_decay 0.1 -- it hardly learns in
a short run); give both to see the loop learn.  negamax = 1 sets config.negamax_value (non-default: the search maximises
the mover's own value instead of the reference's child-mover value).  hip / hip_x3: the training step's tower convolutions on
the hand-written kernels (bf16 / split precision).  Last argument x3: self-play serves fp32 on the split-precision tower kernel
(InferenceEngine(tower="hip_x3")) instead of bf16."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import arena, config, train as T
from muzero_amd.loop import MuZeroLoop
from muzero_amd.network import InferenceEngine, Network

G = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
games = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
T.LOG = None
config.num_unroll_steps = 5
if len(sys.argv) > 6 and int(sys.argv[6]):
    config.negamax_value = True
    print("config.negamax_value = True")
convs = sys.argv[7] if len(sys.argv) > 7 else "fp32"
print("training convolutions: %s" % convs)
serve = sys.argv[8] if len(sys.argv) > 8 else "bf16"
print("self-play inference: %s" % serve)
lp = MuZeroLoop(games, num_simulations=50, window_size=4 * games, batch_size=512, train_convs=convs,
                **(dict(infer_dtype=torch.float32, infer_tower="hip_x3") if serve == "x3" else {}))
init = copy.deepcopy(lp.network.state_dict())
if len(sys.argv) > 5:
    lp.optimizer = torch.optim.SGD(lp.network.parameters(), lr=float(sys.argv[4]), momentum=config.momentum, weight_decay=float(sys.argv[5]))
    print("optimiser: SGD lr %s momentum %s weight_decay %s" % (sys.argv[4], config.momentum, sys.argv[5]))
t0 = time.perf_counter()
for g in range(G):
    out = lp.generation(steps)
    torch.cuda.synchronize()
    if any(not (float(a) == float(a) and float(b) == float(b)) for a, b in out):   # a non-finite loss: show the steps of this generation and stop
        print("non-finite loss in generation %d; its steps (p_loss, v_loss):" % g)
        print(" ".join("(%.4f, %.4f)" % (float(a), float(b)) for a, b in out), flush=True)
        bad = [k for k, p in lp.network.named_parameters() if not torch.isfinite(p).all()]
        print("non-finite parameters:", bad[:8], len(bad))
        sys.exit(3)
    pl = sum(float(a) for a, _ in out) / max(1, len(out))
    vl = sum(float(b) for _, b in out) / max(1, len(out))
    print("generation %2d: %5d games in the window, %4d steps, mean p_loss %.4f v_loss %.4f, %.1f s elapsed" %
          (g, lp.replay.buffer_size, lp.network.steps, pl, vl, time.perf_counter() - t0), flush=True)
n_arena = 128
base, target = Network(), Network()
base.load_state_dict(init)
target.load_state_dict(lp.network.state_dict())
be = InferenceEngine(base, n_arena, torch.bfloat16)
te = InferenceEngine(target, n_arena, torch.bfloat16)
wins, r1, r2 = arena.test(be, te, n_arena, num_simulations=50)
draws = int((r1.white_score == r1.black_score).sum()) + int((r2.white_score == r2.black_score).sum())
print("arena (test.py): trained network won %d of %d games against the initial network (%d draws)" % (wins, 2 * n_arena, draws))
