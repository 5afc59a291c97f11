"""Knobs of the self-play engine.  Names and default values are the reference's module-level
constants (config.py:10-50, cited per line) because its code reads them as ``config.<name>`` at
call time; everything here can be reassigned at run time the same way.

Only the knobs marked [search] / [game] are read by the HIP path; the training knobs are carried
so code written against the reference's ``config`` keeps importing.
"""

# ---- [game] --------------------------------------------------------------------------------------
action_space_size = 65           # config.py:10   64 squares + pass
max_moves = 60                   # config.py:13   play_game loop bound (main.py:72)
state_history_len = 4            # config.py:48   board pairs in make_image
state_shape = (state_history_len * 2 + 1, 8, 8)  # config.py:49
num_channels = 128               # config.py:50   network width

# ---- [search] ------------------------------------------------------------------------------------
num_simulations = 60             # config.py:14
discount = 1.0                   # config.py:15
root_dirichlet_alpha = 0.03      # config.py:18
root_exploration_fraction = 0.25  # config.py:19
pb_c_base = 19652                # config.py:22
pb_c_init = 1.25                 # config.py:23
known_bounds = (-1, 1)           # config.py:29   MinMaxStats initialisation; None = unbounded


def visit_softmax_temperature(num_moves, training_steps):
    """config.py:3-7.  NOTE: the reference's select_action ignores this function and hard-codes
    ``t = 1.0 if num_moves < 20 else 0.0`` (main.py:133-136); the engine follows main.py."""
    return 1.0 if num_moves < 6 else 0.0


visit_softmax_temperature_fn = visit_softmax_temperature  # config.py:12

# ---- training (not read by the self-play path) -------------------------------------------------
training_steps = 100000          # config.py:32
checkpoint_interval = 1          # config.py:33
batch_size = 512                 # config.py:34
window_size = 512                # config.py:35
num_unroll_steps = 3             # config.py:36
td_steps = 60                    # config.py:37
training_interval = 0.5          # config.py:38
weight_decay = 1e-4              # config.py:40
momentum = 0.9                   # config.py:41
lr_init = 0.01                   # config.py:44
lr_decay_rate = 0.1              # config.py:45
lr_decay_steps = 400e3           # config.py:46

# ---- build-only knob -----------------------------------------------------------------------------
# select_child tie-break.  False = the reference: builtin max over (score, action, child) tuples
# resolves equal scores to the HIGHEST action index (main.py:145-147).
tie_break_lowest = False
# ucb_score's value term.  False = the reference: normalize(child.value()) (main.py:159) with Node.value() kept from the
# point of view of the player to move AT the child (backpropagate, main.py:182) -- i.e. the parent maximises its
# OPPONENT's value (the two-player sign slip of the published MuZero pseudocode).  Measured with this engine
# (profiles/r01/loop_run_lr0.005_*.txt): after 15,000 training steps of the same recipe the trained
# network wins 43 of 256 arena games against its own random initialisation with the reference's sign and 213 of 256 with
# True = normalize(-child.value()), where the search maximises the mover's own value.
negamax_value = False
