"""Build-defined test vocabulary shared by the golden generator, the oracle tests and the GPU tests.

Nothing in here comes from the reference: it is the *build's* definition of
  * the counter-based RNG that drives random playouts (config 2 of BASELINE.json),
  * the 64-bit digest used to compress per-step env traces into fixtures,
  * ``HashNet`` -- an analytic stand-in for the MuZero network whose outputs depend only on the
    path (root image, action sequence), evaluated bit-identically by python ints (reference
    harness), numpy (oracle tests) and torch int64 ops (GPU tests).  SURVEY.md H6: visit-count
    parity can only be asserted with replayed / analytic network outputs, never with live convs.

The same formulas are restated in C in ``oracle/mz_oracle.c`` and in HIP in
``muzero_amd/csrc/env_kernels.hip`` (RNG only); tests pin all three against each other.
"""
import numpy as np

M64 = (1 << 64) - 1
K_GAME = 0xD1342543DE82EF95
K_STEP = 0x9E3779B97F4A7C15


def splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def playout_rand(seed, game_id, step):
    """64-bit random word for (seed, game, step)."""
    return splitmix64((seed + game_id * K_GAME + step * K_STEP) & M64)


def pick_kth(seed, game_id, step, count):
    """index k in [0,count) of the legal move to play (multiply-high of the top 32 bits)."""
    r = playout_rand(seed, game_id, step)
    return ((r >> 32) * count) >> 32


def kth_set_bit(mask, k):
    """bit index of the k-th (0-based, ascending) set bit of a 64-bit mask."""
    for i in range(64):
        if (mask >> i) & 1:
            if k == 0:
                return i
            k -= 1
    raise ValueError("k out of range")


def playout_action(seed, game_id, step, legal_mask):
    """action index for a random-play step: 64 (pass) iff no legal placement."""
    cnt = bin(legal_mask).count("1")
    if cnt == 0:
        return 64
    return kth_set_bit(legal_mask, pick_kth(seed, game_id, step, cnt))


def digest_step(h, white, black, to_play, steps, done, reward, legal, pass_flag):
    """fold one env state into a running 64-bit digest (to_play/reward are signed small ints)."""
    h = splitmix64(h ^ white)
    h = splitmix64(h ^ black)
    h = splitmix64(h ^ legal)
    meta = ((to_play & 0xFF) | ((steps & 0xFF) << 8) | ((done & 1) << 16) |
            ((reward & 0xFF) << 24) | ((pass_flag & 1) << 32))
    return splitmix64(h ^ meta)


# ------------------------------------------------------------------------------------------------
# HashNet
# ------------------------------------------------------------------------------------------------
P31 = 2147483647  # 2^31 - 1; every intermediate stays below 2^62 so torch int64 never wraps
MODES = {"hash": 0, "uniform": 1, "sat": 2, "const": 3}


def _g(h, i):
    x = (h + i * 1000003) % P31
    x = (x * 48271 + 7) % P31
    x = x ^ (x >> 15)
    x = (x * 69621 + 3) % P31
    return x


def hashnet_image_hash(image):
    """h0 from a (9,8,8) 0/1 image (numpy)."""
    flat = np.asarray(image).reshape(-1).astype(np.int64)
    w = ((np.arange(flat.size, dtype=np.int64) + 1) * 2654435761) % P31
    return int((flat * w).sum() % P31)


def hashnet_action_code(planes):
    """code of an encoded action: (2,8,8) 0/1 planes -> sum of (j+1) over set entries."""
    flat = np.asarray(planes).reshape(-1).astype(np.int64)
    return int((flat * (np.arange(flat.size, dtype=np.int64) + 1)).sum())


def encode_rows(action):
    """row mask (8 bits) that reference Action.encode sets in its layer: rows r and c of
    position (r, c) = divmod(index, 8) (numpy fancy-indexes the tuple along the row axis,
    reversi.py:32-36); pass (64, position None) sets all 8 rows."""
    if action == 64:
        return 0xFF
    r, c = divmod(action, 8)
    return (1 << r) | (1 << c)


def action_code(action, plane):
    """hashnet_action_code computed from (action index, plane) without materialising planes."""
    rows = encode_rows(action)
    s = 0
    for r in range(8):
        if (rows >> r) & 1:
            s += sum(plane * 64 + r * 8 + col + 1 for col in range(8))
    return s


def hashnet_step(h, code):
    x = (h + code * 7919) % P31
    x = (x * 48271 + 11) % P31
    x = x ^ (x >> 13)
    return (x * 69621 + 1) % P31


def hashnet_outputs(h, mode="hash"):
    """(value float32, policy float32[65]) for state hash h. All values exact in fp32."""
    idx = np.arange(65, dtype=np.int64)
    if mode == "uniform":
        return np.float32(0.0), np.full(65, 1.0 / 64.0, dtype=np.float32)
    if mode == "const":
        return np.float32(0.25), np.ones(65, dtype=np.float32)
    v_int = _g(h, 0) % 8193 - 4096
    if mode == "sat":
        v = np.float32(1.0 if (v_int & 1) else -1.0)
    else:
        v = np.float32(v_int / 4096.0)
    pol = np.empty(65, dtype=np.float64)
    for i in range(65):
        m = _g(h, i + 1) % 8388608
        e = _g(h, i + 101) % 41
        pol[i] = (1.0 + m / 8388608.0) * 2.0 ** (-e)
    return v, pol.astype(np.float32)


def hashnet_hidden(h):
    """(512,) float32 hidden state that carries h and a checkable fill."""
    i = np.arange(512, dtype=np.int64)
    hid = (h * (i + 1)) % 65521
    hid[0] = h % 65536
    hid[1] = h // 65536
    return hid.astype(np.float32)


def hashnet_h_from_hidden(hidden):
    flat = np.asarray(hidden).reshape(-1)
    return int(flat[0]) + 65536 * int(flat[1])
