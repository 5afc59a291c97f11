"""mz_search_noalloc (the entry point that allocates nothing: its scratch is a file-scope device array) against
mz_search (caller-owned workspace) through the C ABI: visits and root values bit-identical, on two calls in a row, the
second with a smaller input -- the array keeps the first call's contents.  Collected last (test_zz_*) on purpose."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def noalloc(fs, image, to_play, legal, noise, n, S):
    from muzero_amd import _lib as L
    visits = torch.empty((n, 65), dtype=torch.int32, device="cuda")
    root_value = torch.empty(n, dtype=torch.float32, device="cuda")
    L.call("mz_search_noalloc", *fs._weights(), C.byref(fs.cfg), L.dptr(image), L.dptr(to_play), L.dptr(legal), L.dptr(noise), S, n,
           L.dptr(visits), L.dptr(root_value), L.current_stream())
    return visits, root_value


def test_noalloc_equals_workspace_search_twice_second_smaller():
    from muzero_amd import _lib as L
    from muzero_amd.env import make_image
    from muzero_amd.mcts import FusedSearch
    from muzero_amd.network import InferenceEngine, Network
    from muzero_amd.selfplay import SelfPlayEngine
    from test_gpu_search import roots
    torch.manual_seed(1261)
    served = InferenceEngine(Network(), 128, torch.bfloat16)
    for n, S, moves in ((128, 50, 11), (37, 20, 30)):          # the second call is smaller in trees and in simulations
        e, hw, hb = roots(n, moves, seed=123 + n)
        image = make_image(hw, hb, moves, n, torch.bfloat16)
        noise = SelfPlayEngine(n, served, num_simulations=S, use_graph=False, seed=9).dirichlet_noise(e.legal)
        fs = FusedSearch(served, n, S)
        want_v, want_r = fs(image, e.to_play, e.legal, noise)
        got_v, got_r = noalloc(fs, image, e.to_play, e.legal, noise, n, S)
        torch.cuda.synchronize()
        assert torch.equal(got_v, want_v), (n, S)
        assert torch.equal(got_r.view(torch.int32), want_r.view(torch.int32)), (n, S)
        assert int(got_v.clamp(min=0).sum(1).min()) == S
        # a different pb_c table, then back: the resident table follows the config
        cfg2 = L.SearchCfg.from_buffer_copy(fs.cfg)
        cfg2.pb_c_init = 2.0
        fs2 = FusedSearch(served, n, S)
        fs2.cfg = cfg2
        w2 = fs2(image, e.to_play, e.legal, noise)
        g2 = noalloc(fs2, image, e.to_play, e.legal, noise, n, S)
        assert torch.equal(w2[0], g2[0]) and not torch.equal(w2[0], want_v)
        again = noalloc(fs, image, e.to_play, e.legal, noise, n, S)
        assert torch.equal(again[0], want_v)
        # the Python wrapper of the same entry point
        v3, r3 = FusedSearch(served, n, S, noalloc=True)(image, e.to_play, e.legal, noise)
        assert torch.equal(v3, want_v) and torch.equal(r3.view(torch.int32), want_r.view(torch.int32))
        # what the header promises beyond the results: after the warm-up calls above, a call (a) takes no device memory --
        # the free bytes the driver reports do not move around it -- and (b) can be recorded into a hipGraph (a hipMalloc, a
        # synchronisation or a synchronous copy inside it would fail the capture) and replayed with the same results
        out_v = torch.empty((n, 65), dtype=torch.int32, device="cuda")
        out_r = torch.empty(n, dtype=torch.float32, device="cuda")

        def call_into():
            L.call("mz_search_noalloc", *fs._weights(), C.byref(fs.cfg), L.dptr(image), L.dptr(e.to_play), L.dptr(e.legal), L.dptr(noise), S, n,
                   L.dptr(out_v), L.dptr(out_r), L.current_stream())

        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        call_into()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, "mz_search_noalloc changed the device's free memory"
        assert torch.equal(out_v, want_v)
        out_v.fill_(-7)
        out_r.fill_(0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            call_into()
        torch.cuda.synchronize()
        for _ in range(2):
            out_v.fill_(-7)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out_v, want_v) and torch.equal(out_r.view(torch.int32), want_r.view(torch.int32))


def test_noalloc_rejects_sizes_beyond_the_static_array():
    from muzero_amd import _lib as L
    from muzero_amd.env import make_image
    from muzero_amd.mcts import FusedSearch
    from muzero_amd.network import InferenceEngine, Network
    from test_gpu_search import roots
    torch.manual_seed(1261)
    n, S = 512, 50                                             # 512 x 51 nodes x 2 KB > 16 MiB
    served = InferenceEngine(Network(), n, torch.bfloat16)
    e, hw, hb = roots(n, 0)
    image = make_image(hw, hb, 0, n, torch.bfloat16)
    fs = FusedSearch(served, n, S)
    with pytest.raises(L.MzError, match="static array"):
        noalloc(fs, image, e.to_play, e.legal, None, n, S)
    with pytest.raises(L.MzError, match="static limit"):
        noalloc(fs, image, e.to_play, e.legal, None, 1, 5000)
