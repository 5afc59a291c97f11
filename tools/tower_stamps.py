"""per-phase cycle breakdown of the tower kernel from its in-kernel s_memtime stamps:
   python tools/tower_stamps.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd.network import Network, InferenceEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
eng = InferenceEngine(Network(), B, torch.bfloat16, tower="hip")
x = torch.rand(B, 10, 8, 8, device="cuda").to(torch.bfloat16)
import time
t0 = time.time()
while time.time() - t0 < 2.0:   # the clock the chip holds under sustained load (MI355X_MICROARCH.md, DVFS item 6)
    for _ in range(50):
        eng.recurrent(x)
    torch.cuda.synchronize()
nb = (B + 7) // 8   # 8-board tiles
st = torch.zeros((nb, 24), dtype=torch.int64, device="cuda")
eng._packs["dyn"]["struct"].debug_stamps = st.data_ptr()
eng.recurrent(x)
torch.cuda.synchronize()
eng._packs["dyn"]["struct"].debug_stamps = None
s = st.cpu().double()
names = ["first layer", "weight prologue"] + [n for l in range(6) for n in ("mid %d K loop" % l, "mid %d epilogue" % l)] + ["last layer", "heads"]
d = s[:, 1:17] - s[:, 0:16]
tot = (s[:, 16] - s[:, 0])
print("s_memtime ticks (100 MHz constant clock on gfx950? compare total with wall): blocks %d" % nb)
for i, n in enumerate(names):
    print("%-18s mean %9.0f  min %9.0f  max %9.0f   %5.1f %%" % (n, d[:, i].mean(), d[:, i].min(), d[:, i].max(), 100 * d[:, i].mean() / tot.mean()))
print("%-18s mean %9.0f" % ("total per block", tot.mean()))
rt = (s[:, 18] - s[:, 17])
print("in-kernel clock: %.3f GHz (shader cycles / 100 MHz real-time ticks, median over blocks)" % float((tot / rt).median() * 0.1))
# where the launch's wall time goes (s_memrealtime, 10 ns ticks): whole-kernel span vs the tiles' own durations
r0, r1 = s[:, 17], s[:, 18]
span = float(r1.max() - r0.min()) * 0.01
dur = (r1 - r0) * 0.01
print("kernel span first start -> last end: %.1f us; tile duration mean %.1f us (min %.1f, max %.1f); tiles per CU %.2f" %
      (span, float(dur.mean()), float(dur.min()), float(dur.max()), nb / 256.0))
order = torch.argsort(r0)
starts = (r0[order] - r0.min()) * 0.01
print("tile start times (us) at ranks 0, 255, 256, 511, 512, 767, 768, last: " + ", ".join("%.1f" % float(starts[min(i, nb - 1)]) for i in (0, 255, 256, 511, 512, 767, 768, nb - 1)))
ends = (torch.sort(r1).values - r0.min()) * 0.01
print("tile end times (us) at ranks 0, 255, 256, 767, 768, last: " + ", ".join("%.1f" % float(ends[min(i, nb - 1)]) for i in (0, 255, 256, 767, 768, nb - 1)))
# heads batch internals (stamps 19, 21..23 of the tiles whose heads ran): staging | 1x1 convs | Linear 1+2 | softmax
hb = s[s[:, 19] > 0]
if len(hb):
    seg = torch.stack([hb[:, 21] - hb[:, 19], hb[:, 22] - hb[:, 21], hb[:, 23] - hb[:, 22], hb[:, 16] - hb[:, 23], hb[:, 19] - hb[:, 15]], 1)
    print("heads batch (%d batches): staging %.0f | 1x1 convs %.0f | Linear layers %.0f | softmax+out %.0f | (last layer end -> heads start %.0f)"
          % ((len(hb),) + tuple(float(x) for x in seg.mean(0))))
# spread between CUs: a workgroup's whole run (its first tile's start -> its last tile's end), grouped by XCD (workgroups are dealt
# round-robin over the 8 XCDs: wg % 8) -- are whole XCDs slow, or single CUs?
ncu = 256
if nb >= 2 * ncu and nb % ncu == 0:
    k = nb // ncu
    wg_t = (r1.view(k, ncu)[-1] - r0.view(k, ncu)[0]) * 0.01            # us per workgroup
    cyc = (s[:, 16] - s[:, 0]).view(k, ncu).sum(0)                       # shader cycles per workgroup (sum over its tiles)
    print("workgroup run time: mean %.1f us, min %.1f, max %.1f; cycles per workgroup: mean %.0f, min %.0f, max %.0f" %
          (float(wg_t.mean()), float(wg_t.min()), float(wg_t.max()), float(cyc.mean()), float(cyc.min()), float(cyc.max())))
    for x in range(8):
        sel = torch.arange(ncu) % 8 == x
        print("  XCD %d: run time mean %.1f us (min %.1f, max %.1f), cycles mean %.0f, clock %.3f GHz" %
              (x, float(wg_t[sel].mean()), float(wg_t[sel].min()), float(wg_t[sel].max()), float(cyc[sel].mean()),
               float((cyc[sel] / wg_t[sel]).mean()) * 1e-3))
