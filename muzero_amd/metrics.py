"""Search-level comparison of two runs of the same searches (e.g. bf16 serving vs the reference's fp32 network,
network.py:146-163): what a change of inference arithmetic does to the thing the search outputs -- root visit counts."""
import torch


def visit_agreement(visits_a, visits_b):
    """visits_*: int32 (n,65), -1 = not a root child, same roots / noise / simulation count.  Returns a dict:
    top1 = share of trees whose most-visited action (first maximum, as select_action at temperature 0, main.py:279) agrees;
    identical = share of trees with identical visit vectors; tv = mean total-variation distance of N/sum N;
    kl = mean KL(a || b) of the visit distributions over the root's children with +1/2 pseudo-count smoothing."""
    a, b = visits_a.clamp(min=0).double(), visits_b.clamp(min=0).double()
    child = (visits_a >= 0).double()
    top1 = (a.argmax(1) == b.argmax(1)).double().mean()
    same = (visits_a == visits_b).all(1).double().mean()
    pa, pb = a / a.sum(1, keepdim=True).clamp_min(1), b / b.sum(1, keepdim=True).clamp_min(1)
    tv = 0.5 * (pa - pb).abs().sum(1).mean()
    sa, sb = (a + 0.5) * child, (b + 0.5) * child
    sa, sb = sa / sa.sum(1, keepdim=True), sb / sb.sum(1, keepdim=True)
    kl = (torch.where(child > 0, sa * (sa.clamp_min(1e-300) / sb.clamp_min(1e-300)).log(), torch.zeros_like(sa))).sum(1).mean()
    return {"top1_agreement": float(top1), "identical_visit_vectors": float(same), "mean_total_variation": float(tv),
            "mean_kl_smoothed": float(kl), "trees": int(visits_a.shape[0])}
