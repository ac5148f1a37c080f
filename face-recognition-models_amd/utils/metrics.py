"""Top-k accuracy with the reference's signature (main_code/utils/metrics.py:3-16)."""
import torch


def accuracy(output, target, topk=(1,)):
    """Percent of rows whose target is among the k largest entries of `output` [N,C].
    Returns a list of 1-element tensors, one per k, like the reference.  Instead of a full
    top-k selection it counts, per row, how many scores beat the target's (rank < k)."""
    with torch.no_grad():
        n = target.size(0)
        tgt = output.gather(1, target.view(-1, 1))
        rank = (output > tgt).sum(dim=1)
        return [(rank < k).float().sum(0, keepdim=True).mul_(100.0 / n) for k in topk]


def accuracy_from_counts(topk_counts, n):
    """Same percentages from the (#top-1, #top-5) hit counters the fused head kernel produces."""
    return [topk_counts[i:i + 1].float() * (100.0 / n) for i in range(2)]
